"""world_size-2 `gloo` run of the REAL training loop on the CPU (kernels replaced by tests/ops_emulator.py): two replicas, each with half of a batch, against one process
with the whole batch.  What runs is the product's N > 1 path end to end — `Trainer.__init__` (replicas start from rank 0's weights, GradSync attached to the flat gradient
arena), the engine's backward handing slices to the exchange, the 1/world scale folded into the fused optimizer, the sample-weighted loss gather — with the Flux engine's real
forward / backward.  (The GPU form, two processes sharing one MI355X over gloo, is tests/test_distributed_gpu.py; RCCL at N > 1 is the driver's SCALE run.)"""
import os
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

STEPS = 3


def _run(rank, world, full, accum=1, target="default"):
    import pytest
    from simpletuner_amd.flux import transformer as T
    from simpletuner_amd.flux.model import Flux
    from simpletuner_amd.training.trainer import St355Accelerator, Trainer, default_config
    from tests import ops_emulator as EMU
    from tests import parity_utils as PU
    patch = pytest.MonkeyPatch()
    EMU.install(patch)
    patch.setattr(T, "_FUSED_QKV", False); patch.setattr(T, "_BLOCK_ABI", False)
    B = 2 // world
    cfg = default_config(train_batch_size=B, seed=3, flow_schedule_shift=3.0, lora_rank=8, lora_init_b_std=0.02, learning_rate=1e-3, model_type="full" if full else "lora",
                         gradient_accumulation_steps=accum, flux_lora_target=target)
    acc = St355Accelerator(torch.device("cpu"), gradient_accumulation_steps=accum)
    plugin = Flux(cfg, acc)
    torch.manual_seed(100 + rank)                                 # replicas deliberately start apart: the constructor must bring them to rank 0's weights
    plugin.load_model(**PU.small_flux_cfg(layers=1, single=1))
    with torch.no_grad():
        for p in plugin.model.parameters():
            p.add_(0.01 * rank)
    if full:
        plugin.freeze_components()
    else:
        plugin.add_lora_adapter()
    trainer = Trainer(cfg, plugin, acc)
    _, devt = PU.make_inputs(2, 8, 8, 24, 128, 64, "cpu", seed=3)
    mine = {k: v[rank * B:(rank + 1) * B] for k, v in devt.items()}
    sig = mine["sigmas"]
    plugin.sample_flow_sigmas = lambda batch, state: (sig, sig * 1000.0)
    losses = []
    for _ in range(STEPS * accum):
        losses.append(float(trainer.train_step({"latent_batch": mine["latents"], "prompt_embeds": mine["prompt"], "add_text_embeds": mine["pooled"], "noise": mine["noise"]})))
    comp = plugin.get_trained_component()
    flat = comp.arena.clone() if full else comp.lora_flat.clone()
    patch.undo()
    return flat, losses, comp.grad_sync is not None


def _worker(rank, world, init_file, out_dir, full, accum=1, target="default"):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    flat, losses, has_sync = _run(rank, world, full, accum, target)
    torch.save({"flat": flat, "losses": losses, "has_sync": has_sync}, os.path.join(out_dir, f"tr_{int(full)}_{rank}.pt"))
    dist.destroy_process_group()


def _check(full, accum=1, target="default"):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "init"), d, full, accum, target), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"tr_{int(full)}_{r}.pt")) for r in range(2))
    one, losses_one, _ = _run(0, 1, full, accum, target)
    assert (r0["has_sync"] and r1["has_sync"]) == (accum == 1)              # with accumulation the exchange is ONE all-reduce of the accumulated gradient at the boundary
    assert torch.equal(r0["flat"], r1["flat"]), "replicas must hold identical weights after every synchronised step"
    assert r0["losses"] == r1["losses"]                                        # the logged loss is the sample-weighted mean over ranks
    assert max(abs(a - b) for a, b in zip(r0["losses"], losses_one)) < 2e-3    # == the whole batch's loss
    # weights after K steps: Adam's update is lr * sign-like, so the two runs may differ by a few lr per element where a gradient is rounding noise
    diff = (r0["flat"].float() - one.float()).abs().max().item()
    assert diff <= 2.05 * 1e-3 * STEPS, diff
    moved = (one.float() - _run(0, 1, full, accum, target)[0].float()).abs().max().item()
    assert moved == 0.0                                                        # and the single-process run itself is reproducible


def test_two_lora_replicas_equal_one_process_with_the_whole_batch():
    _check(False)


def test_two_full_rank_replicas_equal_one_process_with_the_whole_batch():
    _check(True)


def test_two_replicas_with_gradient_accumulation_reduce_once_at_the_boundary():
    """gradient_accumulation_steps = 2 (accelerator.accumulate / DDP no_sync semantics, trainer.py:7009): micro-steps accumulate locally, the boundary step all-reduces the
    accumulated flat gradient once; same weights as one process accumulating over the whole batch"""
    _check(False, accum=2)


def test_two_replicas_with_modulation_adapters():
    """flux_lora_target = "ai-toolkit": the modulation Linears' adapter group is the LAST slice handed to the exchange (its dy is complete only after block 0)"""
    _check(False, target="ai-toolkit")


def test_two_replicas_with_feed_forward_and_embedder_adapters():
    """flux_lora_target = "all+ffs+embedder": the feed-forward, proj_mlp / proj_out, output-projection and x_embedder adapter groups hand their slices of the flat gradient to
    the exchange as the host-sequenced backward produces them (the output projection first, x_embedder last); two replicas end on identical weights, equal to one
    process with the whole batch"""
    _check(False, target="all+ffs+embedder")
