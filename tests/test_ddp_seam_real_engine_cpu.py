"""The DDP seam with the REAL Flux engine, on the CPU (gloo, world 2; kernels replaced by tests/ops_emulator.py): the trained component wrapped in torch's own
DistributedDataParallel exactly as `accelerator.prepare` wraps it (trainer.py:4564-4571) + `install_ddp_comm_hook`, and in the reducer-free `St355DistributedDataParallel`.
After a synchronised backward every adapter's `.grad` is the MEAN over the replicas of the local gradients (DDP's contract); under `no_sync()` nothing is exchanged.
(tests/test_ddp_seam_cpu.py checks the same contract on a toy component, incl. the accumulate boundary.)"""
import os
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, init_file, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    import pytest
    from simpletuner_amd.training.ddp_seam import St355DistributedDataParallel, install_ddp_comm_hook
    from tests import test_flux_host_sequencing_cpu as TT
    patch = pytest.MonkeyPatch()
    res = {}
    d = TT._inputs(1, 8, 8, 24, seed=60 + rank)                   # each replica sees its own sample

    def flat_grads(model):
        return torch.cat([p.grad.reshape(-1) for p in model.trainable_parameters()]).clone()

    def call(m):
        out = m(hidden_states=d["packed"], encoder_hidden_states=d["prompt"], pooled_projections=d["pooled"], timestep=d["t"], img_ids=d["img_ids"], txt_ids=d["txt_ids"],
                guidance=d["guidance"], return_dict=False)[0]
        ((out.float() - d["target"].float()) ** 2).mean().backward()

    for tag in ("ddp_hook", "wrapper"):
        model = TT._model(patch, 1, 1, seed=11 + 7 * rank)        # replicas deliberately start apart: wrapping brings them to rank 0's state
        model.add_lora_adapter(rank=8, alpha=8.0, targets="default", seed=5 + rank, init_b_std=0.02)
        if tag == "ddp_hook":
            wrapped = torch.nn.parallel.DistributedDataParallel(model)
            install_ddp_comm_hook(wrapped)
        else:
            wrapped = St355DistributedDataParallel(model)
        start = model.lora_flat.clone()
        gathered = [torch.empty_like(start) for _ in range(world)]
        dist.all_gather(gathered, start)
        res[tag + "_start_equal"] = all(torch.equal(g, gathered[0]) for g in gathered)
        # the local gradient, nothing exchanged
        with wrapped.no_sync():
            call(wrapped)
        local = flat_grads(model)
        for p in model.parameters():
            p.grad = None
        want = local.clone()
        dist.all_reduce(want)
        want /= world
        # a synchronised backward
        call(wrapped)
        got = flat_grads(model)
        res[tag + "_mean"] = bool(torch.allclose(got, want, rtol=1e-5, atol=1e-7))
        res[tag + "_differs_from_local"] = not torch.allclose(got, local, rtol=1e-3, atol=1e-6)
    patch.undo()
    torch.save(res, os.path.join(out_dir, f"seam_{rank}.pt"))
    dist.destroy_process_group()


def test_real_flux_engine_under_torch_ddp_with_the_comm_hook_and_under_the_reducer_free_wrapper():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        results = [torch.load(os.path.join(d, f"seam_{r}.pt")) for r in range(2)]
    for res in results:
        assert all(res.values()), res
