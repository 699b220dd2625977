"""world_size-2 `gloo` test of the N>1 path of full-rank Flux training (runs in the CPU-only container): the engine's hand-written backward
(`_engine_backward_full`, executed against tests/ops_emulator.py) hands the gradient arena to GradSync slice by slice — output head, single blocks back to
front, double blocks back to front, embedders + modulation matrix — and the exchange overlaps the rest of the backward.  Checked: the slices tile the arena
exactly once, both replicas end with the same reduced arena, and it equals the sum of the two replicas' local gradients (fp32-accumulated, one rounding)."""
import os
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, init_file, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    import pytest
    from simpletuner_amd.training.grad_sync import GradSync
    from tests import test_flux_host_sequencing_cpu as TT
    patch = pytest.MonkeyPatch()
    model = TT._model(patch, 2, 2)                       # same seed on both ranks: replicas start from identical weights
    model.enable_full_finetune()
    d = TT._inputs(1, 8, 8, 24, seed=50 + rank)          # each replica sees its own samples
    TT._hip_side(model, d)
    local = model._last_grad_flat.clone()
    for p in model.parameters():
        p.grad = None
    model.grad_sync = GradSync(model.grad_arena, bucket_bytes=2 * 40_000)
    TT._hip_side(model, d)
    torch.save({"local": local, "synced": model._last_grad_flat.clone(), "scale": model.grad_scale_from_sync, "slices": list(model.grad_sync.launched_slices),
                "numel": model.grad_arena.numel(), "ops": sorted({k for k, _, _ in model.grad_sync.launched_ops})}, os.path.join(out_dir, f"fr_{rank}.pt"))
    patch.undo()
    dist.destroy_process_group()


def test_two_replicas_of_full_rank_flux_exchange_the_whole_gradient_arena():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"fr_{r}.pt")) for r in range(2))
    n = r0["numel"]
    covered = torch.zeros(n, dtype=torch.int32)
    for lo, hi in r0["slices"]:
        covered[lo:hi] += 1
    assert bool((covered == 1).all()), "the slices handed to the exchange must tile the gradient arena exactly once"
    assert r0["slices"] == r1["slices"] and len(r0["slices"]) >= 4             # bucketed (adjacent ready() ranges coalesce up to the bucket size), same order on both replicas
    assert r0["scale"] == r1["scale"] == 0.5
    assert torch.equal(r0["synced"], r1["synced"])
    want = (r0["local"].float() + r1["local"].float())
    diff = (r0["synced"].float() - want).abs()
    tol = 2.0 ** -7 * want.abs() + 1e-30                                      # one bf16 rounding of the fp32 sum (tails of a slice: a bf16 add, same bound)
    assert bool((diff <= tol).all()), (diff.max().item(), want.abs().max().item())
    assert not torch.equal(r0["local"], r1["local"])
