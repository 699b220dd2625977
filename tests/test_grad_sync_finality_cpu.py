"""What the gradient exchange may assume of the hand-written backward, and what it does with what it is handed (CPU; r06).

1. FINALITY: a region of the gradient arena handed to GradSync.ready() is never written again in that backward, and the regions tile the arena exactly once — checked
   with a spy that snapshots every region at ready() (the engines run against tests/ops_emulator.py).  The SD3 full fine-tune hands over its fused modulation matrix
   ROW BLOCK BY ROW BLOCK behind the blocks (two interleaved descending sequences of regions), Flux full-rank its blocks back to front.
2. SLICING: several regions may be pending at once, adjacent ranges merge (also across a gap that closes later), a region leaves as soon as it holds a bucket, and no
   collective spans more than max_slice_elems (RCCL's all_to_all_single delivered half of a > 1 GiB chunk on the MI355X: tools/probes/rccl_large_slice_probe.py) —
   sub-slices are cut on multiples of 8 * world from the region's start; world-2 gloo run of all three forms under a tiny cap against the plain sums."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _Spy:
    def __init__(self, model):
        self.model, self.snaps, self.enabled = model, [], True
        self.flat = model.grad_arena

    def begin(self):
        self.snaps = []

    def ready(self, lo, hi):
        self.snaps.append((lo, hi, self.model.grad_arena[lo:hi].clone(), self.model.grad_arena.data_ptr()))

    def finish(self):
        return 1.0


def _check(spy, arena, what):
    cov = torch.zeros(arena.numel(), dtype=torch.int32)
    for lo, hi, snap, ptr in spy.snaps:
        assert ptr == arena.data_ptr(), what
        cov[lo:hi] += 1
        assert torch.equal(snap, arena[lo:hi]), f"{what}: region [{lo}, {hi}) was written after it was handed to the exchange"
    assert bool((cov == 1).all()), f"{what}: the regions handed to the exchange must tile the gradient arena exactly once"


@pytest.mark.parametrize("sd35", [False, True])
def test_sd3_full_finetune_hands_over_final_regions_that_tile_the_arena(monkeypatch, sd35):
    from tests import test_sd3_host_sequencing_cpu as TS
    model = TS._model(monkeypatch, 4, sd35)
    model.enable_full_finetune()
    d = TS._inputs(2, 16, 16, 24)
    TS._hip_side(model, d)                                # reference gradients: no exchange attached
    ref = [p.grad.clone() for p in model.parameters()]
    spy = _Spy(model)
    model.grad_sync = spy
    for step in range(2):                                 # twice: the two gradient arenas take turns
        for p in model.parameters():
            p.grad = None
        arena = model.grad_arena
        TS._hip_side(model, d)
        _check(spy, arena, f"sd3{'.5' if sd35 else ''} step {step}")
        assert all(torch.equal(a, p.grad) for a, p in zip(ref, model.parameters()))
    # the modulation matrix leaves in row blocks BEHIND the blocks, not as one region at the end: its rows appear among the first regions handed over
    D = model.D
    mw_lo = (model.mod_w.data_ptr() - model.arena.data_ptr()) // 2
    mw_hi = mw_lo + model.mod_total * D
    first_mod = next(i for i, (lo, hi, _, _) in enumerate(spy.snaps) if mw_lo <= lo and hi <= mw_hi)
    n_mod = sum(1 for lo, hi, _, _ in spy.snaps if mw_lo <= lo and hi <= mw_hi)
    assert first_mod <= 1 and n_mod == len(model.blocks) + 1, (first_mod, n_mod)


def test_flux_full_rank_hands_over_final_regions_that_tile_the_arena(monkeypatch):
    from tests import test_flux_host_sequencing_cpu as TT
    model = TT._model(monkeypatch, 2, 2)
    model.enable_full_finetune()
    d = TT._inputs(1, 8, 8, 24, seed=50)
    spy = _Spy(model)
    model.grad_sync = spy
    for step in range(2):
        for p in model.parameters():
            p.grad = None
        arena = model.grad_arena
        TT._hip_side(model, d)
        _check(spy, arena, f"flux step {step}")


def test_pending_regions_merge_and_leave_by_the_bucket_and_the_slice_cap():
    from simpletuner_amd.training.grad_sync import GradSync
    flat = torch.zeros(100_000)
    gs = GradSync(flat, bucket_bytes=4 * 10_000)
    gs.max_slice_elems = 16_000
    gs.begin()
    # two interleaved descending sequences (blocks from 100 000 down to 40 000, "modulation rows" from 40 000 down to 4 000) + a front that closes the gaps
    a = [(hi - 6_000, hi) for hi in range(100_000, 40_000, -6_000)]
    b = [(hi - 3_000, hi) for hi in range(40_000, 4_000, -3_000)]
    for i in range(max(len(a), len(b))):
        if i < len(a):
            gs.ready(*a[i])
        if i < len(b):
            gs.ready(*b[i])
    gs.ready(0, 1_000)
    gs.ready(1_000, 4_000)                                  # closes the gap between [0, 1000) and what is left of the second sequence
    assert gs.finish() == 1.0
    cov = torch.zeros(100_000, dtype=torch.int32)
    for lo, hi in gs.launched_slices:
        cov[lo:hi] += 1
        assert 0 < hi - lo <= 16_000
    assert bool((cov == 1).all())
    # the first sequence left in 12 000-element slices (two ranges reach the 10 000-element bucket), the second in 12 000 as well (four ranges)
    assert gs.launched_slices[0] == (88_000, 100_000) and (28_000, 40_000) in gs.launched_slices
    # a region larger than the cap goes out as consecutive sub-slices cut on multiples of 8 from its start
    gs.begin()
    gs.ready(8, 50_003)
    gs.finish()
    assert gs.launched_slices == [(8, 16_008), (16_008, 32_008), (32_008, 48_008), (48_008, 50_003)]


def _worker_cap(rank, world, init_file, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from simpletuner_amd.training.grad_sync import GradSync
    out = {}
    n = 40_008
    for name, dt, mode, fp32 in (("ar", torch.float32, "allreduce", False), ("rs", torch.float32, "rs_ag", False), ("fp32", torch.bfloat16, "rs_ag", True)):
        g = torch.Generator().manual_seed(7 + rank)
        mine = torch.randn(n, generator=g).to(dt)
        flat = mine.clone()
        gs = GradSync(flat, bucket_bytes=flat.element_size() * 6_000, mode=mode, fp32_reduce=fp32)
        gs.max_slice_elems = 2_500                           # every region below is larger: each leaves as several capped sub-slices
        gs.begin()
        hi_a, hi_b = n, 16_000                               # interleaved: [16 000, n) in 8 000s and [0, 16 000) in 4 000s, both back to front
        while hi_a > 16_000 or hi_b > 0:
            if hi_a > 16_000:
                gs.ready(max(16_000, hi_a - 8_000), hi_a); hi_a = max(16_000, hi_a - 8_000)
            if hi_b > 0:
                gs.ready(hi_b - 4_000, hi_b); hi_b -= 4_000
        scale = gs.finish()
        out[name] = (mine, flat.clone(), scale, list(gs.launched_slices), sorted({k for k, _, _ in gs.launched_ops}))
    torch.save(out, os.path.join(out_dir, f"cap_{rank}.pt"))
    dist.destroy_process_group()


def test_two_process_exchange_under_a_slice_cap_with_interleaved_regions():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_cap, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"cap_{r}.pt")) for r in range(2))
    for name in ("ar", "rs", "fp32"):
        m0, f0, s0, sl0, ops0 = r0[name]
        m1, f1, s1, sl1, ops1 = r1[name]
        assert s0 == s1 == 0.5 and sl0 == sl1 and ops0 == ops1
        assert all(hi - lo <= 2_500 for lo, hi in sl0) and sum(hi - lo for lo, hi in sl0) == 40_008
        assert torch.equal(f0, f1), name
        want = m0.float() + m1.float()
        if name == "fp32":
            assert "all_to_all" in ops0
            assert torch.equal(f0, want.to(torch.bfloat16)), name          # fp32 accumulation, ONE rounding
        else:
            assert torch.equal(f0, want), name
