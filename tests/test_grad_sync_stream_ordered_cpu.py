"""The `nccl` branch of GradSync (training/grad_sync.py: async work handles that are only joined in finish(), dependent collectives issued back to back, the
all-to-all + fp32 local sum + all-gather form) has never met more than one rank on hardware: the lease is one GPU.  This test executes exactly that bookkeeping at
world 2 and 4 over gloo by standing in for what makes RCCL different — STREAM ORDER: every collective the module issues runs to completion in issue order (as
kernels on one comm stream do) and hands back a handle whose wait() only counts.  Checked: the arena holds the SUM over ranks, the ops are issued in the nccl
branch's order with no wait between dependent collectives, every handle is joined exactly once and only inside finish()."""
import os
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _OrderedWork:
    """an RCCL work handle as GradSync sees it: the collective is already ordered behind its predecessors on the comm stream; wait() joins, nothing else"""
    issued = []
    in_finish = False
    early_waits = 0

    def __init__(self, work):
        work.wait()                      # executed in issue order = stream order
        self.waits = 0
        _OrderedWork.issued.append(self)

    def wait(self):
        self.waits += 1
        if not _OrderedWork.in_finish:
            _OrderedWork.early_waits += 1
        return True


class _StreamOrderedDist:
    """torch.distributed as the module sees it, with a stream-ordered backend's semantics over gloo"""

    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        return getattr(dist, name)

    def get_backend(self, group=None):
        return "nccl"

    def _run(self, name, fn, args, kw):
        async_op = kw.pop("async_op", False)
        self.calls.append((name, async_op))
        w = _OrderedWork(fn(*args, async_op=True, **kw))
        return w if async_op else None

    def all_reduce(self, *a, **k):
        return self._run("all_reduce", dist.all_reduce, a, k)

    def reduce_scatter_tensor(self, *a, **k):
        return self._run("reduce_scatter", dist.reduce_scatter_tensor, a, k)

    def all_gather_into_tensor(self, *a, **k):
        return self._run("all_gather", dist.all_gather_into_tensor, a, k)

    def all_to_all_single(self, *a, **k):
        return self._run("all_to_all", dist.all_to_all_single, a, k)


def _worker(rank, world, init_file, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    import simpletuner_amd.training.grad_sync as GS

    shim = _StreamOrderedDist()
    GS.dist = shim
    out = {}
    n = 40_000 + 8 * world + 5                                       # every slice leaves a tail for the all-reduce form
    for name, mode, dtype, fp32 in (("allreduce", "allreduce", torch.float32, False), ("rs_ag", "rs_ag", torch.float32, False),
                                    ("rs_ag_fp32_reduce", "rs_ag", torch.bfloat16, True)):
        g = torch.Generator().manual_seed(7)
        base = (torch.randn(n, generator=g) * 4).round() / 4                  # exactly representable sums in bf16 as well
        flat = (base * (rank + 1)).to(dtype)
        gs = GS.GradSync(flat, bucket_bytes=flat.element_size() * 9000, mode=mode, fp32_reduce=fp32)
        if fp32:
            gs._fp32_alltoall_ok = lambda: True                      # (a device arena under RCCL: the three steps are ordered by the comm stream)
        assert gs._stream_ordered()
        _OrderedWork.issued, _OrderedWork.early_waits, shim.calls = [], 0, []
        gs.begin()
        edges = list(range(0, n, 7008)) + [n]                        # 8-aligned region starts (arena tensors are padded to 8 elements), a ragged end
        for lo, hi in reversed(list(zip(edges[:-1], edges[1:]))):    # back to front, like the hand-written backward
            gs.ready(lo, hi)
        pending = sum(w.waits for w in _OrderedWork.issued)
        _OrderedWork.in_finish = True
        scale = gs.finish()
        _OrderedWork.in_finish = False
        want = (base * (world * (world + 1) / 2)).to(dtype)
        out[name] = dict(ok=bool(torch.equal(flat, want)), scale=scale, waits_before_finish=pending, early=_OrderedWork.early_waits,
                         handles=len(_OrderedWork.issued),                          async_handles=sum(1 for _, a in shim.calls if a), waits_total=sum(w.waits for w in _OrderedWork.issued),
                         ops=[o[0] for o in gs.launched_ops], calls=list(shim.calls), works_left=len(gs._works))
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def _run(world):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, os.path.join(d, "init"), d), nprocs=world, join=True)
        return [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]


def _check(res, world):
    for r in res:
        for name, o in r.items():
            assert o["ok"], (name, "the arena does not hold the SUM over ranks")
            assert o["scale"] == 1.0 / world
            assert o["early"] == 0 and o["waits_before_finish"] == 0, (name, "a handle was joined before finish(): the nccl branch must not block the backward")
            assert o["waits_total"] == o["async_handles"] and o["works_left"] == 0, (name, o["waits_total"], o["async_handles"])
        # rs_ag: reduce-scatter and the all-gather that depends on it are issued back to back, asynchronously, then the < world tail
        ops = r["rs_ag"]["ops"]
        assert ops[:2] == ["reduce_scatter", "all_gather"] and ops.count("reduce_scatter") == ops.count("all_gather") >= 3 and "all_reduce" in ops
        assert all(ops[i + 1] == "all_gather" for i, o in enumerate(ops) if o == "reduce_scatter")
        assert all(a for _, a in r["rs_ag"]["calls"]), "every collective of the rs_ag form is asynchronous under a stream-ordered backend"
        # fp32 reduce: all-to-all (blocking call, stream-ordered), local fp32 sum, all-gather, tail all-reduce
        ops = r["rs_ag_fp32_reduce"]["ops"]
        assert ops[:2] == ["all_to_all", "all_gather"] and "reduce_scatter" not in ops and "all_reduce" in ops
        assert all(ops[i + 1] == "all_gather" for i, o in enumerate(ops) if o == "all_to_all")
        assert set(r["allreduce"]["ops"]) == {"all_reduce"}


def test_stream_ordered_branch_world_2():
    _check(_run(2), 2)


def test_stream_ordered_branch_world_4():
    _check(_run(4), 4)
