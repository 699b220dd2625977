"""world_size-2 `gloo` tests of the N>1 path (runs in the CPU-only container): the bucketed gradient all-reduce driven in
back-to-front order like the hand-written backward, the folded 1/world scale, no_sync semantics, and the sample-weighted loss
gather (port of the reference's one real multi-process test, tests/test_distributed_batch_layout.py:48-119)."""
import os
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, init_file, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from simpletuner_amd.training.grad_sync import GradSync
    from simpletuner_amd.training.multi_process import any_rank_reached_epoch_end, gather_sample_weighted_scalar

    res = {}
    # --- C1: bucketed all-reduce over a flat gradient arena, slices finishing back-to-front ---
    n = 10_000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)          # rank r holds (r+1) * arange
    gs = GradSync(flat, bucket_bytes=4 * 2048)
    gs.begin()
    edges = list(range(n, 0, -700)) + [0]
    for hi, lo in zip(edges[:-1], edges[1:]):
        gs.ready(lo, hi)
    scale = gs.finish()
    res["scale"] = scale
    res["avg_ok"] = torch.allclose(flat * scale, torch.arange(n, dtype=torch.float32) * (1 + 2) / 2)
    res["n_buckets"] = len(gs.launched_slices)
    # --- no_sync: nothing is reduced, scale stays 1 ---
    flat2 = torch.full((100,), float(rank + 1))
    gs2 = GradSync(flat2)
    with gs2.no_sync():
        gs2.begin(); gs2.ready(0, 100); s2 = gs2.finish()
    res["nosync_ok"] = (s2 == 1.0) and bool((flat2 == rank + 1).all())
    # --- C2: weighted loss: rank0 loss 2.0 (1 sample), rank1 loss 4.0 (3 samples) -> 3.5 ---
    loss = torch.tensor(2.0 if rank == 0 else 4.0)
    res["weighted"] = gather_sample_weighted_scalar(loss, 1 if rank == 0 else 3).item()
    # --- C3: epoch-end consensus (MAX) ---
    res["epoch_end"] = any_rank_reached_epoch_end(rank == 1, torch.device("cpu"))
    # --- replica start state: every distinct storage behind the module's parameters / buffers is broadcast from rank 0 as raw bytes ---
    from simpletuner_amd.training.grad_sync import sync_module_states
    torch.manual_seed(100 + rank)                                      # replicas deliberately start apart
    arena = torch.randn(4096)                                          # one arena, three parameter views + padding that is no parameter
    mod = torch.nn.Module()
    mod.a = torch.nn.Parameter(arena[0:1000].view(10, 100))
    mod.b = torch.nn.Parameter(arena[1024:2048].view(32, 32), requires_grad=False)
    mod.c = torch.nn.Parameter(torch.randn(7, 3).to(torch.bfloat16))
    mod.register_buffer("tab", torch.randn(5))
    calls = []
    mod._refresh_transposed = lambda: calls.append(1)
    mod.inner = torch.nn.Module()
    mod.inner._prepared = True                                         # a lazily prepared sub-model must be re-prepared from the new weights
    nbytes = sync_module_states(mod, chunk_bytes=5000)                 # forces the chunked path (arena = 16 KiB)
    torch.manual_seed(100)
    want_arena = torch.randn(4096); want_c = torch.randn(7, 3).to(torch.bfloat16); want_tab = torch.randn(5)
    res["sync_ok"] = (torch.equal(arena, want_arena) and torch.equal(mod.c.data, want_c) and torch.equal(mod.tab, want_tab)
                      and torch.equal(mod.a.data, want_arena[:1000].view(10, 100)) and calls == [1] and mod.inner._prepared is False)
    res["sync_bytes"] = nbytes
    # --- Trainer construction under world 2: replicas leave __init__ with rank 0's trainable weights, adapters' gradient arena gets a GradSync ---
    from simpletuner_amd.foundation import ModelFoundation
    from simpletuner_amd.training.trainer import St355Accelerator, Trainer, default_config

    class Comp(torch.nn.Module):
        def __init__(self, seed):
            super().__init__()
            self.flat = (torch.randn(64, generator=torch.Generator().manual_seed(seed)) * 0.1)
            self.lora_grad_flat = torch.zeros(64)
            self.a = torch.nn.Parameter(self.flat[:32].view(4, 8))
            self.b = torch.nn.Parameter(self.flat[32:].view(8, 4))
            self.grad_sync = None

        def trainable_parameters(self):
            return [self.a, self.b]

    acc = St355Accelerator(torch.device("cpu"))
    cfg = default_config(model_type="lora")
    plug = ModelFoundation(cfg, acc)
    plug.model = Comp(seed=500 + rank)
    tr = Trainer(cfg, plug, acc)
    want = torch.randn(64, generator=torch.Generator().manual_seed(500)) * 0.1
    res["trainer_sync_ok"] = bool(acc.num_processes == 2 and torch.equal(plug.model.flat, want) and plug.model.grad_sync is not None
                                  and plug.model.grad_sync.world_size == 2 and len(tr.params) == 2)
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_process_gloo_grad_sync_and_loss_gather():
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "init")
        mp.spawn(_worker, args=(2, init_file, d), nprocs=2, join=True)
        for r in range(2):
            res = torch.load(os.path.join(d, f"r{r}.pt"))
            assert res["scale"] == 0.5
            assert res["avg_ok"]
            assert res["n_buckets"] >= 3
            assert res["nosync_ok"]
            assert abs(res["weighted"] - 3.5) < 1e-6
            assert res["epoch_end"] is True
            assert res["trainer_sync_ok"]
            assert res["sync_ok"] and res["sync_bytes"] == 4096 * 4 + 7 * 3 * 2 + 5 * 4


def _worker_rs_ag(rank, world, init_file, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from simpletuner_amd.training.grad_sync import GradSync

    n = 10_007                                                        # odd: every slice leaves a < world tail for the all-reduce form
    base = torch.arange(n, dtype=torch.float32)
    out = {}
    for mode in ("rs_ag", "allreduce"):
        flat = base * (rank + 1) + (0.25 if rank == 1 else 0.0)
        gs = GradSync(flat, bucket_bytes=4 * 1500, mode=mode)
        gs.begin()
        edges = list(range(n, 0, -701)) + [0]
        for hi, lo in zip(edges[:-1], edges[1:]):
            gs.ready(lo, hi)
        scale = gs.finish()
        out[mode] = (flat.clone(), scale, list(gs.launched_ops), list(gs.launched_slices))
    # auto: small arenas take the all-reduce form, >= RS_AG_MIN_BYTES the reduce-scatter + all-gather form
    import simpletuner_amd.training.grad_sync as GS
    out["auto_small"] = GradSync(torch.zeros(64)).mode
    old = GS.RS_AG_MIN_BYTES
    GS.RS_AG_MIN_BYTES = 128
    out["auto_big"] = GradSync(torch.zeros(64)).mode
    GS.RS_AG_MIN_BYTES = old
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_process_reduce_scatter_all_gather_form_equals_all_reduce():
    """the rs_ag exchange (reduce-scatter into this rank's shard, all-gather of the shards, in place on the arena; a < world tail of every
    slice goes through all-reduce) leaves the same SUM in every rank's arena as the all-reduce form, bit for bit"""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_rs_ag, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"r{r}.pt")) for r in range(2))
    want = torch.arange(10_007, dtype=torch.float32) * 3 + 0.25
    for res in (r0, r1):
        for mode in ("rs_ag", "allreduce"):
            flat, scale, ops_, slices = res[mode]
            assert torch.equal(flat, want), mode
            assert scale == 0.5
            assert sum(hi - lo for lo, hi in slices) == 10_007
        kinds = [k for k, _, _ in res["rs_ag"][2]]
        assert kinds.count("reduce_scatter") == kinds.count("all_gather") >= 3 and "all_reduce" in kinds     # odd slices leave 1-element tails
        for (k1, lo1, hi1), (k2, lo2, hi2) in zip(res["rs_ag"][2], res["rs_ag"][2][1:]):
            if k1 == "reduce_scatter":
                assert (k2, lo2, hi2) == ("all_gather", lo1, hi1) and (hi1 - lo1) % 2 == 0
        assert all(k == "all_reduce" for k, _, _ in res["allreduce"][2])
        assert res["auto_small"] == "allreduce" and res["auto_big"] == "rs_ag"


def _worker_world4(rank, world, init_file, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from simpletuner_amd.training.grad_sync import GradSync
    n = 20_011                                                        # prime: slices leave 1..3-element tails for the all-reduce form
    out = {}
    # fp32 arena, both forms
    for mode in ("rs_ag", "allreduce"):
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
        gs = GradSync(flat, bucket_bytes=4 * 3000, mode=mode)
        gs.begin()
        edges = list(range(n, 0, -1303)) + [0]
        for hi, lo in zip(edges[:-1], edges[1:]):
            gs.ready(lo, hi)
        scale = gs.finish()                                           # (joins the async works: read the arena only after it)
        out[mode] = (flat.clone(), scale, sorted({k for k, _, _ in gs.launched_ops}))
    # bf16 arena: the fp32-accumulating form (all-to-all of the W chunks, fp32 sum in rank order, all-gather)
    # (arena regions start on 8-element boundaries — tensors are padded to 8 — which the fp32 form needs: st355_sum_chunks_bf16's 16-byte shards)
    n2 = 20_008
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(n2, generator=g).to(torch.bfloat16)
    flat = mine.clone()
    gs = GradSync(flat, bucket_bytes=2 * 4096, mode="rs_ag", fp32_reduce=True)        # (opt-in: grad_sync.py)
    gs.begin()
    for hi in range(n2, 0, -2776):
        gs.ready(max(0, hi - 2776), hi)
    scale = gs.finish()
    out["bf16"] = (flat.clone(), scale, sorted({k for k, _, _ in gs.launched_ops}), list(gs.launched_slices))
    torch.save(out, os.path.join(out_dir, f"w4_{rank}.pt"))
    dist.destroy_process_group()


def test_four_process_gradient_exchange_all_forms():
    """the driver's scaling run uses 2, 4 and 8 ranks: the same exchange at world 4 over gloo — both fp32 forms leave the exact SUM, the bf16 arena's
    fp32-accumulating form (all-to-all + local fp32 sum in rank order + all-gather; a < 8 x world tail of every slice through all-reduce) leaves the sum of the four
    ranks' values accumulated in fp32 and rounded once, identical on every rank; scale = 1/4"""
    W = 4
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_world4, args=(W, os.path.join(d, "init"), d), nprocs=W, join=True)
        res = [torch.load(os.path.join(d, f"w4_{r}.pt")) for r in range(W)]
    n = 20_011
    want = torch.arange(n, dtype=torch.float32) * (1 + 2 + 3 + 4)
    n2 = 20_008
    vals = [torch.randn(n2, generator=torch.Generator().manual_seed(100 + r)).to(torch.bfloat16) for r in range(W)]
    exact = (vals[0].float() + vals[1].float() + vals[2].float() + vals[3].float()).to(torch.bfloat16)
    for r in res:
        for mode in ("rs_ag", "allreduce"):
            flat, scale, kinds = r[mode]
            assert torch.equal(flat, want) and scale == 0.25, mode
        assert "reduce_scatter" in res[0]["rs_ag"][2] and "all_gather" in res[0]["rs_ag"][2] and res[0]["allreduce"][2] == ["all_reduce"]
        flat, scale, kinds, slices = r["bf16"]
        assert scale == 0.25 and "all_to_all" in kinds and "all_gather" in kinds
        assert sum(hi - lo for lo, hi in slices) == n2
        assert torch.equal(flat, res[0]["bf16"][0])                                   # every rank holds the same reduced arena
        # the rs_ag part is the exactly-rounded fp32 sum; the < 8 x world tails went through a bf16 all-reduce (pairwise bf16 adds): compare those loosely
        diff = (flat.float() - exact.float()).abs()
        assert (diff == 0).float().mean().item() > 0.99 and diff.max().item() <= 0.0625 * exact.float().abs().max().item()


def _worker_world3_fp32_selection(rank, world, init_file, out_dir):
    import logging
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from simpletuner_amd.training.grad_sync import GradSync
    out = {}
    n = 30_000 + 16                                                    # ragged: the slices below are no multiple of 8 x 3
    vals = [torch.randn(n, generator=torch.Generator().manual_seed(300 + r)).to(torch.bfloat16) for r in range(world)]

    def run(flat, lo_edges, **kw):
        gs = GradSync(flat, bucket_bytes=2 * 4000, mode="rs_ag", **kw)
        gs.begin()
        for hi, lo in zip(lo_edges[:-1], lo_edges[1:]):
            gs.ready(lo, hi)
        gs.finish()
        return gs
    edges = list(range(n, 0, -4104)) + [0]                             # 4104 = 8 * 513: every slice starts 8-aligned
    # default: RCCL / gloo's own reduce-scatter in the arena dtype (bf16 on the wire AND in the adds)
    os.environ.pop("ST355_FP32_REDUCE", None)
    gs = run(vals[rank].clone(), edges)
    out["default_flag"], out["default_ops"] = gs.fp32_reduce, sorted({k for k, _, _ in gs.launched_ops})
    # ST355_FP32_REDUCE=1: all-to-all + fp32 sum in rank order + all-gather
    os.environ["ST355_FP32_REDUCE"] = "1"
    flat = vals[rank].clone()
    gs = run(flat, edges)
    out["env_flag"], out["env_ops"], out["env_flat"] = gs.fp32_reduce, sorted({k for k, _, _ in gs.launched_ops}), flat
    out["env_on_fp32_arena"] = GradSync(torch.zeros(64), mode="rs_ag").fp32_reduce          # the flag only concerns bf16 arenas
    # an unaligned region start: the fp32 form steps aside for that slice, and says so once
    seen = []
    h = logging.Handler(); h.emit = lambda rec: seen.append(rec.getMessage())
    logging.getLogger("st355.grad_sync").addHandler(h)
    flat2 = vals[rank].clone()
    gs = run(flat2, [n, 4 + 8 * 3 * 100, 4, 0])                        # [4 + 2400, n) is 8-aligned?  no: lo = 2404 -> falls back; [4, 2404) too
    out["unaligned_ops"], out["unaligned_logged"], out["unaligned_flat"] = sorted({k for k, _, _ in gs.launched_ops}), len(seen), flat2
    os.environ.pop("ST355_FP32_REDUCE", None)
    torch.save(out, os.path.join(out_dir, f"w3_{rank}.pt"))
    dist.destroy_process_group()


def test_fp32_reduce_selection_default_env_and_unaligned_fallback_world3():
    """the default exchange of a bf16 arena is the backend's own reduce-scatter; ST355_FP32_REDUCE=1 selects all-to-all + fp32 sum + all-gather (world 3,
    ragged slices: the sum of three ranks accumulated in fp32 and rounded once, the same arena on every rank); a slice whose start is not 8-element aligned falls back
    to the backend's reduce-scatter for that slice and logs it once"""
    W = 3
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_world3_fp32_selection, args=(W, os.path.join(d, "init"), d), nprocs=W, join=True)
        res = [torch.load(os.path.join(d, f"w3_{r}.pt")) for r in range(W)]
    n = 30_000 + 16
    vals = [torch.randn(n, generator=torch.Generator().manual_seed(300 + r)).to(torch.bfloat16) for r in range(W)]
    exact = (vals[0].float() + vals[1].float() + vals[2].float()).to(torch.bfloat16)
    for r in res:
        assert r["default_flag"] is False and "all_to_all" not in r["default_ops"] and "reduce_scatter" in r["default_ops"]
        assert r["env_flag"] is True and "all_to_all" in r["env_ops"] and "reduce_scatter" not in r["env_ops"] and r["env_on_fp32_arena"] is False
        assert torch.equal(r["env_flat"], res[0]["env_flat"])
        diff = (r["env_flat"].float() - exact.float()).abs()
        assert (diff == 0).float().mean().item() > 0.99                 # all but the < 8 x world tails of the slices are the exactly-rounded fp32 sum
        assert "reduce_scatter" in r["unaligned_ops"] and r["unaligned_logged"] == 1
        assert torch.equal(r["unaligned_flat"], res[0]["unaligned_flat"])
        assert (r["unaligned_flat"].float() - exact.float()).abs().max().item() <= 0.0625 * exact.float().abs().max().item()
