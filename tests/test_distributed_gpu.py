"""Replica start state with DEVICE storages (two processes sharing cuda:0 over gloo — a 1-GPU box cannot run RCCL between two ranks; the RCCL
path is the same code with backend "nccl"): sync_module_states broadcasts the raw bytes of every distinct storage behind the module, including
bytes that belong to no registered parameter (arena padding / fused LoRA columns)."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, init_file, out_dir):
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from simpletuner_amd.training.grad_sync import GradSync, sync_module_states
    dev = torch.device("cuda", 0)
    torch.manual_seed(100 + rank)
    arena = torch.randn(1 << 16, device=dev).to(torch.bfloat16)          # 128 KiB bf16 arena: two views + a gap
    mod = torch.nn.Module()
    mod.a = torch.nn.Parameter(arena[:30000].view(100, 300), requires_grad=False)
    mod.b = torch.nn.Parameter(arena[32768:].view(128, 256))
    mod.c = torch.nn.Parameter(torch.randn(33, device=dev))
    n = sync_module_states(mod, chunk_bytes=50_000)
    # bucketed gradient all-reduce on a device arena, back to front
    flat = torch.full((40_000,), float(rank + 1), device=dev)
    gs = GradSync(flat, bucket_bytes=4 * 8192)
    gs.begin()
    for hi in range(40_000, 0, -5000):
        gs.ready(hi - 5000, hi)
    scale = gs.finish()
    torch.cuda.synchronize()
    torch.save({"arena": arena.cpu(), "c": mod.c.detach().cpu(), "n": n, "avg": (flat * scale).cpu(), "buckets": len(gs.launched_slices)},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_process_module_state_broadcast_and_grad_sync_on_device_storages():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"r{r}.pt")) for r in range(2))
    torch.manual_seed(100)
    want = torch.randn(1 << 16, device="cuda:0").to(torch.bfloat16).cpu()
    assert torch.equal(r0["arena"], want) and torch.equal(r1["arena"], want) and torch.equal(r1["c"], r0["c"])
    assert r0["n"] == r1["n"] == (1 << 16) * 2 + 33 * 4
    assert torch.equal(r0["avg"], torch.full((40_000,), 1.5)) and torch.equal(r1["avg"], r0["avg"]) and r0["buckets"] >= 4
