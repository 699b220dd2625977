"""Which step of the fp32-accumulating exchange breaks on a 1.6 GB slice at world 1?  (r06, tools/probes/grad_sync_identity_probe.py: elements past 766 MiB of the slice came back different)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"file://{tempfile.mkdtemp()}/pg", rank=0, world_size=1, device_id=dev)
from simpletuner_amd import ops
for m in (64 << 20, 256 << 20, 383 << 20, 384 << 20, 512 << 20, 803155968, 1 << 30):
    g = torch.Generator(device=dev).manual_seed(1)
    seg = torch.randn(m, device=dev, generator=g, dtype=torch.float32).to(torch.bfloat16) if m <= (256 << 20) else torch.randint(-30000, 30000, (m,), device=dev, dtype=torch.int16).view(torch.bfloat16)
    seg = torch.nan_to_num(seg.float()).to(torch.bfloat16)
    want = seg.clone()
    recv = torch.empty_like(seg)
    dist.all_to_all_single(recv, seg)
    torch.cuda.synchronize()
    ok_a2a = torch.equal(recv, want)
    first_bad = None if ok_a2a else int((recv != want).nonzero()[0])
    out = torch.empty_like(seg)
    ops.sum_chunks_bf16(want, 1, out)
    torch.cuda.synchronize()
    ok_sum = torch.equal(out, want)
    sb = None if ok_sum else int((out != want).nonzero()[0])
    x = seg.clone()
    dist.all_reduce(x); dist.all_gather_into_tensor(x, x); dist.reduce_scatter_tensor(x, x)
    torch.cuda.synchronize()
    print(f"m = {m} bf16 ({m * 2 >> 20} MiB): all_to_all_single identity {ok_a2a} (first bad element {first_bad}), sum_chunks identity {ok_sum} (first bad {sb}), in-place all_reduce/all_gather/reduce_scatter identity {torch.equal(x, want)}", flush=True)
    del seg, want, recv, out, x
    torch.cuda.empty_cache()
dist.destroy_process_group()
