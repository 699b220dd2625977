"""Probe (GPU box): can TWO ranks of one RCCL communicator share ONE device?    Each rank binds cuda:0 and runs the collectives
GradSync issues (all_reduce, reduce_scatter_tensor, all_gather_into_tensor, all_to_all_single) on bf16 / fp32 buffers.  Prints one verdict line per rank;
run under `timeout` (a refused duplicate device can also show up as a hang in communicator setup)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        x = torch.full((1 << 20,), float(rank + 1), device="cuda")
        dist.all_reduce(x)
        torch.cuda.synchronize()
        ok = bool((x == 3.0).all())
        y = torch.arange(4096, device="cuda", dtype=torch.float32).to(torch.bfloat16)
        out = torch.empty(2048, device="cuda", dtype=torch.bfloat16)
        dist.reduce_scatter_tensor(out, y)
        g = torch.empty(4096, device="cuda", dtype=torch.bfloat16)
        dist.all_gather_into_tensor(g, out)
        a = torch.empty_like(y)
        dist.all_to_all_single(a, y)
        torch.cuda.synchronize()
        print(f"[rccl-shared-gpu] rank {rank}: collectives completed, all_reduce correct = {ok}", flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print(f"[rccl-shared-gpu] rank {rank}: REFUSED / failed: {type(e).__name__}: {str(e)[:600]}", flush=True)
        sys.exit(3)


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29611), nprocs=2, join=True)
