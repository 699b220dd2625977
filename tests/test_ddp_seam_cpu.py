"""The gradient-synchronisation seam under the reference Trainer's own calls, world_size 2 over gloo on CPU:

  * `accelerator.prepare(model)` -> torch DistributedDataParallel (helpers/training/trainer.py:4564-4571) + `install_ddp_comm_hook`, and the reducer-free
    `St355DistributedDataParallel` wrapper: after every synchronised backward `param.grad` is the MEAN over ranks of the accumulated local gradients,
    with `no_sync()` micro-steps in between (`accelerator.accumulate`, trainer.py:7009) — DDP's contract, checked against plain autograd + a manual mean;
  * `set_prepared_model` (trainer.py:4577) recognises the DDP wrapper and installs the hook itself; `unwrap_model` strips it;
  * the fp32-accumulating reduce-scatter form (all-to-all + local fp32 sum + all-gather) of a bf16 arena equals the exactly-rounded fp32 sum.

The component is a stand-in with the seam attributes of a real st355 component (flat parameter / gradient arenas, ONE autograd.Function whose backward fills
the arena back to front through `GradSync.ready` and ends with `hand_over_gradients`): the HIP models need a GPU, the seam logic does not."""
import os
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _ToyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, w1, w2):
        ctx.model = model
        ctx.save_for_backward(x, w1.detach(), w2.detach())
        return torch.tanh(x @ w1.detach().t()) @ w2.detach()

    @staticmethod
    def backward(ctx, dy):
        from simpletuner_amd.training.grad_sync import hand_over_gradients
        m = ctx.model
        x, w1, w2 = ctx.saved_tensors
        h = torch.tanh(x @ w1.t())
        gs = m.grad_sync
        if gs is not None:
            gs.begin()
        n1 = w1.numel()
        m.lora_grad_flat[n1:n1 + w2.numel()].copy_((h.t() @ dy).reshape(-1))                   # the LAST parameter's gradient first (back to front)
        if gs is not None:
            gs.ready(n1, n1 + w2.numel())
        dh = (dy[:, None] * w2[None, :]) * (1 - h * h)
        m.lora_grad_flat[:n1].copy_((dh.t() @ x).reshape(-1))
        if gs is not None:
            gs.ready(0, n1)
            m.grad_scale_from_sync = gs.finish()
        gflat = hand_over_gradients(m, m.lora_grad_flat)
        return None, None, gflat[:n1].view_as(w1), gflat[n1:n1 + w2.numel()].view_as(w2)


class ToyComponent(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.lora_flat = torch.randn(6 * 5 + 6, generator=g) * 0.5
        self.lora_grad_flat = torch.zeros_like(self.lora_flat)
        self.w1 = torch.nn.Parameter(self.lora_flat[:30].view(6, 5))
        self.w2 = torch.nn.Parameter(self.lora_flat[30:].view(6))
        self._lora_params = [self.w1, self.w2]
        self.grad_sync = None

    def trainable_parameters(self):
        return list(self._lora_params)

    def forward(self, x):
        return _ToyFn.apply(self, x, self.w1, self.w2)


def _plain_grads(comp, xs):
    """sum over the micro-batches of the local gradients, by plain autograd on detached copies"""
    w1, w2 = comp.w1.detach().clone().requires_grad_(True), comp.w2.detach().clone().requires_grad_(True)
    for x in xs:
        (torch.tanh(x @ w1.t()) @ w2).sum().backward()
    return torch.cat([w1.grad.reshape(-1), w2.grad.reshape(-1)])


def _mean_over_ranks(t):
    t = t.clone()
    dist.all_reduce(t)
    return t / dist.get_world_size()


def _run_steps(wrapper, comp, rank, tag, res):
    g = torch.Generator().manual_seed(50 + rank)
    # (a) every backward synchronised
    x = torch.randn(7, 5, generator=g)
    for p in comp.parameters():
        p.grad = None
    wrapper(x).sum().backward()
    got = torch.cat([comp.w1.grad.reshape(-1), comp.w2.grad.reshape(-1)])
    res[tag + "_sync"] = torch.allclose(got, _mean_over_ranks(_plain_grads(comp, [x])), atol=1e-6)
    res[tag + "_sync_overlapped"] = len(comp.grad_sync.launched_ops) > 0          # the exchange ran INSIDE the backward (GradSync), not after it
    # (b) accelerator.accumulate: two no_sync micro-steps, then the boundary step
    xs = [torch.randn(7, 5, generator=g) for _ in range(3)]
    for p in comp.parameters():
        p.grad = None
    for x in xs[:2]:
        with wrapper.no_sync():
            wrapper(x).sum().backward()
    local_only = torch.cat([comp.w1.grad.reshape(-1), comp.w2.grad.reshape(-1)]).clone()
    res[tag + "_nosync_local"] = torch.allclose(local_only, _plain_grads(comp, xs[:2]), atol=1e-6)      # nothing exchanged yet
    wrapper(xs[2]).sum().backward()
    got = torch.cat([comp.w1.grad.reshape(-1), comp.w2.grad.reshape(-1)])
    res[tag + "_accum"] = torch.allclose(got, _mean_over_ranks(_plain_grads(comp, xs)), atol=1e-6)
    # (c) and a plain synchronised step right after (the accumulation state is cleared)
    x = torch.randn(7, 5, generator=g)
    for p in comp.parameters():
        p.grad = None
    wrapper(x).sum().backward()
    got = torch.cat([comp.w1.grad.reshape(-1), comp.w2.grad.reshape(-1)])
    res[tag + "_after"] = torch.allclose(got, _mean_over_ranks(_plain_grads(comp, [x])), atol=1e-6)


def _worker(rank, world, init_file, out_dir):
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    from simpletuner_amd.foundation import ModelFoundation
    from simpletuner_amd.training.ddp_seam import St355DistributedDataParallel, capture_safe, install_ddp_comm_hook
    from simpletuner_amd.training.grad_sync import GradSync
    res = {}
    # 1. the reducer-free wrapper (replicas deliberately start apart: construction broadcasts rank 0's state, like DDP)
    comp = ToyComponent(seed=10 + rank)
    wrapper = St355DistributedDataParallel(comp)
    ref = ToyComponent(seed=10)
    res["start_state"] = torch.equal(comp.lora_flat, ref.lora_flat)
    res["attrs"] = wrapper.module is comp and wrapper.require_backward_grad_sync is True and list(wrapper.state_dict()) == ["module.w1", "module.w2"]
    _run_steps(wrapper, comp, rank, "wrapper", res)
    # 2. torch's own DDP, as accelerator.prepare builds it, + the st355 communication hook
    comp2 = ToyComponent(seed=20 + rank)
    ddp = torch.nn.parallel.DistributedDataParallel(comp2)
    install_ddp_comm_hook(ddp)
    _run_steps(ddp, comp2, rank, "ddp_hook", res)
    # 3. set_prepared_model installs the hook itself; unwrap_model strips the wrapper
    comp3 = ToyComponent(seed=30 + rank)
    ddp3 = torch.nn.parallel.DistributedDataParallel(comp3)
    plug = ModelFoundation.__new__(ModelFoundation)
    plug.set_prepared_model(ddp3)
    res["set_prepared"] = getattr(ddp3, "_st355_seam", None) is not None and plug.model is ddp3 and ModelFoundation.unwrap_model(ddp3) is comp3
    _run_steps(ddp3, comp3, rank, "prepared", res)
    res["capture_safe_gloo"] = capture_safe()
    # 4. fp32-accumulating reduce-scatter of a bf16 arena (all-to-all + local fp32 sum + all-gather)
    n = 4096 + 6                                                       # a tail that is no multiple of the world size: goes through the all-reduce form
    gen = torch.Generator().manual_seed(70 + rank)
    mine = torch.randn(n, generator=gen).to(torch.bfloat16)
    both = [torch.randn(n, generator=torch.Generator().manual_seed(70 + r)).to(torch.bfloat16) for r in range(world)]
    flat = mine.clone()
    assert not GradSync(flat, bucket_bytes=2 * 1024, mode="rs_ag").fp32_reduce       # opt-in until the sequence has run over RCCL with > 1 rank
    gs = GradSync(flat, bucket_bytes=2 * 1024, mode="rs_ag", fp32_reduce=True)
    assert gs.fp32_reduce
    gs.begin()
    for hi in range(n, 0, -1000):
        gs.ready(max(0, hi - 1000), hi)
    gs.finish()
    exact = (both[0].float() + both[1].float()).to(torch.bfloat16)
    res["fp32_reduce_exact"] = torch.equal(flat, exact)
    res["fp32_reduce_ops"] = sorted({op for op, _, _ in gs.launched_ops})
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_fp32_reduce_span_keeps_the_sum_chunks_alignment_contract():
    """st355_sum_chunks_bf16 wants n % 8 == 0 per chunk and a 16-byte-aligned shard: GradSync.fp32_span must only hand it such spans, for every ragged slice
    and world size (the UNet arenas hold 8-element tensors, e.g. the padded conv_out bias: a span rounded to W only broke this at W = 2, 8, 16 ...)"""
    from simpletuner_amd.training.grad_sync import GradSync
    for W in (2, 3, 4, 8, 16):
        for lo in (0, 8, 64, 72, 4096 + 8):
            for length in (8, 24, 8 * W, 8 * W + 8, 1000, 4096 + 8, 65536 + 72, 12345):
                m = GradSync.fp32_span(lo, lo + length, W)
                assert 0 <= m <= length and m % (8 * W) == 0 and (m // W) % 8 == 0
                assert length - m < 8 * W or m == 0                     # the all-reduce tail stays below 8 W elements
                for r in range(W):
                    assert ((lo + r * (m // W)) * 2) % 16 == 0          # every rank's shard starts on a 16-byte boundary
        assert GradSync.fp32_span(4, 4 + 8 * W * 3, W) == 0             # a span that does not start 8-aligned takes RCCL's own reduce-scatter


def test_ddp_seam_world2_gloo():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "init"), d), nprocs=2, join=True)
        results = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(2)]
    for r, res in enumerate(results):
        for k, v in res.items():
            if k == "capture_safe_gloo":
                assert v is False
            elif k == "fp32_reduce_ops":
                assert "all_to_all" in v and "all_gather" in v, v
            else:
                assert v is True, f"rank {r}: {k} = {v}"
        # the DDP-hook path exchanges inside the backward on plain steps too
        assert res["ddp_hook_sync_overlapped"] and res["wrapper_sync_overlapped"]
