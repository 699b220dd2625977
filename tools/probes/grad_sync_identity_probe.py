"""Is the 1-rank exchange an identity at the REAL arena size, and does anything write a region after it was declared ready?   (r06: the SD3 full fine-tune
under ST355_BENCH_SINGLE_RANK_PG=1 + ST355_FP32_REDUCE=1 ended on another loss than the plain step.)
  python tools/probes/grad_sync_identity_probe.py [batch] [layers]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"file://{tempfile.mkdtemp()}/pg", rank=0, world_size=1, device_id=dev)
from simpletuner_amd.sd3.model import SD3
from simpletuner_amd.training.trainer import St355Accelerator, Trainer, default_config
import simpletuner_amd.training.grad_sync as GS


def build(single, fp32):
    os.environ["ST355_COMM_SINGLE_RANK"], os.environ["ST355_FP32_REDUCE"] = single, fp32
    cfg = default_config(model_family="sd3", model_type="full", train_batch_size=B, seed=42, learning_rate=1e-5, use_ema=True)
    acc = St355Accelerator(dev)
    pl = SD3(cfg, acc)
    pl.load_model(sample_size=128, num_layers=L, num_attention_heads=24, attention_head_dim=64, caption_projection_dim=1536, pooled_projection_dim=2048, pos_embed_max_size=192)
    pl.enable_full_finetune()
    tr = Trainer(cfg, pl, acc)
    return pl, tr


def batches():
    g = torch.Generator(device=dev).manual_seed(42)
    return [{"latent_batch": torch.randn(B, 16, 128, 128, device=dev, generator=g).to(torch.bfloat16), "prompt_embeds": torch.randn(B, 231, 4096, device=dev, generator=g).to(torch.bfloat16),
             "add_text_embeds": torch.randn(B, 2048, device=dev, generator=g).to(torch.bfloat16)} for _ in range(2)]


def run(single, fp32, mode):
    """mode: plain | sync_each (device-synchronise around every slice's collectives and compare the slice before / after) | spy (snapshot at ready(), compare at finish())"""
    torch.manual_seed(42)
    pl, tr = build(single, fp32)
    comp = pl.get_trained_component()
    gs = getattr(comp, "grad_sync", None)
    log = []
    if gs is not None and mode == "sync_each":
        inner = gs._fire_on_comm_stream
        def fire(lo, hi, W):
            torch.cuda.synchronize()
            before = gs.flat[lo:hi].clone()
            torch.cuda.synchronize()
            n0 = len(gs.launched_ops)
            inner(lo, hi, W)
            torch.cuda.synchronize()
            same = torch.equal(before, gs.flat[lo:hi])
            log.append((lo, hi, same, [k for k, _, _ in gs.launched_ops[n0:]]))
            if not same:
                d = (before != gs.flat[lo:hi]).nonzero().flatten()
                print(f"   slice [{lo}, {hi}) NOT identity under full synchronisation: {d.numel()} elements differ, first {lo + int(d[0])}, last {lo + int(d[-1])}", flush=True)
        gs._fire_on_comm_stream = fire
    if gs is not None and mode == "spy":
        snaps = []
        r0, f0 = gs.ready, gs.finish
        def ready(lo, hi):
            torch.cuda.synchronize()
            snaps.append((lo, hi, gs.flat[lo:hi].clone(), gs.flat.data_ptr()))
            r0(lo, hi)
        def finish():
            flat = gs.flat
            s = f0()
            torch.cuda.synchronize()
            for lo, hi, snap, ptr in snaps:
                if ptr != flat.data_ptr() or not torch.equal(snap, flat[lo:hi]):
                    d = (snap != flat[lo:hi]).nonzero().flatten()
                    print(f"   region [{lo}, {hi}) differs from its value at ready(): {d.numel()} elements (arena switched: {ptr != flat.data_ptr()})", flush=True)
            log.append(len(snaps)); snaps.clear()
            return s
        gs.ready, gs.finish = ready, finish
    bs = batches()
    losses = [float(tr.train_step(dict(bs[i % 2]))) for i in range(4)]
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in tr.params[:50]]).float().cpu()
    sl = [] if gs is None else [(lo, hi) for lo, hi in gs.launched_slices]
    print(f"single={single} fp32={fp32} mode={mode}: losses {losses}; slices of the last step {len(sl)}: {[(hi - lo) * 2 >> 20 for lo, hi in sl]} MiB", flush=True)
    del tr, pl, comp, gs
    import gc; gc.collect(); torch.cuda.empty_cache()
    return losses, flat


base = run("0", "0", "plain")
for single, fp32, mode in (("1", "0", "plain"), ("1", "1", "plain"), ("1", "1", "sync_each"), ("1", "1", "spy")):
    r = run(single, fp32, mode)
    print(f"   -> losses equal to the plain step: {r[0] == base[0]}, first 50 parameter tensors bit-equal: {bool(torch.equal(r[1], base[1]))}", flush=True)
dist.destroy_process_group()
