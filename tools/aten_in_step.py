#!/usr/bin/env python3
"""Which python lines of the Flux train step still launch ATen / runtime kernels (copies, fills, elementwise)?   

    python tools/aten_in_step.py [double single batch]      default 4 8 8 -> gpurun_out/aten_in_step.txt

One profiled step (torch.profiler, CPU + device activities, with_stack): every operator that is not a libst355 launch is listed with its device time,
call count and the innermost frame under simpletuner_amd/ that issued it."""
import os
import sys
from collections import defaultdict
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    nd, ns, B = (int(a) for a in (sys.argv[1:4] + ["4", "8", "8"][len(sys.argv) - 1:]))
    from simpletuner_amd.flux.model import Flux
    from simpletuner_amd.training.trainer import St355Accelerator, Trainer, default_config
    dev = torch.device("cuda:0")
    cfg = default_config(model_family="flux", lora_rank=32, train_batch_size=B, seed=42, lora_init_b_std=1e-3, model_type="lora", learning_rate=1e-4)
    plugin = Flux(cfg, St355Accelerator(dev))
    plugin.load_model(num_layers=nd, num_single_layers=ns, guidance_embeds=True)
    plugin.add_lora_adapter()
    trainer = Trainer(cfg, plugin, plugin.accelerator)
    gen = torch.Generator(device=dev).manual_seed(1)
    batch = {"latent_batch": torch.randn(B, 16, 128, 128, device=dev, generator=gen).to(torch.bfloat16),
             "prompt_embeds": torch.randn(B, 512, 4096, device=dev, generator=gen).to(torch.bfloat16),
             "add_text_embeds": torch.randn(B, 768, device=dev, generator=gen).to(torch.bfloat16)}
    for _ in range(2):
        trainer.train_step(dict(batch))
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        trainer.train_step(dict(batch))
        torch.cuda.synchronize()
    agg = defaultdict(lambda: [0.0, 0, set()])
    for ev in prof.events():
        dt = getattr(ev, "device_time_total", None)
        if dt is None:
            dt = getattr(ev, "cuda_time_total", 0)
        if not dt or not ev.name.startswith("aten::"):
            continue
        if ev.cpu_children and any(c.name.startswith("aten::") and (getattr(c, "device_time_total", 0) or getattr(c, "cuda_time_total", 0)) for c in ev.cpu_children):
            continue            # count the leaf operator only
        frame = next((f for f in (ev.stack or []) if "simpletuner_amd" in f), (ev.stack or ["?"])[0] if ev.stack else "?")
        k = (ev.name, frame.strip())
        agg[k][0] += dt
        agg[k][1] += 1
        agg[k][2].add(str(ev.input_shapes)[:80])
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    out = [f"Flux {nd} double + {ns} single blocks, batch {B}: ATen operators with device time in ONE train step (leaf operators, by issuing line)",
           f"total {sum(v[0] for v in agg.values()) / 1e3:.2f} ms in {sum(v[1] for v in agg.values())} calls", ""]
    for (name, frame), (us, n, shapes) in rows[:60]:
        out.append(f"{us / 1e3:8.3f} ms  {n:5d} x  {name:28s} {frame[:150]}   {sorted(shapes)[0] if shapes else ''}")
    text = "\n".join(out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "aten_in_step.txt"), "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
