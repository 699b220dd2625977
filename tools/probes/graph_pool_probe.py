"""How much memory does each captured aspect bucket add?  SD3-Medium full fine-tune, hip_graph on, one train step per bucket shape; prints allocated / reserved
bytes after every first encounter (= 2 eager warm-up steps + capture + replay).  Usage: python tools/probes/graph_pool_probe.py [batch] [layers]"""
import gc
import sys

import torch

sys.path.insert(0, ".")
from simpletuner_amd.sd3.model import SD3  # noqa: E402
from simpletuner_amd.training.trainer import St355Accelerator, Trainer, default_config  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
cfg = default_config(model_family="sd3", model_type="full", train_batch_size=B, seed=1, learning_rate=1e-5, use_ema=True, hip_graph=True)
acc = St355Accelerator(dev)
plugin = SD3(cfg, acc)
plugin.load_model(sample_size=128, num_layers=L, num_attention_heads=24, attention_head_dim=64, caption_projection_dim=1536, pooled_projection_dim=2048, pos_embed_max_size=192)
plugin.enable_full_finetune()
trainer = Trainer(cfg, plugin, acc)
gib = lambda x: round(x / 2 ** 30, 1)
print(f"model built: allocated {gib(torch.cuda.memory_allocated())} GiB reserved {gib(torch.cuda.memory_reserved())} GiB", flush=True)
g = torch.Generator(device=dev).manual_seed(0)
for (h, w) in ((128, 128), (96, 168), (168, 96), (112, 144), (144, 112)):
    batch = {"latent_batch": torch.randn(B, 16, h, w, device=dev, generator=g).to(torch.bfloat16),
             "prompt_embeds": torch.randn(B, 231, 4096, device=dev, generator=g).to(torch.bfloat16),
             "add_text_embeds": torch.randn(B, 2048, device=dev, generator=g).to(torch.bfloat16)}
    trainer.train_step(dict(batch))
    torch.cuda.synchronize()
    gc.collect()
    st = torch.cuda.memory_stats()
    print(f"bucket {h}x{w}: allocated {gib(torch.cuda.memory_allocated())} GiB reserved {gib(torch.cuda.memory_reserved())} GiB  active {gib(st['active_bytes.all.current'])} "
          f"inactive_split {gib(st['inactive_split_bytes.all.current'])} peak {gib(torch.cuda.max_memory_allocated())}", flush=True)
    trainer.train_step(dict(batch))        # a replay
torch.cuda.synchronize()
print("ok")
