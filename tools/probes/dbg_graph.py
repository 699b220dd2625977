import sys, torch
sys.path.insert(0, '.')
from simpletuner_amd.training.trainer import St355Accelerator, Trainer, default_config
from simpletuner_amd.sdxl.model import SDXL
dev = torch.device("cuda", 0)
SMALL = dict(block_out_channels=(64, 128), layers_per_block=1, down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"),
             up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"), transformer_layers_per_block=(1, 2), attention_head_dim=(1, 2), cross_attention_dim=128,
             projection_class_embeddings_input_dim=64 + 6 * 64, addition_time_embed_dim=64)
import os, gc
if os.environ.get("BIG"):
    SMALL = {}
NB = int(os.environ.get('NB', '2'))
LAT = 128 if os.environ.get("BIG") else 16
CTX = (77, 2048, 1280) if os.environ.get("BIG") else (9, 128, 64)
def run(graph):
    torch.manual_seed(0)
    cfg = default_config(model_family="sdxl", model_type="full", train_batch_size=NB, learning_rate=1e-4, hip_graph=graph)
    acc = St355Accelerator(dev)
    pl = SDXL(cfg, acc); pl.load_model(**SMALL); pl.enable_full_finetune()
    tr = Trainer(cfg, pl, acc)
    g = torch.Generator(device=dev).manual_seed(1)
    out = []
    for i in range(int(os.environ.get('STEPS', '6'))):
        b = {"latent_batch": torch.randn(NB, 4, LAT, LAT, device=dev, generator=g).to(torch.bfloat16),
             "prompt_embeds": torch.randn(NB, CTX[0], CTX[1], device=dev, generator=g).to(torch.bfloat16),
             "add_text_embeds": torch.randn(NB, CTX[2], device=dev, generator=g).to(torch.bfloat16),
             "batch_time_ids": torch.tensor([[128., 128, 0, 0, 128, 128]] * NB, device=dev, dtype=torch.bfloat16),
             }
        if not os.environ.get("SAMPLED"):
            b["timesteps"] = torch.tensor([100 + i, 700 - i, 300, 900][:NB]); b["noise"] = torch.randn(NB, 4, LAT, LAT, device=dev, generator=g).to(torch.bfloat16)
        if os.environ.get("DEVSYNC") and i == 3:
            torch.cuda.synchronize(); print("device sync", flush=True)
        l = tr.train_step(b)
        out.append(float(l)); comp = pl.get_trained_component()
        if os.environ.get("NOCHECK"):
            print(i, out[-1], flush=True); continue
        print(i, out[-1], "grad nan:", bool(torch.isnan(comp._last_grad_flat.float()).any()) if comp._last_grad_flat is not None else None, "w nan:", bool(torch.isnan(comp.arena.float()).any()), flush=True)
    del tr, pl; gc.collect(); torch.cuda.empty_cache()
    return out
import sys as _s
if "eager" in _s.argv: print("eager", run(False), flush=True)
if "graph" in _s.argv: print("graph", run(True), flush=True)
