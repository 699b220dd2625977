import torch, sys
sys.path.insert(0, "/root/repo")
from simpletuner_amd import ops
G = torch.load("/root/repo/tests/golden/fp8_vectors.pt")
x2 = G["x"].reshape(-1, 384).cuda().contiguous()
xq, sa = ops.fp8_quantize_act(x2)
d = (xq.cpu() != G["x_q"])
idx = d.flatten().nonzero().flatten()[:12]
isc = (57344.0 / x2.abs().amax().clamp(min=1e-12)).clamp(max=57344.0)
scaled = (x2 * isc).flatten().cpu()
print("isc", isc.item())
for i in idx.tolist():
    s = scaled[i].float().item()
    print(i, "x", x2.flatten()[i].item(), "scaled", s, hex(scaled[i].view(torch.int16).item() & 0xffff), "got", xq.flatten()[i].item(), "exp", G["x_q"].flatten()[i].item(),
          "torch-gpu", (x2 * isc).clamp(-57344, 57344).to(torch.float8_e5m2).flatten()[i].view(torch.uint8).item())
tg = (x2 * isc).clamp(-57344, 57344).to(torch.float8_e5m2).view(torch.uint8)
print("kernel vs torch-on-gpu mismatches:", (tg != xq).sum().item(), " torch-gpu vs golden(cpu):", (tg.cpu() != G["x_q"]).sum().item())
print("kernel scale_a", sa.item(), "golden", G["scale_a"][0].item(), "1/7680 bf16", torch.tensor(1/7680.).to(torch.bfloat16).item())
print("amax torch", x2.abs().amax().item())
