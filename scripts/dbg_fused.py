import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(706)
q = (torch.randn((1, 48, 136, 240), generator=g)).to(dev)
vals = ops.disp_sample_values(192, 0, 1)
up = ops.trilinear_ac(q, (192, 544, 960))
a = ops.soft_argmin(up, vals, 3.0)
b = ops.trilinear_soft_argmin(q, (192, 544, 960), vals, 3.0)
d = (a - b).abs()
print("max", d.max().item(), "mean", d.mean().item(), "n>1e-4", (d > 1e-4).sum().item())
idx = d.flatten().argmax().item()
y, x = idx // 960, idx % 960
col = up[0, :, y, x].double() * 3.0
p = torch.softmax(col, 0)
truth = (p * torch.arange(192, device=dev, dtype=torch.float64)).sum().item()
print("pixel", y, x, "unfused", a[0, 0, y, x].item(), "fused", b[0, 0, y, x].item(), "fp64 of unfused logits", truth)
ref = torch.nn.functional.interpolate(q.cpu().unsqueeze(1), [192, 544, 960], mode="trilinear", align_corners=True).squeeze(1)
print("trilinear vs torch cpu max", (up.cpu() - ref).abs().max().item())
colr = ref[0, :, y, x].double() * 3.0
print("fp64 from torch-cpu logits", (torch.softmax(colr, 0) * torch.arange(192, dtype=torch.float64)).sum().item(), "max|col diff|", (col.cpu() - colr).abs().max().item())
