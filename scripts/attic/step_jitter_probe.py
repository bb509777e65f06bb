"""Development aid: per-step times of the PSMNet cost path over a few hundred back-to-back steps (one HIP event per step), to see
whether single steps are stretched by something outside the kernels (clock / power management, another client of the GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densematchingbenchmark_amd import synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model

dev = torch.device("cuda:0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(ROOT, "configs", *os.environ.get("AB_CONFIG", "PSMNet/scene_flow.py").split("/")))
model = build_model(cfg, backbone=None).eval()
synthetic.init_params_(model, seed=0, classif_gain=10.0)
model = model.to(dev)
Hp, Wp = cfg.data.eval.input_shape
left, right = synthetic.feature_batch(0, 1, 4, 32, Hp // 4, Wp // 4, dev)
batch = dict(leftFeature=left, rightFeature=right)
N = int(os.environ.get("JITTER_N", "300"))
with torch.no_grad():
    for _ in range(5):
        model(batch)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        model(batch)
        ev[i + 1].record()
    torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
s = sorted(ts)
print("steps %d: min %.3f  median %.3f  mean %.3f  p95 %.3f  max %.3f ms" % (N, s[0], s[N // 2], sum(ts) / N, s[int(N * 0.95)], s[-1]))
t = 0.0
for i, d in enumerate(ts):
    if d > 1.05 * s[N // 2]:
        print("  step %3d at %7.1f ms: %.3f ms" % (i, t, d))
    t += d
