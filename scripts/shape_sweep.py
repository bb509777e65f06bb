"""Development aid: the PSMNet cost path (bench.py's step) over image shapes other than the two operating points the kernels
were tuned on -- time per step, pairs/s, and the time per million quarter-resolution voxels relative to 544x960 (the arithmetic
of the path is proportional to D/4 x H/4 x W/4, so that ratio is the shape's efficiency against the headline shape).
    python scripts/shape_sweep.py            (SWEEP_B = pairs per step, default 4)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model

dev = torch.device("cuda:0")
B = int(os.environ.get("SWEEP_B", "4"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = Config.fromfile(os.path.join(ROOT, "configs", "PSMNet", "scene_flow.py"))
model = build_model(cfg, backbone=None).eval()
synthetic.init_params_(model, seed=0, classif_gain=10.0)
model = model.to(dev)
SHAPES = [(544, 960), (384, 1248), (256, 512), (320, 960), (368, 1232), (480, 640), (512, 1024), (576, 1024), (720, 1280), (1088, 1920)]


def run(batch, n):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        model(batch)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


print("# PSMNet cost path, max_disp 192, %d pairs per step, FP32 exact mode; 'rel' = time per quarter-resolution voxel relative to 544x960" % B)
base = None
with torch.no_grad():
    for H, W in SHAPES:
        b = B if H * W <= 1024 * 1280 else max(1, B // 4)
        left, right = synthetic.feature_batch(0, 1, b, 32, H // 4, W // 4, dev)
        batch = dict(leftFeature=left, rightFeature=right)
        run(batch, 3)
        ms = min(run(batch, 8) for _ in range(3))
        per_vox = ms / (b * 48 * (H // 4) * (W // 4))
        if base is None:
            base = per_vox
        print("%4d x %4d  features %3d x %3d  pairs %d  %8.3f ms/step  %8.1f pairs/s  rel %.3f" %
              (H, W, H // 4, W // 4, b, ms, b / ms * 1e3, per_vox / base), flush=True)
        del left, right, batch
        torch.cuda.empty_cache()
