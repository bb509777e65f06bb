"""Development aid (round 4): one step of a configuration eagerly against the same step captured in a HIP graph and replayed
(the library is capture-safe since round 4): what the host-side launch path costs.  GR_CONFIG / GR_B / GR_SHAPE (feature H,W)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from densematchingbenchmark_amd import synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
for rel, B, (fh, fw) in (("StereoNet/scene_flow_8x_2stage.py", 8, (48, 156)), ("PSMNet/scene_flow.py", 4, (136, 240)),
                         ("PSMNet/scene_flow.py", 1, (136, 240))):
    cfg = Config.fromfile(os.path.join(ROOT, "configs", rel))
    model = build_model(cfg, backbone=None).eval()
    synthetic.init_params_(model, seed=0, classif_gain=10.0)
    model = model.to(dev)
    left, right = synthetic.feature_batch(0, 1, B, 32, fh, fw, dev)
    batch = dict(leftFeature=left, rightFeature=right)

    def run(fn, n):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n

    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                model(batch)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = model(batch)[0]["disps"]
        n = 200 if "StereoNet" in rel else 20
        te = min(run(lambda: model(batch), n) for _ in range(3))
        tg = min(run(g.replay, n) for _ in range(3))
    print("%-36s batch %d: eager %.3f ms/step (%.1f pairs/s), graph replay %.3f ms/step (%.1f pairs/s)" % (rel, B, te, B / te * 1e3, tg, B / tg * 1e3), flush=True)
    del model, g, out
    torch.cuda.empty_cache()
