import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import ops, synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
dev = torch.device("cuda:0")
for cfgrel, B, C, fh, fw in (("StereoNet/scene_flow_8x_2stage.py", 8, 32, 48, 156), ("AcfNet/scene_flow_adaptive.py", 4, 32, 136, 240)):
    cfg = Config.fromfile(os.path.join(ROOT, "configs", cfgrel))
    model = build_model(cfg, backbone=None).eval()
    synthetic.init_params_(model, seed=0, classif_gain=10.0)
    model = model.to(dev)
    left, right = synthetic.feature_batch(0, 1, B, C, fh, fw, dev)
    for fused in (True, False, True, False):
        ops.set_cat_fusion(fused)
        with torch.no_grad():
            for _ in range(5):
                model(dict(leftFeature=left, rightFeature=right))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 30 if B == 8 else 8
            for _ in range(n):
                model(dict(leftFeature=left, rightFeature=right))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        print("%s  2-D first layer %s: %.3f ms/step  %.1f pairs/s" % (cfgrel, fused, dt * 1e3, B / dt), flush=True)
    del model
