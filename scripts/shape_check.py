import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import ops, synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
dev = torch.device("cuda:0")
cfg = Config.fromfile(os.path.join(ROOT, "configs", "PSMNet", "scene_flow.py"))
model = build_model(cfg, backbone=None).eval(); synthetic.init_params_(model, seed=0, classif_gain=10.0); model = model.to(dev)
for (fh, fw, B) in ((96, 312, 4), (64, 128, 4), (136, 244, 1)):
    left, right = synthetic.feature_batch(0, 1, B, 32, fh, fw, dev)
    outs = {}
    for fused in (True, False):
        ops.set_cat_fusion(fused)
        with torch.no_grad():
            for _ in range(3): r, _ = model(dict(leftFeature=left, rightFeature=right))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): r, _ = model(dict(leftFeature=left, rightFeature=right))
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        outs[fused] = [d.clone() for d in r["disps"]]
        print("features %dx%d B=%d  2-D first layer %s: %.2f ms/step %.1f pairs/s" % (fh, fw, B, fused, dt * 1e3, B / dt), flush=True)
    print("   max |disp fused - materialised| =", max((a - b).abs().max().item() for a, b in zip(outs[True], outs[False])))
