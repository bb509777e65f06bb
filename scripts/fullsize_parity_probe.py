"""Development probe: per pair / level max |disp - reference golden| at the BASELINE size, with the first layer in its
2-D form and with the materialised volume."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from densematchingbenchmark_amd import ops, synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_psmnet.npz"))
dev = torch.device("cuda:0")
cfg = Config.fromfile(os.path.join(ROOT, "configs", "PSMNet", "scene_flow.py"))
model = build_model(cfg, backbone=None).eval()
synthetic.init_params_(model, seed=0, classif_gain=10.0)
model = model.to(dev)
left, right = synthetic.feature_batch(0, 1, 4, 32, 136, 240, dev)
for fused in (True, False):
    ops.set_cat_fusion(fused)
    res, _ = model(dict(leftFeature=left, rightFeature=right))
    print("first layer 2-D form" if fused else "materialised volume")
    for lvl in range(3):
        row = []
        for i in range(4):
            d = res["disps"][lvl][i:i + 1][:, :, 3::8, 5::8].cpu()
            r = torch.as_tensor(g["pair%d_disp%d" % (i, 3 - lvl)])
            row.append("%.3g/%.2g" % ((d - r).abs().max().item(), (d - r).abs().mean().item()))
        print("  level %d  max/mean per pair: %s" % (3 - lvl, "  ".join(row)))
