"""Development probe: classifier branches on a second stream next to the following hourglass (ops.set_branch_overlap).
    python scripts/overlap_probe.py [config] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from densematchingbenchmark_amd import ops, synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
cfg_rel = sys.argv[1] if len(sys.argv) > 1 else "PSMNet/scene_flow.py"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = Config.fromfile(os.path.join(ROOT, "configs", cfg_rel))
model = build_model(cfg, backbone=None).eval()
synthetic.init_params_(model, seed=0, classif_gain=10.0)
model = model.to(dev)
left, right = synthetic.feature_batch(0, 1, 4, 32, 136, 240, dev)
batch = dict(leftFeature=left, rightFeature=right)


def run(flag, n):
    ops.set_branch_overlap(flag)
    for _ in range(4):
        res, _ = model(batch)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        res, _ = model(batch)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n, res


for rep in range(2):
    t0, r0 = run(False, steps)
    t1, r1 = run(True, steps)
    same = all(torch.equal(a, b) for k in r0 for a, b in zip(r0[k], r1[k]))
    print("sequential %.3f ms   overlapped %.3f ms   (%+.2f%%)   results %s" % (t0, t1, (t1 / t0 - 1) * 100, "identical" if same else "DIFFERENT"), flush=True)
ops.set_branch_overlap(False)
