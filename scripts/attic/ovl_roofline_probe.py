import os, sys
sys.path.insert(0, "/root/repo")
import torch
from densematchingbenchmark_amd import ops, synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model
dev = torch.device("cuda:0")
cfg = Config.fromfile("/root/repo/configs/PSMNet/scene_flow.py")
model = build_model(cfg, backbone=None).eval(); synthetic.init_params_(model, seed=0, classif_gain=10.0); model = model.to(dev)
l, r = synthetic.feature_batch(0, 1, 4, 32, 136, 240, dev)
batch = dict(leftFeature=l, rightFeature=r)
flop = 2.0 * 27 * 32 * 32 * 4 * 48 * 136 * 240
for rep in range(2):
    for ovl in (False, True):
        ops.set_branch_overlap(ovl)
        with torch.no_grad():
            for _ in range(3): model(batch)
            t = ops.KernelTimer(["conv3d_k3_s1_32to32"]); ops.set_kernel_timer(t)
            torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
            for _ in range(10): model(batch)
            e.record(); torch.cuda.synchronize(); ops.set_kernel_timer(None)
        ev = t.events["conv3d_k3_s1_32to32"]
        ms = [a.elapsed_time(b) for a, b in ev]
        per = [sum(ms[i::6]) / len(ms[i::6]) for i in range(6)]
        print("overlap", ovl, "step %.3f ms" % (s.elapsed_time(e) / 10), "dominant mean %.4f ms -> %.4f of peak; per position in the step:" % (sum(ms)/len(ms), flop / (sum(ms)/len(ms)) / 1e9 / 157.3), ["%.3f" % v for v in per])
ops.set_branch_overlap(False)
