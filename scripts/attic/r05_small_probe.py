"""Development aid (round 5): what host-level co-scheduling buys where the launches do not fill the chip.
  (a) batch 1 (cfg0 256x512 / D 64, 544x960, KITTI): the step eagerly and from a HIP graph, each with and without the classifier
      branches on a second stream (ops.set_branch_overlap);
  (b) [removed with its switch after the measurement, profiles/r05_cosched_probe.log] batch 4 at 544x960: the deepest hourglass
      level (conv3 -> conv4 -> conv5) as two half-batch chains on two streams: 27.05 against 26.92 ms -- slower, identical results."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from densematchingbenchmark_amd import ops, synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.graph_runner import GraphedForward
from densematchingbenchmark_amd.modeling import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")


def timed(fn, n):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def build(rel):
    cfg = Config.fromfile(os.path.join(ROOT, "configs", rel))
    model = build_model(cfg, backbone=None).eval()
    synthetic.init_params_(model, seed=0, classif_gain=10.0)
    return cfg, model.to(dev)


for rel, (fh, fw) in (("PSMNet/baseline_cfg0_256x512_d64.py", (64, 128)), ("PSMNet/scene_flow.py", (136, 240)), ("PSMNet/kitti_2015.py", (96, 312))):
    cfg, model = build(rel)
    left, right = synthetic.feature_batch(0, 1, 1, 32, fh, fw, dev)
    batch = dict(leftFeature=left, rightFeature=right)
    n = 200 if fh == 64 else 40
    row = []
    with torch.no_grad():
        base = [d.clone() for d in model(batch)[0]["disps"]]
        for ovl in (False, True):
            ops.set_branch_overlap(ovl)
            for _ in range(3):
                model(batch)
            te = min(timed(lambda: model(batch), n) for _ in range(3))
            g = GraphedForward(model, track_parameters=False)
            for _ in range(3):
                out = g(batch)
            same = all(torch.equal(a, b) for a, b in zip(out[0]["disps"], base))
            tg = min(timed(lambda: g(batch), n) for _ in range(3))
            row.append("overlap %-5s eager %.3f ms  graph %.3f ms  identical %s" % (ovl, te, tg, same))
            g.reset()
        ops.set_branch_overlap(False)
    print("%-40s B=1: %s" % (rel, " | ".join(row)), flush=True)
    del model

