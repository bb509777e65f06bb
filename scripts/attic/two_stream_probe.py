"""Development probe: does running the batch as S independent sub-batches on S HIP streams fill the tail rounds and the
under-filled quarter-resolution launches?  Pairs are independent in eval mode, so results are identical.
    python scripts/attic/two_stream_probe.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from densematchingbenchmark_amd import synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = Config.fromfile(os.path.join(ROOT, "configs", "PSMNet", "scene_flow.py"))
model = build_model(cfg, backbone=None).eval()
synthetic.init_params_(model, seed=0, classif_gain=10.0)
model = model.to(dev)
B = 4
left, right = synthetic.feature_batch(0, 1, B, 32, 136, 240, dev)


def run(nstreams):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    per = B // nstreams
    parts = [dict(leftFeature=left[i * per:(i + 1) * per].contiguous(), rightFeature=right[i * per:(i + 1) * per].contiguous())
             for i in range(nstreams)]

    def step():
        outs = []
        cur = torch.cuda.current_stream(dev)
        for s, p in zip(streams, parts):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                res, _ = model(p)
                outs.append(res["disps"])
        for s in streams:
            cur.wait_stream(s)
        return outs

    with torch.no_grad():
        for _ in range(3):
            outs = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            outs = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    return dt, torch.cat([o[0] for o in outs])


base = None
for n in (1, 2, 4, 1, 2):
    dt, d = run(n)
    if base is None:
        base = d
    print("streams=%d  %.3f ms/step  %.2f pairs/s  max|d - d(1 stream)| = %g" % (n, dt * 1e3, B / dt, (d - base).abs().max().item()), flush=True)
