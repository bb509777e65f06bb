"""Development aid (round 4): only the memory-bound TAIL of each classifier branch -- the 32 -> 1 head and the up-sampling +
regression of level k -- on a second stream under the matrix-bound 32 -> 32 convolution of level k + 1 (the opt-in branch overlap of
ops.set_branch_overlap moves the whole branch next to the following hourglass).  Same kernels, same operands; prints the step
time of the sequential form, of this form and of the opt-in form, alternated in one process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

RESERVE_KB = int(os.environ.get("TAIL_RESERVE_KB", "0"))   # needs DMB_LIB=dev
from densematchingbenchmark_amd import _lib, ops, synthetic
from densematchingbenchmark_amd.config import Config
from densematchingbenchmark_amd.modeling import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
cfg = Config.fromfile(os.path.join(ROOT, "configs", "PSMNet", "scene_flow.py"))
model = build_model(cfg, backbone=None).eval()
synthetic.init_params_(model, seed=0, classif_gain=10.0)
model = model.to(dev)
left, right = synthetic.feature_batch(0, 1, 4, 32, 136, 240, dev)
batch = dict(leftFeature=left, rightFeature=right)
agg = model.cost_processor.aggregator
vals = ops.disp_sample_values(192, 0, 1)


def tail_overlapped():
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.cat_fms import LazyCatVolume
    raw = LazyCatVolume(left, right, kind="cat", **model.cost_processor.default_args)   # as the cost processor hands it over in eval mode
    B, C, D, H, W = raw.shape
    size = (192, H * 4, W * 4)
    main, side = torch.cuda.current_stream(dev), ops.side_stream(dev)
    cost0 = agg.dres0(raw)
    cost0 = agg.dres1[1](agg.dres1[0](cost0), residual=cost0)
    out1, pre1, post1 = agg.dres2(cost0, None, None, skip=cost0)
    out2, pre2, post2 = agg.dres3(out1, pre1, post1, skip=cost0)
    out3, pre3, post3 = agg.dres4(out2, pre2, post2, skip=cost0)
    ups = []
    prev = None
    hid = agg.classif1[0](out1)
    for k, (cl, out_next) in enumerate(((agg.classif1, out2), (agg.classif2, out3), (agg.classif3, None))):
        ev = main.record_event()
        nxt_cl = (agg.classif2, agg.classif3, None)[k]
        if out_next is not None:
            with torch.cuda.stream(side):
                side.wait_event(ev)
                cost = cl[1](hid, residual=prev)
                c, d = ops.trilinear_ac_soft_argmin(cost.squeeze(1), size, vals, 1.0)
                for t in (cost, c, d):
                    t.record_stream(main)
                hid.record_stream(side)
                done = side.record_event()
            if RESERVE_KB:   # development build: pad the convolution's LDS request so that only two of its workgroups fit a CU
                _lib.load().dmb_dev_set_option(17, RESERVE_KB)
            hid = nxt_cl[0](out_next)          # the matrix-bound convolution of the next level, on the caller's stream
            if RESERVE_KB:
                _lib.load().dmb_dev_set_option(17, 0)
            main.wait_event(done)
        else:
            cost = cl[1](hid, residual=prev)
            c, d = ops.trilinear_ac_soft_argmin(cost.squeeze(1), size, vals, 1.0)
        prev = cost
        ups.append(d)
    return ups[::-1]


def sequential():
    return model(batch)[0]["disps"]


def overlap_opt_in():
    ops.set_branch_overlap(True)
    try:
        return model(batch)[0]["disps"]
    finally:
        ops.set_branch_overlap(False)


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
    a, b = sequential(), tail_overlapped()
    print("identical to the sequential form:", all(torch.equal(x, y) for x, y in zip(a, b)))
    run(sequential, 5)
    acc = {"sequential": [], "tail under next conv": [], "opt-in branch overlap": []}
    for rep in range(4):
        for name, fn in (("sequential", sequential), ("tail under next conv", tail_overlapped), ("opt-in branch overlap", overlap_opt_in)):
            run(fn, 2)
            acc[name].append(run(fn, 8))
for name, ts in acc.items():
    print("%-24s %s  -> min %.3f ms" % (name, " ".join("%.3f" % t for t in ts), min(ts)))
