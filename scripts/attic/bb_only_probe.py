"""Development aid: the PSMNet backbone alone (4 pairs of 544x960), optionally (BB_NOSPP=1) without its four pooling branches --
how much of the forward is the serial chain avgpool -> 1x1 conv -> bilinear x 4 (twelve small launches per view)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import ops, synthetic
from densematchingbenchmark_amd.modeling.stereo.backbones import PSMNetBackbone
dev = torch.device("cuda:0")
bb = PSMNetBackbone(3, True).eval()
synthetic.init_params_(bb, seed=8, classif_gain=1.0)
bb = bb.to(dev)
l = torch.randn(4, 3, 544, 960, device=dev); r = torch.randn(4, 3, 544, 960, device=dev)


def timeit():
    with torch.no_grad():
        for _ in range(3): bb(l, r)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): bb(l, r)
        e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 10


print("backbone %.3f ms" % timeit())
if os.environ.get("BB_NOSPP"):
    real = ops.avgpool2d, ops.bilinear_ac
    for i in (1, 2, 3, 4):
        getattr(bb, "branch%d" % i)[1].forward = lambda x: x          # no 1x1 convolution
    ops.avgpool2d = lambda feat, k, in_window=None: None               # no pooling
    ops.bilinear_ac = lambda x, size, out=None, out_ch_offset=0: out   # no up-sampling (those channels stay uninitialised)
    print("backbone without the pooling branches %.3f ms" % timeit())
    ops.avgpool2d, ops.bilinear_ac = real
if os.environ.get("BB_OPT18"):
    from densematchingbenchmark_amd import _lib
    lib = _lib.load()
    for v in (0, 1, 2, 0):
        lib.dmb_dev_set_option(18, v)
        print("option 18 = %d (0 = by rounds x rows, 1 = always 4 rows per wave, 2 = always 2): backbone %.3f ms" % (v, timeit()))
    lib.dmb_dev_set_option(18, 0)
