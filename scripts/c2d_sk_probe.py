"""Development aid (round 6): the 3x3 layers of the PSMNet backbone for ONE image per view (backbones/PSMNet.py:8-129 as the serving
API calls it) -- conv2d.hip's persistent tiles against the split-K form (csrc/conv3d_sk.hip, KZ = 1): correctness on awkward shapes,
then time per layer type at a list of image sizes (KB_SIZES = HxW,...), eager and as a replayed HIP graph.  Development library
(option 26: 1 = never split-K, 2 = always)."""
import os
os.environ.setdefault("DMB_LIB", "dev")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from densematchingbenchmark_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()


def timeit_graph(fn, n=20, reps=10):
    fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (n * reps) * 1e3


gen = torch.Generator().manual_seed(3)
bad = 0
for (B, Ci, Co, H, W, dil, use_res, relu) in ((1, 64, 64, 64, 128, 1, True, False), (2, 32, 32, 9, 20, 1, False, True), (1, 128, 128, 7, 36, 2, True, True),
                                              (1, 64, 128, 5, 12, 1, False, True), (3, 16, 32, 11, 44, 2, True, False), (1, 128, 32, 6, 28, 1, False, False)):
    x = torch.randn((B, Ci, H, W), generator=gen)
    w = torch.randn((Co, Ci, 3, 3), generator=gen) / (Ci * 9) ** 0.5
    sc, sh = 0.5 + torch.rand(Co, generator=gen), torch.rand(Co, generator=gen) - 0.5
    ref = F.conv2d(x, w, None, padding=dil, dilation=dil) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    res = torch.randn(ref.shape, generator=gen) if use_res else None
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    wp = ops.pack_conv2d_weights(w.to(dev))
    outs = []
    for o in (1, 2):
        lib.dmb_dev_set_option(26, o)
        got = ops.conv2d(x.to(dev), wp, Co, 3, 1, dil, sc.to(dev), sh.to(dev), res.to(dev) if res is not None else None, relu)
        outs.append(got)
        err = (got.cpu() - ref).abs().max().item()
        if not err <= 3e-5:
            bad += 1
            print("FAIL form", o, (B, Ci, Co, H, W, dil, use_res, relu), err)
    print((B, Ci, Co, H, W, dil), "split-K vs tiles max diff %.2e" % (outs[0] - outs[1]).abs().max().item())
lib.dmb_dev_set_option(26, 0)
print("correctness: failures", bad, flush=True)

for size in os.environ.get("KB_SIZES", "256x512,384x768,384x1248,544x960").split(","):
    Hh, Ww = (int(v) for v in size.split("x"))
    print("one image of %d x %d per view (launches of B = 1)" % (Hh, Ww))
    for name, Ci, Co, sc_, dil in (("32->32 /2", 32, 32, 2, 1), ("64->64 /4", 64, 64, 4, 1), ("128->128 /4", 128, 128, 4, 1), ("128->128 /4 dil 2", 128, 128, 4, 2)):
        h, w_ = Hh // sc_, Ww // sc_
        x = torch.randn(1, Ci, h, w_, device=dev)
        wp = ops.pack_conv2d_weights(torch.randn(Co, Ci, 3, 3, device=dev) * 0.05)
        s1, s0 = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
        out = torch.empty(1, Co, h, w_, device=dev)
        f = lambda: ops.conv2d(x, wp, Co, 3, 1, dil, s1, s0, None, True, out=out)   # noqa: E731
        fl = 2.0 * 9 * Ci * Co * h * w_
        units = ((h + 1) // 2) * ((w_ + 15) // 16) * (Co // 32)
        row = "  %-20s %6d units  at the matrix peak %6.1f us |" % (name, units, fl / 157.3e6)
        for o, what in ((1, "tiles"), (2, "split-K")):
            lib.dmb_dev_set_option(26, o)
            row += "  %s %7.1f us" % (what, timeit_graph(f))
        lib.dmb_dev_set_option(26, 0)
        print(row, flush=True)
