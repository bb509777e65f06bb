"""Micro-benchmark of the backward kernels at the cfg2 shapes (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import ops

dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "4"))


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def report(name, ms, fl):
    print("%-34s %8.3f ms  %7.1f TFLOP/s (%.0f%% of 157.3)" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100), flush=True)


for ci, co, d, h, w in ((32, 32, 48, 136, 240), (64, 32, 48, 136, 240), (64, 64, 24, 68, 120), (64, 64, 12, 34, 60)):
    x = torch.randn(B, ci, d, h, w, device=dev)
    dc = torch.randn(B, co, d, h, w, device=dev)
    wt = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.03
    fl = 2.0 * 27 * ci * co * B * d * h * w
    report("wgrad s1 %d->%d %dx%dx%d" % (ci, co, d, h, w), timeit(lambda: ops.conv3d_k3_wgrad(x, dc)), fl)
    wp = ops.pack_conv3d_dgrad_weights(wt)
    report("dgrad s1 %d->%d %dx%dx%d" % (ci, co, d, h, w), timeit(lambda: ops.conv3d_k3(dc, wp, ci)), fl)
