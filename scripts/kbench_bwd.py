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

for ci, co, d, h, w in ((32, 64, 48, 136, 240), (64, 64, 24, 68, 120)):
    x = torch.randn(B, ci, d, h, w, device=dev)
    dc = torch.randn(B, co, d // 2, h // 2, w // 2, device=dev)
    fl = 2.0 * 27 * ci * co * B * (d // 2) * (h // 2) * (w // 2)
    report("wgrad s2 conv %d->%d from %dx%dx%d" % (ci, co, d, h, w), timeit(lambda: ops.conv3d_k3s2_wgrad(x, dc)), fl)
    wt = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.03
    report("dgrad s2 conv %d->%d" % (ci, co), timeit(lambda: ops.conv3d_k3_dgrad(dc, wt, 2)), fl)
for ci, co, d, h, w in ((64, 32, 24, 68, 120), (64, 64, 12, 34, 60)):
    x = torch.randn(B, ci, d, h, w, device=dev)
    dy = torch.randn(B, co, 2 * d, 2 * h, 2 * w, device=dev)
    fl = 2.0 * 27 * ci * co * B * d * h * w
    report("wgrad deconv %d->%d from %dx%dx%d" % (ci, co, d, h, w), timeit(lambda: ops.deconv3d_k3s2_wgrad(x, dy)), fl)
    wt = torch.randn(ci, co, 3, 3, 3, device=dev) * 0.03
    report("dgrad deconv %d->%d" % (ci, co), timeit(lambda: ops.deconv3d_k3s2_dgrad(dy, wt)), fl)

# AcfNet confidence head (cmn/cmn.py:21-36): 2-D convolution 192 -> 64 over the full-resolution cost volume
for ci, co, h, w in ((192, 64, 544, 960),):
    x = torch.randn(B, ci, h, w, device=dev)
    dc = torch.randn(B, co, h, w, device=dev)
    wt = torch.randn(co, ci, 3, 3, device=dev) * 0.03
    fl = 2.0 * 9 * ci * co * B * h * w
    report("wgrad 2-D %d->%d %dx%d" % (ci, co, h, w), timeit(lambda: ops.conv2d_k3_wgrad(x, dc)), fl)
    report("dgrad 2-D %d->%d %dx%d" % (ci, co, h, w), timeit(lambda: ops.conv2d_dgrad(dc, wt)), fl)
