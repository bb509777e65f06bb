"""Micro-benchmark of the opt-in bf16x6 split kernels at the cfg2 shapes (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import ops

dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "4"))


def timeit(fn, n=10, warm=3):
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


for rep in range(2):
    for ci, co, d, h, w in ((32, 32, 48, 136, 240), (64, 32, 48, 136, 240), (64, 64, 24, 68, 120)):
        x = torch.randn(B, ci, d, h, w, device=dev)
        wt = torch.randn(co, ci, 3, 3, 3, device=dev) * 0.03
        wp = ops.pack_conv3d_x6_weights(wt)
        sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
        ms = timeit(lambda: ops.conv3d_k3_x6(x, wp, co, sc, sh, None, True))
        fl = 2.0 * 27 * ci * co * B * d * h * w
        print("x6 %d->%d %dx%dx%d  %8.3f ms  %7.1f TFLOP/s FP32-equivalent, %6.0f TFLOP/s issued bf16" %
              (ci, co, d, h, w, ms, fl / ms / 1e9, fl * 6 * 28 / 27 / ms / 1e9), flush=True)
