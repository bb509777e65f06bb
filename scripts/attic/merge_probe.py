"""Development probe: what merging the three head / up-sampling launches of a step into one would buy (a batch of 12 in one launch
against three launches of 4: the same work, 9.6 instead of 3 x 3.2 tile rounds).   python scripts/attic/merge_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from densematchingbenchmark_amd import ops

dev = torch.device("cuda:0")
D, H, W = 48, 136, 240


def timeit(fn, n=30, warm=10):
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


w1 = torch.randn(1, 32, 3, 3, 3, device=dev)
x4 = [torch.randn(4, 32, D, H, W, device=dev) for _ in range(3)]
x12 = torch.randn(12, 32, D, H, W, device=dev)
vals = ops.disp_sample_values(192, 0, 1)
c4 = [torch.randn(4, D, H, W, device=dev) for _ in range(3)]
c12 = torch.randn(12, D, H, W, device=dev)
for rep in range(2):
    t3 = timeit(lambda: [ops.conv3d_k3_c1(x, w1, 0.0, None) for x in x4])
    t1 = timeit(lambda: ops.conv3d_k3_c1(x12, w1, 0.0, None))
    print("32->1 head:      3 launches of 4: %.3f ms   1 launch of 12: %.3f ms" % (t3, t1), flush=True)
    t3 = timeit(lambda: [ops.trilinear_ac_soft_argmin(c, (192, 544, 960), vals, 1.0) for c in c4], n=10, warm=3)
    t1 = timeit(lambda: ops.trilinear_ac_soft_argmin(c12, (192, 544, 960), vals, 1.0), n=10, warm=3)
    print("up-sampling:     3 launches of 4: %.3f ms   1 launch of 12: %.3f ms" % (t3, t1), flush=True)
