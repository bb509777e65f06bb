"""Development aid: conv6 (transposed 64 -> 32, half -> full resolution, with its skip operand) under a start-up stagger of the
workgroups (development option 12: delay unit in s_sleep(127) periods of 3.4 us), for the three forms of the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
B, D, H, W = 4, 24, 68, 120


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


x = torch.randn(B, 64, D, H, W, device=dev)
wp = ops.pack_deconv3d_weights(torch.randn(64, 32, 3, 3, 3, device=dev) * 0.03)
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
r = torch.randn(B, 32, 2 * D, 2 * H, 2 * W, device=dev)
_x = torch.randn(B, 32, 48, 136, 240, device=dev)
_wp = ops.pack_conv3d_weights(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
for _ in range(100):
    ops.conv3d_k3(_x, _wp, 32, None, None, None, 1, False)
for rep in range(2):
    for form, name in ((0, "sixteen-wave"), (2, "z/y-parity items")):
        for st in (0, 1, 2, 3, 4, 6):
            lib.dmb_dev_set_option(4, form)
            lib.dmb_dev_set_option(11, 1 if form == 0 else 0)
            lib.dmb_dev_set_option(12, st)
            t_res = timeit(lambda: ops.deconv3d_k3s2(x, wp, 32, sc, sh, r, True))
            t_no = timeit(lambda: ops.deconv3d_k3s2(x, wp, 32, sc, sh, None, True))
            print("%-18s stagger %d: %.3f ms with the skip operand, %.3f ms without" % (name, st, t_res, t_no), flush=True)
lib.dmb_dev_set_option(4, 0)
lib.dmb_dev_set_option(11, 0)
lib.dmb_dev_set_option(12, 0)
