"""Development aid (round 5): the stride-2 layers at batch 4 on the one- and two-row tiles (development option 10 = 3 / 4) that the
library picks for launches of at most one workgroup per CU -- do finer work units also pay on a full grid?"""
import os
os.environ.setdefault("DMB_LIB", "dev")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, n=40, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B, (D, H, W) in ((4, (48, 136, 240)), (4, (48, 96, 312)), (4, (48, 64, 128))):
    for name, Ci, sc in (("conv1 s2 32->64", 32, 1), ("conv3 s2 64->64", 64, 2)):
        d, h, w = D // sc, H // sc, W // sc
        x = torch.randn(B, Ci, d, h, w, device=dev)
        wp = ops.pack_conv3d_weights(torch.randn(64, Ci, 3, 3, 3, device=dev) * 0.03)
        s1, s0 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
        fl = 2.0 * 27 * Ci * 64 * B * ((d - 1) // 2 + 1) * ((h - 1) // 2 + 1) * ((w - 1) // 2 + 1)
        ref = None
        for opt, what in ((0, "library's pick"), (3, "one-row tiles"), (4, "two-row tiles")):
            lib.dmb_dev_set_option(10, opt)
            y = ops.conv3d_k3(x, wp, 64, s1, s0, None, 2, True)
            ref = y if ref is None else ref
            us = timeit(lambda: ops.conv3d_k3(x, wp, 64, s1, s0, None, 2, True))
            print("B=%d [%d,%d,%d] %-16s %-15s %8.1f us  %.3f of peak  identical %s" % (B, d, h, w, name, what, us, fl / us / 1e6 / 157.3, torch.equal(y, ref)), flush=True)
        lib.dmb_dev_set_option(10, 0)
