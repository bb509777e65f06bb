"""Development aid (round 6): the 32 -> 32 stride-1 layer on its tile candidates (development option 19 = k forces candidate k - 1:
1 = 48 x 4, 3 = 32 x 4, 5 = 64 x 4) at the training crop and other 64 / 128-column shapes, with and without the skip operand."""
import os
os.environ.setdefault("DMB_LIB", "dev")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, n=20, warm=5):
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


for B, (D, H, W) in ((4, (48, 64, 128)), (2, (48, 64, 128)), (4, (48, 32, 64)), (4, (48, 136, 240)), (1, (48, 96, 312)), (4, (16, 64, 128))):
    x = torch.randn(B, 32, D, H, W, device=dev)
    r = torch.randn(B, 32, D, H, W, device=dev)
    wp = ops.pack_conv3d_weights(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
    fl = 2.0 * 27 * 32 * 32 * B * D * H * W
    for res in (None, r):
        ref = None
        for opt, what in ((0, "library's pick"), (1, "48 x 4"), (3, "32 x 4"), (5, "64 x 4")):
            lib.dmb_dev_set_option(19, opt)
            y = ops.conv3d_k3(x, wp, 32, None, None, res, 1, False)
            ref = y if ref is None else ref
            us = timeit(lambda: ops.conv3d_k3(x, wp, 32, None, None, res, 1, False))
            print("B=%d [%d,%d,%d] %-5s %-15s %8.1f us  %.3f of peak  identical %s" %
                  (B, D, H, W, "+res" if res is not None else "", what, us, fl / us / 1e6 / 157.3, torch.equal(y, ref)), flush=True)
        lib.dmb_dev_set_option(19, 0)
