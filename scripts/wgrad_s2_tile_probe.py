"""Development aid (round 6): the stride-2 / transposed weight gradient on its 2 x 12 and 4 x 8 tiles (development option 27 = 1 / 2;
0 = the library's pick) at the training crop, the BASELINE size and the KITTI size: time, fraction of the FP32 matrix peak, agreement."""
import os
os.environ.setdefault("DMB_LIB", "dev")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()


def timeit(fn, n=30, warm=8):
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


for B, (D, H, W) in ((4, (48, 64, 128)), (2, (48, 136, 240)), (2, (48, 96, 312)), (4, (24, 48, 96))):
    for name, Cs, Cb, sc in (("32 <-> 64 full / half", 64, 32, 1), ("64 <-> 64 half / quarter", 64, 64, 2)):
        d, h, w = D // sc, H // sc, W // sc
        big = torch.randn(B, Cb, d, h, w, device=dev)
        small = torch.randn(B, Cs, d // 2, h // 2, w // 2, device=dev)
        fl = 2.0 * 27 * Cs * Cb * small[0, 0].numel() * B
        ref = None
        for opt, what in ((0, "library's pick"), (1, "2 x 12 tiles"), (2, "4 x 8 tiles")):
            lib.dmb_dev_set_option(27, opt)
            y = ops.conv3d_k3s2_wgrad(big, small)
            ref = y if ref is None else ref
            us = timeit(lambda: ops.conv3d_k3s2_wgrad(big, small))
            print("B=%d small [%d,%d,%d] %-26s %-15s %8.1f us  %.3f of peak  max |diff| / range %.1e" %
                  (B, d // 2, h // 2, w // 2, name, what, us, fl / us / 1e6 / 157.3, float((y - ref).abs().max() / ref.abs().max())), flush=True)
        lib.dmb_dev_set_option(27, 0)
