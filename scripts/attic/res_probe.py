"""Development aid: what the skip operand costs the stride-1 layers (64 -> 64 at half resolution, 32 -> 32 at full resolution):
without it, with it, and with the outputs of the launch before it as the operand (warm in L2 or not).   python scripts/attic/res_probe.py"""
import os, sys
os.environ.setdefault("DMB_LIB", "dev")   # kernel-variant switches exist only in the development build (build.py --dev)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.load()
def timeit(fn, n=20, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (C, D, H, W) in ((64, 24, 68, 120), (32, 48, 136, 240)):
    x = torch.randn(4, C, D, H, W, device=dev)
    r = torch.randn(4, C, D, H, W, device=dev)
    wp = ops.pack_conv3d_weights(torch.randn(C, C, 3, 3, 3, device=dev) * 0.03)
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    for rep in range(3):
        a = timeit(lambda: ops.conv3d_k3(x, wp, C, sc, sh, None, 1, True))
        b = timeit(lambda: ops.conv3d_k3(x, wp, C, sc, sh, r, 1, True))
        lib.dmb_dev_set_option(6, 1)
        a1 = timeit(lambda: ops.conv3d_k3(x, wp, C, sc, sh, None, 1, True))
        lib.dmb_dev_set_option(6, 0)
        print("%d -> %d  %dx%dx%d: no operand %.4f ms, with %.4f ms (+%.1f %%), no epilogue at all %.4f ms" % (
            C, C, D, H, W, a, b, 100 * (b / a - 1), a1), flush=True)
