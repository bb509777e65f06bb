"""Development aid: the dominant stride-1 layer (32 -> 32, full resolution) and the 64 -> 64 half-resolution layer under a start-up
stagger of the first round's workgroups by their slot on the CU (g_dev_opts[14], unit 3.4 us), alternated inside one process; and
the launch at 2, 4 and 8 pairs (fixed cost per launch against cost per workgroup).   python scripts/attic/s1_stagger_probe.py"""
import os, sys
os.environ.setdefault("DMB_LIB", "dev")   # kernel-variant switches exist only in the development build (build.py --dev)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.load()
D, H, W = 48, 136, 240
def timeit(fn, n=20, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
wp = ops.pack_conv3d_weights(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
wp64 = ops.pack_conv3d_weights(torch.randn(64, 64, 3, 3, 3, device=dev) * 0.03)
sc64, sh64 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
xs = {B: torch.randn(B, 32, D, H, W, device=dev) for B in (2, 4, 8)}
x64 = torch.randn(4, 64, D // 2, H // 2, W // 2, device=dev)
for rep in range(3):
    for unit in (0, 4, 8, 10, 12, 16):
        lib.dmb_dev_set_option(14, unit)
        a = timeit(lambda: ops.conv3d_k3(xs[4], wp, 32, sc, sh, None, 1, True))
        b = timeit(lambda: ops.conv3d_k3(x64, wp64, 64, sc64, sh64, None, 1, True))
        print("stagger %2d: 32->32 full %.4f ms   64->64 half %.4f ms" % (unit, a, b), flush=True)
lib.dmb_dev_set_option(14, 0)
for rep in range(2):
    t = {B: timeit(lambda: ops.conv3d_k3(xs[B], wp, 32, sc, sh, None, 1, True)) for B in (2, 4, 8)}
    print("pairs 2 / 4 / 8: %.4f %.4f %.4f ms; per pair %.4f %.4f %.4f; fixed (2*T4 - T8) %.4f ms" % (
        t[2], t[4], t[8], t[2] / 2, t[4] / 4, t[8] / 8, 2 * t[4] - t[8]), flush=True)
