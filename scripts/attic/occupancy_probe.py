"""Development aid: the dominant stride-1 layer with 3, 2 and 1 workgroups per CU (development option 17 = extra KB of LDS per
workgroup), complete and as bare MFMA + LDS-read structure (option 6 = 35): how much of the matrix pipe ONE wave per SIMD can use --
i.e. whether a workgroup in its set-up / epilogue costs its third of the CU or nothing.   python scripts/attic/occupancy_probe.py"""
import os, sys
os.environ.setdefault("DMB_LIB", "dev")   # kernel-variant switches exist only in the development build (build.py --dev)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.load()
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
x = torch.randn(4, 32, 48, 136, 240, device=dev)
wp = ops.pack_conv3d_weights(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
for rep in range(2):
    for extra, label in ((0, "3 workgroups / CU"), (28, "2 workgroups / CU"), (60, "1 workgroup / CU")):
        lib.dmb_dev_set_option(17, extra)
        out = []
        for diag in (0, 3, 35):
            lib.dmb_dev_set_option(6, diag)
            out.append(timeit(lambda: ops.conv3d_k3(x, wp, 32, sc, sh, None, 1, True)))
        lib.dmb_dev_set_option(6, 0)
        print("%-20s complete %.3f ms (%.3f of peak)   bare %.3f (%.3f)   bare, no barrier %.3f (%.3f)" % (
            label, out[0], 2.2030 / out[0], out[1], 2.2030 / out[1], out[2], 2.2030 / out[2]), flush=True)
lib.dmb_dev_set_option(17, 0)
