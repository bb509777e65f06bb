"""Development aid: the two stride-1 layers of the step with the diagnostic switches (g_dev_opts[6]: 1 = no stores, 2 = no staging\nafter the first chunk): how far the full kernels are from their multiply-only structure.   python scripts/kbench_s1_diag.py"""
import os, sys
os.environ.setdefault("DMB_LIB", "dev")   # kernel-variant switches exist only in the development build (build.py --dev)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0"); lib = _lib.load()
B, D, H, W = 4, 48, 136, 240
def timeit(fn, n=20, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
x = torch.randn(B, 32, D, H, W, device=dev)
wp = ops.pack_conv3d_weights(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
x64 = torch.randn(B, 64, D // 2, H // 2, W // 2, device=dev)
wp64 = ops.pack_conv3d_weights(torch.randn(64, 64, 3, 3, 3, device=dev) * 0.03)
sc64, sh64 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
for rep in range(2):
    for opt in (0, 1, 2, 3, 35):
        lib.dmb_dev_set_option(6, opt)
        a = timeit(lambda: ops.conv3d_k3(x, wp, 32, sc, sh, None, 1, True))
        b = timeit(lambda: ops.conv3d_k3(x64, wp64, 64, sc64, sh64, None, 1, True))
        print("diag %d: 32->32 full %.3f ms   64->64 half %.3f ms" % (opt, a, b), flush=True)
lib.dmb_dev_set_option(6, 0)
