import os
os.environ.setdefault("DMB_LIB", "dev")
import sys
sys.path.insert(0, "/root/repo")
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B, (D, H, W), Co in ((4, (24, 32, 64), 32), (4, (12, 16, 32), 64), (2, (24, 68, 120), 32)):
    x = torch.randn(B, 64, D, H, W, device=dev)
    wp = ops.pack_deconv3d_weights(torch.randn(64, Co, 3, 3, 3, device=dev) * 0.03)
    for rep in range(2):
        for opt, what in ((3, "full 64"), (2, "full 32"), (0, "pick")):
            lib.dmb_dev_set_option(29, opt)
            us = timeit(lambda: ops.deconv3d_k3s2(x, wp, Co, None, None, None, False))
            print("B=%d in [%d,%d,%d] 64->%d raw  %-8s %8.1f us" % (B, D, H, W, Co, what, us), flush=True)
        lib.dmb_dev_set_option(4, 1)
        us = timeit(lambda: ops.deconv3d_k3s2(x, wp, Co, None, None, None, False))
        print("B=%d in [%d,%d,%d] 64->%d raw  %-8s %8.1f us" % (B, D, H, W, Co, "both-y", us), flush=True)
        lib.dmb_dev_set_option(4, 0)
    lib.dmb_dev_set_option(29, 0)
