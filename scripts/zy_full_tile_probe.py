import os
os.environ.setdefault("DMB_LIB", "dev")
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load()
def timeit(fn, n=30, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for B, (D, H, W), Co in ((4, (24, 48, 156), 32), (1, (24, 48, 156), 32), (4, (24, 32, 64), 32), (4, (12, 16, 32), 64), (4, (24, 68, 120), 32), (2, (24, 40, 96), 32), (1, (12, 24, 40), 64)):
    x = torch.randn(B, 64, D, H, W, device=dev)
    wp = ops.pack_deconv3d_weights(torch.randn(64, Co, 3, 3, 3, device=dev) * 0.03)
    r = torch.randn(B, Co, 2 * D, 2 * H, 2 * W, device=dev)
    s1, s0 = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    ref = None
    for opt in (1, 0):
        lib.dmb_dev_set_option(29, opt)
        y = ops.deconv3d_k3s2(x, wp, Co, s1, s0, r, True)
        ref = y if ref is None else ref
        us = timeit(lambda: ops.deconv3d_k3s2(x, wp, Co, s1, s0, r, True))
        print("B=%d in [%d,%d,%d] 64->%d +res  %-22s %8.1f us  identical %s" % (B, D, H, W, Co, "28/60-column tiles" if opt else "library's pick", us, torch.equal(y, ref)), flush=True)
    lib.dmb_dev_set_option(29, 0)
