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
for B in (1, 4):
    D, H, W = 12, 24, 80
    x = torch.randn(B, 64, D, H, W, device=dev); x[..., 78:] = 0
    wp = ops.pack_deconv3d_weights(torch.randn(64, 64, 3, 3, 3, device=dev) * 0.03)
    r = torch.randn(B, 64, 2 * D, 2 * H, 156, device=dev)
    s1, s0 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    ref = None
    for rep in range(2):
        for opt, what in ((1, "28/60"), (2, "full 32"), (0, "pick")):
            lib.dmb_dev_set_option(29, opt)
            y = ops.deconv3d_k3s2(x, wp, 64, s1, s0, r, True, out_width=156)
            ref = y if ref is None else ref
            us = timeit(lambda: ops.deconv3d_k3s2(x, wp, 64, s1, s0, r, True, out_width=156))
            print("B=%d padded [12,24,80] -> 156 cols 64->64 +res  %-8s %8.1f us identical %s" % (B, what, us, torch.equal(y, ref)), flush=True)
    lib.dmb_dev_set_option(29, 0)
