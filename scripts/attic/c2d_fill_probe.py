import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import _lib, ops
dev = torch.device("cuda:0")
for kv in filter(None, os.environ.get("KC_OPTS", "").split(",")):   # development options (DMB_LIB=dev...)
    _lib.load().dmb_dev_set_option(*[int(v) for v in kv.split("=")])
def timeit(fn, n=40, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for C, h, w in ((64, 136, 240), (128, 136, 240), (32, 272, 480)):
    for NB in (3, 6, 8, 12, 16):
        x = torch.randn(NB, C, h, w, device=dev)
        wp = ops.pack_conv2d_weights(torch.randn(C, C, 3, 3, device=dev) * 0.05)
        sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        ms = timeit(lambda: ops.conv2d(x, wp, C, 3, 1, 1, sc, sh, None, True))
        fl = 2.0 * 9 * C * C * NB * h * w
        print("%3d->%3d %dx%d  images %2d: %.3f ms  %.4f ms/image  %.1f TF/s (%.1f%%)" % (C, C, h, w, NB, ms, ms / NB, fl / ms / 1e9, fl / ms / 1e9 / 1.573), flush=True)
