import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (Ci, Co, h, w, NB) in ((32, 32, 272, 480, 8), (64, 64, 136, 240, 8), (128, 128, 136, 240, 8), (32, 32, 384, 1248, 16)):
    x = torch.randn(NB, Ci, h, w, device=dev); wt = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
    wp = ops.pack_conv2d_weights(wt); sc = torch.ones(Co, device=dev); sh = torch.zeros(Co, device=dev)
    fl = 2.0 * 9 * Ci * Co * NB * h * w
    for dbg in (1, 0):
        lib.dmb_dev_set_option(3, dbg)
        ms = timeit(lambda: ops.conv2d(x, wp, Co, 3, 1, 1, sc, sh, None, True))
        rs = torch.randn(NB, Co, h, w, device=dev)
        ms2 = timeit(lambda: ops.conv2d(x, wp, Co, 3, 1, 1, sc, sh, rs, False))
        del rs
        print("%d->%d %dx%d x%d scalar_path=%d: %.3f ms %.1f TF | +residual %.3f ms %.1f TF" % (Ci, Co, h, w, NB, dbg, ms, fl / ms / 1e9, ms2, fl / ms2 / 1e9), flush=True)
    lib.dmb_dev_set_option(3, 0)
