"""Development aid: the two transposed layers of the hourglass (conv5, conv6 with / without the skip operand) under ONE build of
the library -- run once per build-time variant (DMB_LIB=dev_st16 ...: cache policy of the epilogue's stores / loads, build.py) and
compare.  Three passes over the list; KB_SHAPE as in kbench_hg.py."""
import os
import sys

os.environ.setdefault("DMB_LIB", "dev")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from densematchingbenchmark_amd import ops

dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "4"))
D, H, W = [int(v) for v in os.environ.get("KB_SHAPE", "48,136,240").split(",")]


def timeit(fn, n=40, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def case(Ci, Co, d, h, w, res):
    x = torch.randn(B, Ci, d, h, w, device=dev)
    wp = ops.pack_deconv3d_weights(torch.randn(Ci, Co, 3, 3, 3, device=dev) * 0.03)
    sc, sh = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    r = torch.randn(B, Co, 2 * d, 2 * h, 2 * w, device=dev) if res else None
    return timeit(lambda: ops.deconv3d_k3s2(x, wp, Co, sc, sh, r, True))


_x = torch.randn(B, 32, D, H, W, device=dev)
_wp = ops.pack_conv3d_weights(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
for _ in range(150):
    ops.conv3d_k3(_x, _wp, 32, None, None, None, 1, False)
torch.cuda.synchronize()
del _x
rows = []
for rep in range(3):
    rows.append((case(64, 64, D // 4, H // 4, W // 4, True), case(64, 32, D // 2, H // 2, W // 2, False), case(64, 32, D // 2, H // 2, W // 2, True)))
best = [min(r[i] for r in rows) for i in range(3)]
print("%-14s conv5+res %.3f ms   conv6 %.3f ms   conv6+res %.3f ms   (min of 3 passes; all: %s)" % (
    os.environ["DMB_LIB"], best[0], best[1], best[2], " | ".join("%.3f %.3f %.3f" % r for r in rows)), flush=True)
