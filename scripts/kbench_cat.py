"""Development aid: the first aggregator layer on a concatenation volume -- materialised (cat_fms + 3-D conv) against the
2-D form (csrc/catconv.hip) at the BASELINE cfg2 shape, piece by piece."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import ops

dev = torch.device("cuda:0")
B, C, D, H, W = 4, 32, 48, 136, 240


def timeit(fn, n=20, warm=5):
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


L, R = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
w = torch.randn(32, 64, 3, 3, 3, device=dev) * 0.03
sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
idx = ops.disp_index_list(D, 0, 1)
wp = ops.pack_conv3d_weights(w)
packs = ops.catconv_pack(w)
for _ in range(100):
    ops.conv3d_k3(torch.empty(B, 32, D, H, W, device=dev), ops.pack_conv3d_weights(w[:, :32].contiguous()), 32)
torch.cuda.synchronize()
t_vol = timeit(lambda: ops.cat_fms(L, R, idx))
vol = ops.cat_fms(L, R, idx)
t_conv = timeit(lambda: ops.conv3d_k3(vol, wp, 32, sc, sh, None, 1, True))
del vol
t_fused = timeit(lambda: ops.catconv_first(L, R, D, packs, sc, sh, True))
print("cat_fms %.3f ms + conv3d 64->32 %.3f ms = %.3f ms;  2-D form %.3f ms" % (t_vol, t_conv, t_vol + t_conv, t_fused))
Wc = D + 4
print("  conv2d L  -> 128 ch [%d]   %.3f ms" % (W, timeit(lambda: ops.conv2d(L, packs["A"], 128, 3))))
Lc = ops.copy_window(L, Wc, 0)
print("  conv2d Lc -> 128 ch [%d]    %.3f ms (x2)" % (Wc, timeit(lambda: ops.conv2d(Lc, packs["B1"], 128, 3))))
Rz = ops.copy_window(R, W + 4, -4)
print("  conv2d Rz -> 128 ch [%d]   %.3f ms" % (W + 4, timeit(lambda: ops.conv2d(Rz, packs["HC"], 128, 3))))
print("  copy_window x3             %.3f ms" % timeit(lambda: (ops.copy_window(L, Wc, 0), ops.copy_window(R, W + 4, -4), ops.copy_window(R, Wc, W - Wc))))
