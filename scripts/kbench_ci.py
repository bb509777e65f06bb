import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import ops
dev = torch.device("cuda:0")
B, D, H, W = 4, 48, 136, 240
def timeit(fn, n=20, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for rep in range(2):
  for Ci in (32, 40, 48, 64):
    x = torch.randn(B, Ci, D, H, W, device=dev)
    wp = ops.pack_conv3d_weights(torch.randn(32, Ci, 3, 3, 3, device=dev) * 0.03)
    sc, sh = torch.ones(32, device=dev), torch.zeros(32, device=dev)
    r = torch.randn(B, 32, D, H, W, device=dev)
    ms = timeit(lambda: ops.conv3d_k3(x, wp, 32, sc, sh, None, 1, True))
    ms_r = timeit(lambda: ops.conv3d_k3(x, wp, 32, sc, sh, r, 1, True))
    fl = 2.0 * 27 * Ci * 32 * B * D * H * W
    print("Ci=%d -> 32: %.3f ms  %.1f TF/s   with residual %.3f ms" % (Ci, ms, fl / ms / 1e9, ms_r), flush=True)
    del x, r
