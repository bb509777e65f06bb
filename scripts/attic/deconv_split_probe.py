import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from densematchingbenchmark_amd import ops
dev = torch.device("cuda:0")
B, D, H, W = 4, 12, 34, 60
def timeit(fn, n=50, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
x = torch.randn(B, 64, D, H, W, device=dev)
xw = torch.randn(B, 32, 48, 136, 240, device=dev); wpw = ops.pack_conv3d_weights(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
for _ in range(100): ops.conv3d_k3(xw, wpw, 32)
w64 = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.03
wp64 = ops.pack_deconv3d_weights(w64)
wpa, wpb = ops.pack_deconv3d_weights(w64[:, :32].contiguous()), ops.pack_deconv3d_weights(w64[:, 32:].contiguous())
sc64, sh64 = torch.ones(64, device=dev), torch.zeros(64, device=dev)
r64 = torch.randn(B, 64, 2 * D, 2 * H, 2 * W, device=dev)
r32 = r64[:, :32].contiguous()
for rep in range(2):
    print("64->64 one launch      %.3f ms" % timeit(lambda: ops.deconv3d_k3s2(x, wp64, 64, sc64, sh64, r64, True)))
    print("2 x (64->32) launches  %.3f ms" % timeit(lambda: (ops.deconv3d_k3s2(x, wpa, 32, sc64[:32], sh64[:32], r32, True), ops.deconv3d_k3s2(x, wpb, 32, sc64[:32], sh64[:32], r32, True))))
