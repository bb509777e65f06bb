"""Is the dominant kernel power/clock limited?  Same launch on zero-filled vs random data, short vs long bursts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from densematchingbenchmark_amd import ops
dev = torch.device("cuda:0")
B, D, H, W = 4, 48, 136, 240
w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03
wp = ops.pack_conv3d_weights(w)
fl = 2.0 * 27 * 32 * 32 * B * D * H * W

def run(x, n, tag):
    torch.cuda.synchronize(); time.sleep(1.0)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        ops.conv3d_k3(x, wp, 32)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / n
    print("%-34s n=%3d  %.3f ms  %.1f TFLOP/s" % (tag, n, ms, fl / ms / 1e9), flush=True)

xr = torch.randn(B, 32, D, H, W, device=dev)
xz = torch.zeros(B, 32, D, H, W, device=dev)
xs = torch.full((B, 32, D, H, W), 1.0, device=dev)
for _ in range(3): ops.conv3d_k3(xr, wp, 32)
for n in (1, 3, 20, 100):
    run(xr, n, "random N(0,1) input")
    run(xz, n, "zero input")
run(xs, 20, "all-ones input")
wz = ops.pack_conv3d_weights(torch.zeros_like(w))
wp_save = wp; wp = wz
run(xz, 20, "zero input + zero weights")
