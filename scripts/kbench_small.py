"""Development aid (round 5): the hourglass layers where a launch does NOT fill the chip -- batch 1 at BASELINE configs[0]
(256x512 / D 64: quarter-resolution volume 16 x 64 x 128), at 544x960 and at the KITTI shape -- per tile candidate (development
options 19 / 10), against the library's own pick.  KB_B (default 1)."""
import os
os.environ.setdefault("DMB_LIB", "dev")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from densematchingbenchmark_amd import _lib, ops

dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "1"))
lib = _lib.load()


def timeit(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def conv(Ci, Co, stride, d, h, w, res=False):
    x = torch.randn(B, Ci, d, h, w, device=dev)
    wp = ops.pack_conv3d_weights(torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.03)
    sc, sh = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    do, ho, wo = [(n - 1) // stride + 1 for n in (d, h, w)]
    r = torch.randn(B, Co, do, ho, wo, device=dev) if res else None
    fl = 2.0 * 27 * Ci * Co * B * do * ho * wo
    return (lambda: ops.conv3d_k3(x, wp, Co, sc, sh, r, stride, True)), fl


def deconv(Ci, Co, d, h, w, res=True):
    x = torch.randn(B, Ci, d, h, w, device=dev)
    wp = ops.pack_deconv3d_weights(torch.randn(Ci, Co, 3, 3, 3, device=dev) * 0.03)
    sc, sh = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    r = torch.randn(B, Co, 2 * d, 2 * h, 2 * w, device=dev) if res else None
    return (lambda: ops.deconv3d_k3s2(x, wp, Co, sc, sh, r, True)), 2.0 * 27 * Ci * Co * B * d * h * w


def line(name, us, fl):
    print("  %-58s %8.1f us  %6.1f TFLOP/s (%.2f of peak)" % (name, us, fl / us / 1e6, fl / us / 1e6 / 157.3), flush=True)


for D, H, W in ((16, 64, 128), (48, 136, 240), (48, 96, 312)):
    print("B = %d, quarter-resolution volume %d x %d x %d" % (B, D, H, W))
    f, fl = conv(32, 32, 1, D, H, W)
    line("32->32 full (library's pick)", timeit(f), fl)
    for cand, what in ((1, "48 x 4 pairs"), (2, "24 x 8 quads"), (3, "32 x 4 pairs"), (4, "32 x 2 pairs")):
        lib.dmb_dev_set_option(19, cand)
        line("32->32 full on " + what, timeit(f), fl)
    lib.dmb_dev_set_option(19, 0)
    for name, (Ci, Co, s, sc) in (("conv1 s2 32->64 full->half", (32, 64, 2, 1)), ("conv3 s2 64->64 half->quarter", (64, 64, 2, 2))):
        f, fl = conv(Ci, Co, s, D // sc, H // sc, W // sc)
        line(name + " (library's pick)", timeit(f), fl)
        for opt, what in ((1, "4 x 30, four waves"), (2, "4 x 30 / 4 x 22 (round 4)")):
            lib.dmb_dev_set_option(10, opt)
            line(name + " on " + what, timeit(f), fl)
        lib.dmb_dev_set_option(10, 0)
    for name, sc, res in (("conv2 s1 64->64 half", 2, True), ("conv4 s1 64->64 quarter", 4, False)):
        f, fl = conv(64, 64, 1, D // sc, H // sc, W // sc, res)
        line(name + " (library's pick)", timeit(f), fl)
        for cand, what in ((1, "64-voxel runs"), (2, "40 x 4 quads"), (3, "24 x 4 quads"), (4, "32 x 4 quads"), (5, "16 x 2 pairs")):
            lib.dmb_dev_set_option(19, cand)
            try:
                line(name + " on " + what, timeit(f), fl)
            except Exception as e:  # noqa: BLE001
                print("  %s on %s: %r" % (name, what, e))
        lib.dmb_dev_set_option(19, 0)
    if (W // 4) % 4 == 0:
        f, fl = deconv(64, 64, D // 4, H // 4, W // 4)
        line("conv5 deconv 64->64 quarter->half +res", timeit(f), fl)
    f, fl = deconv(64, 32, D // 2, H // 2, W // 2)
    line("conv6 deconv 64->32 half->full +res", timeit(f), fl)
    x = torch.randn(B, 32, D, H, W, device=dev)
    wc = torch.randn(1, 32, 3, 3, 3, device=dev) * 0.03
    us = timeit(lambda: ops.conv3d_k3_c1(x, wc))
    print("  %-58s %8.1f us  %6.2f TB/s" % ("32->1 head", us, x.numel() * 4 / us / 1e6))
    q = torch.randn(B, D, H, W, device=dev)
    vals = ops.disp_sample_values(4 * D, 0, 1)
    us = timeit(lambda: ops.trilinear_ac_soft_argmin(q, (4 * D, 4 * H, 4 * W), vals, 1.0))
    print("  %-58s %8.1f us  %6.2f TB/s" % ("up-sampling + regression", us, B * 64 * D * H * W * 4 / us / 1e6))
