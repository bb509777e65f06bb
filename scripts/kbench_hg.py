"""Development aid: the hourglass layers that sit below the MFMA roofline (stride-2, transposed, quarter resolution) at the
BASELINE cfg2 shapes, with the residual operands they carry in the real step and with diagnostic switches
(g_dev_opts[6]: 1 = no stores, 2 = no staging after the first chunk) to see which part of a kernel is exposed.
    python scripts/kbench_hg.py [diag]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from densematchingbenchmark_amd import _lib, ops

dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "4"))
D, H, W = 48, 136, 240
lib = _lib.load()
diag = len(sys.argv) > 1 and sys.argv[1] == "diag"


def timeit(fn, n=30, warm=10):
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


def report(name, ms, fl):
    print("%-44s %8.3f ms  %7.2f TFLOP/s  (%.1f%% of 157.3)" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100), flush=True)


def conv_case(Ci, Co, stride, d, h, w, name, res=False):
    x = torch.randn(B, Ci, d, h, w, device=dev)
    wt = torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.03
    wp = ops.pack_conv3d_weights(wt)
    sc = torch.ones(Co, device=dev)
    sh = torch.zeros(Co, device=dev)
    do, ho, wo = (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1
    r = torch.randn(B, Co, do, ho, wo, device=dev) if res else None
    fl = 2.0 * 27 * Ci * Co * B * do * ho * wo
    for opt in ((0, 1, 2, 3) if diag else (0,)):
        lib.dmb_dev_set_option(6, opt)
        report(name + (" +res" if res else "") + (" [diag %d]" % opt if opt else ""),
               timeit(lambda: ops.conv3d_k3(x, wp, Co, sc, sh, r, stride, True)), fl)
    lib.dmb_dev_set_option(6, 0)


def deconv_case(Ci, Co, d, h, w, name, res=False):
    x = torch.randn(B, Ci, d, h, w, device=dev)
    wt = torch.randn(Ci, Co, 3, 3, 3, device=dev) * 0.03
    wp = ops.pack_deconv3d_weights(wt)
    sc = torch.ones(Co, device=dev)
    sh = torch.zeros(Co, device=dev)
    r = torch.randn(B, Co, 2 * d, 2 * h, 2 * w, device=dev) if res else None
    fl = 2.0 * 27 * Ci * Co * B * d * h * w
    for opt in ((0, 1, 2, 3, 4, 8, 64, 65, 72) if diag else (0,)):
        lib.dmb_dev_set_option(6, opt)
        report(name + (" +res" if res else "") + (" [diag %d]" % opt if opt else ""),
               timeit(lambda: ops.deconv3d_k3s2(x, wp, Co, sc, sh, r, True)), fl)
    lib.dmb_dev_set_option(6, 0)


print("B =", B)
# the chip clocks by its power budget: bring it to a steady state first, and run the list twice
_x = torch.randn(B, 32, D, H, W, device=dev)
_wp = ops.pack_conv3d_weights(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
for _ in range(150):
    ops.conv3d_k3(_x, _wp, 32, None, None, None, 1, False)
torch.cuda.synchronize()
del _x
for _rep in range(1 if diag else 2):
  conv_case(32, 64, 2, D, H, W, "conv1 s2 32->64 full->half")
  conv_case(64, 64, 1, D // 2, H // 2, W // 2, "conv2 s1 64->64 half", res=True)
  conv_case(64, 64, 2, D // 2, H // 2, W // 2, "conv3 s2 64->64 half->quarter")
  conv_case(64, 64, 1, D // 4, H // 4, W // 4, "conv4 s1 64->64 quarter")
  deconv_case(64, 64, D // 4, H // 4, W // 4, "conv5 deconv 64->64 quarter->half", res=True)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv 64->32 half->full", res=False)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv 64->32 half->full", res=True)
  lib.dmb_dev_set_option(13, 1)   # A/B: quarter-resolution stride-1 layer on 4 x 4 x 60 boxes (216 workgroups) instead of 64-voxel runs (768)
  conv_case(64, 64, 1, D // 4, H // 4, W // 4, "conv4 s1 64->64 quarter (box tiles)")
  lib.dmb_dev_set_option(13, 0)
  lib.dmb_dev_set_option(10, 1)   # A/B: stride-2 layers on four-wave workgroups (two waves per SIMD)
  conv_case(32, 64, 2, D, H, W, "conv1 s2 32->64 (4-wave workgroups)")
  conv_case(64, 64, 2, D // 2, H // 2, W // 2, "conv3 s2 64->64 (4-wave workgroups)")
  lib.dmb_dev_set_option(10, 0)
  lib.dmb_dev_set_option(11, 1)   # A/B: one sixteen-wave workgroup per CU computes all eight parity classes of a tile (opt-in)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv (sixteen-wave workgroups)", res=False)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv (sixteen-wave workgroups)", res=True)
  lib.dmb_dev_set_option(11, 0)
  lib.dmb_dev_set_option(4, 1)   # A/B: the two-parity form (both y parities per item, two workgroups per CU)
  deconv_case(64, 64, D // 4, H // 4, W // 4, "conv5 deconv (items with both y parities)", res=True)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv (items with both y parities)", res=False)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv (items with both y parities)", res=True)
  lib.dmb_dev_set_option(4, 0)
