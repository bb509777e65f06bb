"""Development aid: the hourglass layers that sit below the MFMA roofline (stride-2, transposed, quarter resolution) with the
residual operands they carry in the real step, and (``diag``) with diagnostic switches (development option 6: 1 = no stores,
2 = no staging after the first chunk) to see which part of a kernel is exposed.  Shape: KB_SHAPE = D,H,W of the FULL-resolution
feature volume (default 48,136,240 = BASELINE cfg2; 48,96,312 = the reference's KITTI operating point), KB_B pairs.
    python scripts/kbench_hg.py [diag]"""
import os
os.environ.setdefault("DMB_LIB", "dev")   # kernel-variant switches exist only in the development build (build.py --dev)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from densematchingbenchmark_amd import _lib, ops

dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "4"))
D, H, W = [int(v) for v in os.environ.get("KB_SHAPE", "48,136,240").split(",")]
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


def padded_case(d, h, w, name):
    """conv4 + conv5 of the hourglass as the module runs them when the rows are not a 16-byte multiple (W % 4 == 2): rows padded
    with zero columns (Hourglass.forward, ops.padded_rows_applicable)."""
    wpad = (w + 3) // 4 * 4
    x = torch.randn(B, 64, d, h, w, device=dev)
    wp4 = ops.pack_conv3d_weights(torch.randn(64, 64, 3, 3, 3, device=dev) * 0.03)
    wp5 = ops.pack_deconv3d_weights(torch.randn(64, 64, 3, 3, 3, device=dev) * 0.03)
    sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    r = torch.randn(B, 64, 2 * d, 2 * h, 2 * w, device=dev)
    fl = 2.0 * 27 * 64 * 64 * B * d * h * w
    xp = ops.copy_window(x, wpad, 0)
    report(name + ": pad copy + conv4 + clear", timeit(lambda: ops.zero_columns_(ops.conv3d_k3(ops.copy_window(x, wpad, 0), wp4, 64, sc, sh, None, 1, True), w)), fl)
    report(name + ": conv5 +res (Wout = %d)" % (2 * w), timeit(lambda: ops.deconv3d_k3s2(xp, wp5, 64, sc, sh, r, True, out_width=2 * w)), fl)


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


print("B =", B, " full-resolution volume D, H, W =", (D, H, W))
# the chip clocks by its power budget: bring it to a steady state first, and run the list twice
_x = torch.randn(B, 32, D, H, W, device=dev)
_wp = ops.pack_conv3d_weights(torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03)
for _ in range(150):
    ops.conv3d_k3(_x, _wp, 32, None, None, None, 1, False)
torch.cuda.synchronize()
del _x
for _rep in range(1 if diag else 2):
  conv_case(32, 32, 1, D, H, W, "classif / dres 32->32 full (dominant)")
  conv_case(32, 32, 1, D, H, W, "classif / dres 32->32 full (dominant)", res=True)
  if not diag:
    for cand, name in ((1, "48 x 4 row pairs"), (2, "24 x 8 row quads"), (3, "32 x 4 row pairs")):   # development option 19: tile candidate
      lib.dmb_dev_set_option(19, cand)
      conv_case(32, 32, 1, D, H, W, "  32->32 full on %s" % name)
    for cand, name in ((1, "64-voxel runs"), (2, "40 x 4 row quads"), (3, "24 x 4 row quads"), (4, "32 x 4 row quads")):
      lib.dmb_dev_set_option(19, cand)
      conv_case(64, 64, 1, D // 2, H // 2, W // 2, "  64->64 half on %s" % name, res=True)
      conv_case(64, 64, 1, D // 4, H // 4, W // 4, "  64->64 quarter on %s" % name)
    lib.dmb_dev_set_option(19, 0)
  conv_case(32, 64, 2, D, H, W, "conv1 s2 32->64 full->half")
  conv_case(64, 64, 1, D // 2, H // 2, W // 2, "conv2 s1 64->64 half", res=True)
  conv_case(64, 64, 2, D // 2, H // 2, W // 2, "conv3 s2 64->64 half->quarter")
  conv_case(64, 64, 1, D // 4, H // 4, W // 4, "conv4 s1 64->64 quarter")
  deconv_case(64, 64, D // 4, H // 4, W // 4, "conv5 deconv 64->64 quarter->half", res=True)
  if (W // 4) % 4 == 2:
    padded_case(D // 4, H // 4, W // 4, "quarter level, rows padded to %d" % ((W // 4 + 3) // 4 * 4))
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv 64->32 half->full", res=False)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv 64->32 half->full", res=True)
  lib.dmb_dev_set_option(13, 1)   # A/B: quarter-resolution stride-1 layer on 4 x 4 x 60 boxes (216 workgroups) instead of 64-voxel runs (768)
  conv_case(64, 64, 1, D // 4, H // 4, W // 4, "conv4 s1 64->64 quarter (box tiles)")
  lib.dmb_dev_set_option(13, 0)
  lib.dmb_dev_set_option(10, 1)   # A/B: stride-2 layers on four-wave workgroups (two waves per SIMD)
  conv_case(32, 64, 2, D, H, W, "conv1 s2 32->64 (4-wave workgroups)")
  conv_case(64, 64, 2, D // 2, H // 2, W // 2, "conv3 s2 64->64 (4-wave workgroups)")
  lib.dmb_dev_set_option(10, 0)
  for run in (1, 4, 16):           # (option 16 = log2(run) + 1)
              # A/B: zy item order in groups of `run` tiles (default: deconv3d_zy.hip ZY_RUN)
    lib.dmb_dev_set_option(16, run.bit_length())
    deconv_case(64, 64, D // 4, H // 4, W // 4, "conv5 deconv (zy groups of %d tiles)" % run, res=True)
    deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv (zy groups of %d tiles)" % run, res=True)
  lib.dmb_dev_set_option(16, 0)
  lib.dmb_dev_set_option(20, 1)    # A/B: ONE class-major item list over the whole layer (the round-3 order)
  deconv_case(64, 64, D // 4, H // 4, W // 4, "conv5 deconv (one class-major list)", res=True)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv (one class-major list)", res=False)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv (one class-major list)", res=True)
  lib.dmb_dev_set_option(20, 0)
  lib.dmb_dev_set_option(4, 1)   # A/B: the two-parity form (both y parities per item, two workgroups per CU)
  deconv_case(64, 64, D // 4, H // 4, W // 4, "conv5 deconv (items with both y parities)", res=True)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv (items with both y parities)", res=False)
  deconv_case(64, 32, D // 2, H // 2, W // 2, "conv6 deconv (items with both y parities)", res=True)
  lib.dmb_dev_set_option(4, 0)
