"""Per-kernel micro-benchmark at the BASELINE cfg2 shapes (development aid; bench.py is the judged entry)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import ops

dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "4"))
D, H, W = 48, 136, 240


def timeit(fn, n=5, warm=2):
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


def conv_case(Ci, Co, stride, d, h, w, name):
    x = torch.randn(B, Ci, d, h, w, device=dev)
    wt = torch.randn(Co, Ci, 3, 3, 3, device=dev) * 0.03
    wp = ops.pack_conv3d_weights(wt)
    sc = torch.ones(Co, device=dev); sh = torch.zeros(Co, device=dev)
    ms = timeit(lambda: ops.conv3d_k3(x, wp, Co, sc, sh, None, stride, True))
    do, ho, wo = (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1
    fl = 2.0 * 27 * Ci * Co * B * do * ho * wo
    print("%-28s %8.3f ms  %7.2f TFLOP/s  (%.1f%% of 157.3)" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100), flush=True)


def deconv_case(Ci, Co, d, h, w, name):
    x = torch.randn(B, Ci, d, h, w, device=dev)
    wt = torch.randn(Ci, Co, 3, 3, 3, device=dev) * 0.03
    wp = ops.pack_deconv3d_weights(wt)
    sc = torch.ones(Co, device=dev); sh = torch.zeros(Co, device=dev)
    ms = timeit(lambda: ops.deconv3d_k3s2(x, wp, Co, sc, sh, None, True))
    fl = 2.0 * 27 * Ci * Co * B * d * h * w
    print("%-28s %8.3f ms  %7.2f TFLOP/s  (%.1f%% of 157.3)" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100), flush=True)


def bw_case(name, fn, nbytes):
    ms = timeit(fn, n=40, warm=10)
    print("%-28s %8.3f ms  %7.1f GB/s  (%.1f%% of 8000)" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000 * 100), flush=True)


print("B =", B)
from densematchingbenchmark_amd import _lib
_l = _lib.load()
if hasattr(_l, "dmb_dev_set_option"):
    for v in (1, 0, 1, 0):
        _l.dmb_dev_set_option(2, v)
        conv_case(32, 32, 1, D, H, W, "conv s1 32->32 full flattened=%d" % v)
    _l.dmb_dev_set_option(2, 0)
x = torch.randn(B, 32, D, H, W, device=dev); wt = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03
wp = ops.pack_conv3d_weights(wt); sc = torch.ones(32, device=dev); sh = torch.zeros(32, device=dev); rs = torch.randn(B, 32, D, H, W, device=dev)
ms = timeit(lambda: ops.conv3d_k3(x, wp, 32, sc, sh, rs, 1, True))
print("conv s1 32->32 full +residual   %8.3f ms" % ms, flush=True)
del x, rs
conv_case(32, 32, 1, D, H, W, "conv s1 32->32 full")
for _ci in (32, 64):   # experimental opt-in split kernel (FP32-equivalent TFLOP/s: algorithmic flops / time)
    _x = torch.randn(B, _ci, D, H, W, device=dev); _w = torch.randn(32, _ci, 3, 3, 3, device=dev) * 0.03
    _wp = ops.pack_conv3d_x6_weights(_w); _sc = torch.ones(32, device=dev); _sh = torch.zeros(32, device=dev)
    _ms = timeit(lambda: ops.conv3d_k3_x6(_x, _wp, 32, _sc, _sh, None, True))
    _fl = 2.0 * 27 * _ci * 32 * B * D * H * W
    print("%-28s %8.3f ms  %7.2f TFLOP/s FP32-equivalent" % ("conv s1 %d->32 bf16x6 split" % _ci, _ms, _fl / _ms / 1e9), flush=True)
    del _x
conv_case(64, 32, 1, D, H, W, "conv s1 64->32 full")
conv_case(32, 64, 2, D, H, W, "conv s2 32->64 full->half")
conv_case(64, 64, 1, D // 2, H // 2, W // 2, "conv s1 64->64 half")
conv_case(64, 64, 2, D // 2, H // 2, W // 2, "conv s2 64->64 half->quarter")
conv_case(64, 64, 1, D // 4, H // 4, W // 4, "conv s1 64->64 quarter")
deconv_case(64, 64, D // 4, H // 4, W // 4, "deconv 64->64 quarter->half")
deconv_case(64, 32, D // 2, H // 2, W // 2, "deconv 64->32 half->full")

x = torch.randn(B, 32, D, H, W, device=dev)
w1 = torch.randn(1, 32, 3, 3, 3, device=dev)
bw_case("conv c1 32->1 full", lambda: ops.conv3d_k3_c1(x, w1, 0.0, None), x.numel() * 4 + B * D * H * W * 4)
L = torch.randn(B, 32, H, W, device=dev); R = torch.randn(B, 32, H, W, device=dev)
idx = ops.disp_index_list(48, 0, 1)
bw_case("cat_fms", lambda: ops.cat_fms(L, R, idx), B * 64 * D * H * W * 4 + 2 * L.numel() * 4)
# the sample-based builders (cat_fms.py:51-82): per-plane and per-pixel samples
fs = ops.fast_disp_samples(48, 0, 1).to(dev)
bw_case("fast_cat_fms (plane samples)", lambda: ops.fast_cat_fms(L, R, fs), B * 64 * D * H * W * 4 + 2 * L.numel() * 4)
fpp = fs.view(1, D, 1, 1) + 0.5 * torch.rand(B, D, H, W, device=dev)      # smooth per-pixel samples, as a sampler produces
bw_case("fast_cat_fms (per-pixel samples)", lambda: ops.fast_cat_fms(L, R, fpp), B * 64 * D * H * W * 4 + 2 * L.numel() * 4 + fpp.numel() * 4)
bw_case("fast_dif_fms (per-pixel samples)", lambda: ops.fast_dif_fms(L, R, fpp), B * 32 * D * H * W * 4 + 2 * L.numel() * 4 + fpp.numel() * 4)
del fpp
# GwcNet volume (BASELINE configs[2]): 40-group correlation of 320-channel features (MFMA inner products) + 2 x 12 concat
lg, rg = torch.randn(B, 320, H, W, device=dev), torch.randn(B, 320, H, W, device=dev)
gout = torch.empty(B, 64, D, H, W, device=dev)
bw_case("gwc 320ch/40 groups (cfg3)", lambda: ops.gwc_fms(lg, rg, idx, 40, out=gout, out_ch_offset=0),
        2 * lg.numel() * 4 + B * 40 * D * H * W * 4)
del lg, rg, gout
c = torch.randn(B, D, H, W, device=dev)
bw_case("trilinear x4", lambda: ops.trilinear_ac(c, (192, 544, 960)), B * 192 * 544 * 960 * 4 + c.numel() * 4)
big = torch.randn(B, 192, 544, 960, device=dev)
vals = ops.disp_sample_values(192, 0, 1)
bw_case("soft_argmin D=192", lambda: ops.soft_argmin(big, vals, 1.0, True), big.numel() * 4 + B * 544 * 960 * 4)
bw_case("fused trilinear+softargmin", lambda: ops.trilinear_soft_argmin(c, (192, 544, 960), vals, 1.0), c.numel() * 4 + B * 544 * 960 * 4)

# training-side losses (first part) at the full cost-volume size
gt = torch.rand(B, 1, 544, 960, device=dev) * 190 + 1
var = torch.rand(B, 1, 544, 960, device=dev) + 0.5
out, stats = ops.stereo_focal_loss_fwd(big, gt, var, vals, 0, 192, 0, 191, 5.0)
bw_case("focal loss fwd D=192", lambda: ops.stereo_focal_loss_fwd(big, gt, var, vals, 0, 192, 0, 191, 5.0), big.numel() * 4)
go = torch.ones(1, device=dev)
bw_case("focal loss bwd D=192", lambda: ops.stereo_focal_loss_bwd(big, gt, var, vals, stats, out, go, 0, 192, 0, 191, 5.0, True), big.numel() * 8)
