"""Backbone (2-D) micro-benchmark at BASELINE cfg2 image size: 4 stereo pairs = 8 images of 544x960 (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import ops, synthetic
from densematchingbenchmark_amd.modeling.stereo.backbones import PSMNetBackbone

dev = torch.device("cuda:0")
if os.environ.get("KB_OPT18"):     # development library: 1 = always the default tile height, 2 = always two rows per wave
    from densematchingbenchmark_amd import _lib
    _lib.load().dmb_dev_set_option(18, int(os.environ["KB_OPT18"]))
NB = 2 * int(os.environ.get("KB_B", "4"))
H, W = 544, 960


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


def case(name, Ci, Co, k, stride, dil, h, w, count):
    x = torch.randn(NB, Ci, h, w, device=dev)
    wt = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    wp = ops.pack_conv2d_weights(wt)
    sc = torch.ones(Co, device=dev); sh = torch.zeros(Co, device=dev)
    ms = timeit(lambda: ops.conv2d(x, wp, Co, k, stride, dil, sc, sh, None, True))
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    fl = 2.0 * k * k * Ci * Co * NB * ho * wo
    print("%-34s x%-2d %7.3f ms  %7.2f TFLOP/s  (%4.1f%% of 157.3)  total %7.3f ms" % (name, count, ms, fl / ms / 1e9, fl / ms / 1e9 / 1.573, ms * count), flush=True)
    return ms * count, fl * count


print("images =", NB)
tot_ms = tot_fl = 0.0
for args in [("firstconv.0 3->32 k3 s2", 3, 32, 3, 2, 1, H, W, 1),
             ("32->32 k3 (firstconv, layer1) /2", 32, 32, 3, 1, 1, H // 2, W // 2, 8),
             ("layer2.0.conv1 32->64 k3 s2", 32, 64, 3, 2, 1, H // 2, W // 2, 1),
             ("layer2.0.down 32->64 k1 s2", 32, 64, 1, 2, 1, H // 2, W // 2, 1),
             ("64->64 k3 (layer2) /4", 64, 64, 3, 1, 1, H // 4, W // 4, 31),
             ("layer3.0.conv1 64->128 k3", 64, 128, 3, 1, 1, H // 4, W // 4, 1),
             ("layer3.0.down 64->128 k1", 64, 128, 1, 1, 1, H // 4, W // 4, 1),
             ("128->128 k3 (layer3) /4", 128, 128, 3, 1, 1, H // 4, W // 4, 5),
             ("128->128 k3 dil2 (layer4) /4", 128, 128, 3, 1, 2, H // 4, W // 4, 6),
             ("lastconv.0 320->128 k3", 320, 128, 3, 1, 1, H // 4, W // 4, 1),
             ("lastconv.1 128->32 k1", 128, 32, 1, 1, 1, H // 4, W // 4, 1)]:
    m, f = case(*args)
    tot_ms += m; tot_fl += f
print("sum of conv layers: %.3f ms, %.1f GFLOP -> %.1f TFLOP/s" % (tot_ms, tot_fl / 1e9, tot_fl / tot_ms / 1e9))

bb = PSMNetBackbone(3, True).eval()
synthetic.init_params_(bb, seed=8, classif_gain=1.0)
bb = bb.to(dev)
l = torch.randn(NB // 2, 3, H, W, device=dev); r = torch.randn(NB // 2, 3, H, W, device=dev)
with torch.no_grad():
    ms = timeit(lambda: bb(l, r), n=5, warm=2)
print("PSMNetBackbone forward (%d pairs): %.3f ms  -> %.1f TFLOP/s on the conv flops" % (NB // 2, ms, tot_fl / ms / 1e9))
x = torch.randn(NB, 320, H // 4, W // 4, device=dev)
for k in (64, 32, 16, 8):
    ms = timeit(lambda: ops.avgpool2d(x, k, in_window=(64, 128)))
    print("avgpool k=%d: %.3f ms (%.0f GB/s)" % (k, ms, NB * 128 * (H // 4) * (W // 4) * 4 / ms / 1e6))
p = torch.randn(NB, 32, 17, 30, device=dev)
ms = timeit(lambda: ops.bilinear_ac(p, (H // 4, W // 4), out=x, out_ch_offset=192))
print("bilinear -> /4: %.3f ms (%.0f GB/s)" % (ms, NB * 32 * (H // 4) * (W // 4) * 4 / ms / 1e6))
