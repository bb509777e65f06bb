"""Development aid: run ONE layer of the cost path a few times (for rocprofv3 counter passes).
    python scripts/kcase.py deconv6|deconv5|s2_1|s2_3|s1q|s1f|s1h|c1|gwc|tri|c2d32|c2d64|c2d128 [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from densematchingbenchmark_amd import _lib, ops

dev = torch.device("cuda:0")
B = int(os.environ.get("KB_B", "4"))
D, H, W = [int(v) for v in os.environ.get("KB_SHAPE", "48,136,240").split(",")]   # full-resolution volume (48,96,312 = KITTI)
# KC_OPTS="16=1,20=0": development options (needs DMB_LIB=dev: the development build of the library)
for kv in filter(None, os.environ.get("KC_OPTS", "").split(",")):
    _lib.load().dmb_dev_set_option(*[int(v) for v in kv.split("=")])
name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = lambda *s: torch.randn(*s, device=dev)
if name in ("deconv6", "deconv5"):
    Ci, Co, d, h, w = (64, 32, D // 2, H // 2, W // 2) if name == "deconv6" else (64, 64, D // 4, H // 4, W // 4)
    x, wp = g(B, Ci, d, h, w), ops.pack_deconv3d_weights(g(Ci, Co, 3, 3, 3) * 0.03)
    sc, sh, r = torch.ones(Co, device=dev), torch.zeros(Co, device=dev), g(B, Co, 2 * d, 2 * h, 2 * w)
    fn = lambda: ops.deconv3d_k3s2(x, wp, Co, sc, sh, r, True)
elif name in ("s2_1", "s2_3", "s1q", "s1f", "s1h"):
    Ci, Co, st, d, h, w = {"s2_1": (32, 64, 2, D, H, W), "s2_3": (64, 64, 2, D // 2, H // 2, W // 2),
                           "s1q": (64, 64, 1, D // 4, H // 4, W // 4), "s1f": (32, 32, 1, D, H, W),
                           "s1h": (64, 64, 1, D // 2, H // 2, W // 2)}[name]
    x, wp = g(B, Ci, d, h, w), ops.pack_conv3d_weights(g(Co, Ci, 3, 3, 3) * 0.03)
    sc, sh = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    fn = lambda: ops.conv3d_k3(x, wp, Co, sc, sh, None, st, True)
elif name == "c1":
    x, w1, r = g(B, 32, D, H, W), g(1, 32, 3, 3, 3), g(B, 1, D, H, W)
    fn = lambda: ops.conv3d_k3_c1(x, w1, 0.0, r)
elif name == "gwc":      # GwcNet volume (BASELINE configs[2]): 40-group correlation of 320-channel features + 2 x 12 concat channels
    lg, rg, lc, rc = g(B, 320, H, W), g(B, 320, H, W), g(B, 12, H, W), g(B, 12, H, W)
    idx = ops.disp_index_list(D, 0, 1)
    out = torch.empty(B, 64, D, H, W, device=dev)
    def fn():
        ops.gwc_fms(lg, rg, idx, 40, out=out, out_ch_offset=0)
        ops.cat_fms_into(lc, rc, idx, out, 40)
elif name == "tri":
    c = g(B, D, H, W)
    vals = ops.disp_sample_values(192, 0, 1)
    fn = lambda: ops.trilinear_ac_soft_argmin(c, (192, 544, 960), vals, 1.0)
elif name in ("c2d64", "c2d128", "c2d32"):   # backbone layers (8 images): layer2 64 -> 64 at 1/4, layer3 128 -> 128 at 1/4, layer1 32 -> 32 at 1/2
    C, h, w = {"c2d64": (64, H, W), "c2d128": (128, H, W), "c2d32": (32, 2 * H, 2 * W)}[name]
    x, wp2 = g(8, C, h, w), ops.pack_conv2d_weights(g(C, C, 3, 3) * 0.05)
    sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    fn = lambda: ops.conv2d(x, wp2, C, 3, 1, 1, sc, sh, None, True)
else:
    raise SystemExit("unknown case " + name)
for _ in range(reps):
    fn()
torch.cuda.synchronize()
