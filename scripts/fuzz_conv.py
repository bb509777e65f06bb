"""Randomised shape sweep of the convolution entry points against torch CPU (development aid; a failure prints the case)."""
import sys, os, math, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from densematchingbenchmark_amd import ops

dev = torch.device("cuda:0")
rng = random.Random(int(os.environ.get("FUZZ_SEED", "1")))
N = int(os.environ.get("FUZZ_N", "160"))
bad = 0


def rnd(shape, scale=1.0):
    return torch.randn(shape) * scale


for it in range(N):
    kind = rng.choice(["s1", "s1", "s2", "deconv", "c2d", "c2d", "x6", "c1", "gwc", "gwc", "catfirst", "padlevel"])
    B = rng.choice([1, 1, 2])
    try:
        if kind in ("s1", "s2", "deconv", "x6"):
            D, H = rng.randint(1, 9), rng.randint(1, 11)
            W = rng.choice([rng.randint(1, 70), 24, 40, 48, 60, 60, 72, 80, 96, 120])
            Ci = rng.choice([1, 3, 7, 8, 16, 32, 33, 64])
            Co = rng.choice([32, 64] if kind != "s1" else [32, 64, 128])
            relu = rng.choice([False, True, "pre"])
            use_res = rng.random() < 0.5
            sc, sh = 0.5 + torch.rand(Co), torch.rand(Co) - 0.5
            x = rnd((B, Ci, D, H, W))
            if kind == "deconv":
                w = rnd((Ci, Co, 3, 3, 3), 1.0 / math.sqrt(Ci * 27 / 8))
                y = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1)
            else:
                if kind == "x6":
                    W = rng.choice([24, 48, 72, 96]) if Co == 64 else rng.choice([48, 96])
                    x = rnd((B, Ci, D, H, W))
                w = rnd((Co, Ci, 3, 3, 3), 1.0 / math.sqrt(Ci * 27))
                y = F.conv3d(x, w, None, stride=2 if kind == "s2" else 1, padding=1)
            y = y * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
            res = rnd(y.shape) if use_res else None
            if relu == "pre":
                y = F.relu(y)
            if res is not None:
                y = y + res
            if relu is True:
                y = F.relu(y)
            args = (sc.to(dev), sh.to(dev), res.to(dev) if res is not None else None)
            if kind == "deconv":
                got = ops.deconv3d_k3s2(x.to(dev), ops.pack_deconv3d_weights(w.to(dev)), Co, *args, relu)
            elif kind == "x6":
                got = ops.conv3d_k3_x6(x.to(dev), ops.pack_conv3d_x6_weights(w.to(dev)), Co, *args, relu)
            else:
                got = ops.conv3d_k3(x.to(dev), ops.pack_conv3d_weights(w.to(dev)), Co, *args, 2 if kind == "s2" else 1, relu)
            desc = (kind, B, Ci, Co, D, H, W, relu, use_res)
        elif kind == "padlevel":   # the hourglass's deepest level on rows padded to a 16-byte multiple (W % 4 == 2), Hourglass.forward
            Co = rng.choice([32, 64])
            D, H, W = rng.randint(1, 7), rng.randint(1, 12), 4 * rng.randint(2, 40) + 2
            x = rnd((B, 64, D, H, W))
            w4, w5 = rnd((64, 64, 3, 3, 3), 1.0 / math.sqrt(64 * 27)), rnd((64, Co, 3, 3, 3), 1.0 / math.sqrt(64 * 27 / 8))
            res = rnd((B, Co, 2 * D, 2 * H, 2 * W))
            y = F.relu(F.conv_transpose3d(F.relu(F.conv3d(x, w4, None, padding=1)), w5, None, stride=2, padding=1, output_padding=1) + res)
            mid = ops.conv3d_k3(ops.copy_window(x.to(dev), (W + 3) // 4 * 4, 0), ops.pack_conv3d_weights(w4.to(dev)), 64, None, None, None, 1, True)
            ops.zero_columns_(mid, W)
            got = ops.deconv3d_k3s2(mid, ops.pack_deconv3d_weights(w5.to(dev)), Co, None, None, res.to(dev), True, out_width=2 * W)
            desc = (kind, B, Co, D, H, W)
        elif kind == "c1":       # 32 -> 1 head: the 16-byte form (W % 4 == 0) and the dword form
            D, H = rng.randint(1, 19), rng.randint(1, 19)
            W = rng.choice([rng.randint(1, 130), 60, 64, 120, 124, 240])
            Ci = rng.choice([1, 2, 5, 32])
            x, w = rnd((B, Ci, D, H, W)), rnd((1, Ci, 3, 3, 3), 1.0 / math.sqrt(Ci * 27))
            res = rnd((B, 1, D, H, W)) if rng.random() < 0.5 else None
            y = F.conv3d(x, w, None, padding=1) + 0.25
            if res is not None:
                y = y + res
            got = ops.conv3d_k3_c1(x.to(dev), w.to(dev), 0.25, res.to(dev) if res is not None else None)
            desc = (kind, B, Ci, D, H, W, res is not None)
        elif kind == "gwc":      # group-wise correlation: matrix-core form (0 <= d <= 64) and the fallback
            G = rng.choice([1, 2, 5, 8])
            CG = rng.choice([2, 4, 8, 16])
            H, W = rng.randint(1, 7), rng.choice([rng.randint(2, 300), 64, 240, 256, 260, 512])
            start, dil = rng.choice([0, 0, 0, -3, 2]), rng.choice([1, 1, 2])
            md = rng.randint(1, 70)
            idx = ops.disp_index_list(md, start, dil)
            L, R = rnd((B, G * CG, H, W)), rnd((B, G * CG, H, W))
            y = torch.zeros(B, G, len(idx), H, W)
            for k, d in enumerate(idx):
                if abs(d) < W:
                    xs = slice(max(d, 0), W + min(d, 0))
                    xt = slice(max(-d, 0), W - max(d, 0))
                    y[:, :, k, :, xs] = (L[..., xs] * R[..., xt]).view(B, G, CG, H, -1).mean(2)
            got = ops.gwc_fms(L.to(dev), R.to(dev), idx, G)
            desc = (kind, B, G, CG, H, W, md, start, dil)
        elif kind == "catfirst":  # first layer on the concatenation / difference volume without the volume
            C, Co = rng.choice([4, 16, 32]), 32
            D = rng.choice([4, 8, 12, 24])
            H, W = rng.randint(1, 9), rng.choice([D + 8, D + 12, 64, 72, 100, 120])
            kd = rng.choice(["cat", "dif"])
            L, R = rnd((B, C, H, W)), rnd((B, C, H, W))
            idx = list(range(D))
            Cin = 2 * C if kd == "cat" else C
            w = rnd((Co, Cin, 3, 3, 3), 1.0 / math.sqrt(Cin * 27))
            sc, sh = 0.5 + torch.rand(Co), torch.rand(Co) - 0.5
            vol = (ops.cat_fms if kd == "cat" else ops.dif_fms)(L.to(dev), R.to(dev), idx).cpu()
            y = F.relu(F.conv3d(vol, w, None, padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1))
            if not ops.catconv_applicable(L.to(dev), R.to(dev), idx, Co):
                continue
            got = ops.catconv_first(L.to(dev), R.to(dev), D, ops.catconv_pack(w.to(dev), kd), sc.to(dev), sh.to(dev), True)
            desc = (kind, kd, B, C, D, H, W)
        else:
            H = rng.randint(1, 40)
            W = rng.choice([rng.randint(1, 100), 16, 48, 52, 96, 100])
            k, stride, dil = rng.choice([(1, 1, 1), (3, 1, 1), (3, 1, 2), (3, 2, 1), (1, 2, 1), (5, 2, 1), (3, 1, 4), (3, 1, 8)])
            Co = rng.choice([1, 32] if (dil > 2 or k == 5) else ([32, 64] if stride == 2 else [1, 32, 64, 128]))
            Ci = rng.choice([1, 3, 4, 8, 20, 32, 64, 128])
            relu = rng.random() < 0.5
            use_res = rng.random() < 0.5
            sc, sh = 0.5 + torch.rand(Co), torch.rand(Co) - 0.5
            x = rnd((B, Ci, H, W))
            w = rnd((Co, Ci, k, k), 1.0 / math.sqrt(Ci * k * k))
            y = F.conv2d(x, w, None, stride=stride, padding=dil * (k // 2), dilation=dil) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
            res = rnd(y.shape) if use_res else None
            if res is not None:
                y = y + res
            if relu:
                y = F.relu(y)
            got = ops.conv2d(x.to(dev), ops.pack_conv2d_weights(w.to(dev)), Co, k, stride, dil, sc.to(dev), sh.to(dev),
                             res.to(dev) if res is not None else None, relu)
            desc = (kind, B, Ci, Co, H, W, k, stride, dil, relu, use_res)
        err = (got.cpu() - y).abs().max().item() if y.numel() else 0.0
        if got.shape != y.shape or not (err <= 3e-5):
            bad += 1
            print("FAIL", desc, "err", err, "shape", tuple(got.shape), tuple(y.shape), flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("EXC", kind, repr(e)[:200], flush=True)
print("cases", N, "failures", bad)
