"""Randomised shape / batch sweep of the convolution entry points through the RELEASE library, against torch CPU FP32.

Which kernel form a launch takes is a cost estimate over its shape AND batch (tile candidates of the stride-1 / stride-2 / transposed
kernels, the split-K forms for launches that leave the chip idle, the row-padded deepest level, 16-byte and dword paths): fixed
parametrisations cannot walk that decision space, so this module draws ~300 seeded cases -- B in {1, 2, 3, 4, 8}, widths with
W % 4 != 0, the KITTI widths 78 / 156 / 312, D < 4, 1 .. 64 input channels -- of
  stride-1 / stride-2 / transposed 3x3x3 units (layers/basic_layers.py:68-100,160-177 with folded BatchNorm, skip, both ReLU orders),
  the 32 -> 1 heads (PSMNet.py:46-54), the 2-D backbone convolutions (basic_layers.py:12-66), the group-wise correlation,
  the volume-free first layer (aggregators/PSMNet.py:31-35 on cat_fms) and the row-padded hourglass level,
and, for a subset, evaluates the same items at two batch sizes (other tile picks, other kernel forms) and compares them with each
other.  Seeds are fixed; a failure prints every failing case.  (The body is scripts/fuzz_conv.py's, which remains the open-ended
development form.)"""
import math
import os
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 3e-5           # one layer, unit-variance inputs, fan-in-normalised weights (the fixed-shape tests use 2e-5)
CASES_PER_CHUNK = 25
CHUNKS = 12
FLOP_BUDGET = 1.5e9   # per case, so that the CPU reference of a chunk takes a few seconds
SEED_BASE = int(os.environ.get("DMB_FUZZ_SEED_BASE", "60000"))   # (fixed by default; another base = another 300 cases, for bug hunts)


def _rnd(shape, g, scale=1.0):
    return torch.randn(shape, generator=g) * scale


def _shrink(B, D, H, W, Ci, Co, taps=27):
    """Keep the CPU reference cheap: drop batch items, then planes."""
    while 2.0 * taps * Ci * Co * B * D * H * W > FLOP_BUDGET and (B > 1 or D > 1):
        if B > 1:
            B -= 1
        else:
            D -= 1
    return B, D


def _unit(x, w, sc, sh, res, relu, kind):
    if kind == "deconv":
        y = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1)
    else:
        y = F.conv3d(x, w, None, stride=2 if kind == "s2" else 1, padding=1)
    y = y * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    if relu == "pre":
        y = F.relu(y)
    if res is not None:
        y = y + res
    if relu is True:
        y = F.relu(y)
    return y


def _case(seed, dev, ops):
    """-> (description, max abs error, tolerance) of one random case; None if the drawn shape does not apply."""
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(seed)
    kind = rng.choice(["s1", "s1", "s2", "s2", "deconv", "deconv", "c2d", "c2d", "c1", "c1", "gwc", "catfirst", "padlevel", "twobatch", "twobatch"])
    B = rng.choice([1, 1, 2, 3, 4, 8])
    if kind in ("s1", "s2", "deconv", "twobatch"):
        sub = kind if kind != "twobatch" else rng.choice(["s1", "s2", "deconv", "c1"])
        D, H = rng.choice([1, 2, 3, rng.randint(4, 9)]), rng.randint(1, 17)
        W = rng.choice([rng.randint(1, 70), 16, 24, 32, 40, 48, 60, 64, 72, 78, 80, 96, 120, 156, 312])
        Ci = rng.choice([1, 3, 7, 8, 16, 32, 32, 33, 48, 64, 64])
        Co = rng.choice([32, 64] if sub != "s1" else [32, 64, 64, 128])
        if sub == "c1":
            Ci, Co = rng.choice([2, 5, 32, 32]), 1
        B, D = _shrink(B, D, H, W, Ci, max(Co, 8))
        relu = rng.choice([False, True, "pre"])
        x = _rnd((B, Ci, D, H, W), g)
        if sub == "c1":
            w = _rnd((1, Ci, 3, 3, 3), g, 1.0 / math.sqrt(Ci * 27))
            res = _rnd((B, 1, D, H, W), g) if rng.random() < 0.5 else None
            ref = F.conv3d(x, w, None, padding=1) + 0.25 + (res if res is not None else 0.0)
            run = lambda xb, rb: ops.conv3d_k3_c1(xb, w.to(dev), 0.25, rb)   # noqa: E731
        else:
            sc, sh = 0.5 + torch.rand(Co, generator=g), torch.rand(Co, generator=g) - 0.5
            if sub == "deconv":
                w = _rnd((Ci, Co, 3, 3, 3), g, 1.0 / math.sqrt(Ci * 27 / 8))
            else:
                w = _rnd((Co, Ci, 3, 3, 3), g, 1.0 / math.sqrt(Ci * 27))
            ref = _unit(x, w, sc, sh, None, False, sub)
            res = _rnd(ref.shape, g) if rng.random() < 0.5 else None
            ref = _unit(x, w, sc, sh, res, relu, sub)
            if sub == "deconv":
                wp = ops.pack_deconv3d_weights(w.to(dev))
                run = lambda xb, rb: ops.deconv3d_k3s2(xb, wp, Co, sc.to(dev), sh.to(dev), rb, relu)   # noqa: E731
            else:
                wp = ops.pack_conv3d_weights(w.to(dev))
                run = lambda xb, rb: ops.conv3d_k3(xb, wp, Co, sc.to(dev), sh.to(dev), rb, 2 if sub == "s2" else 1, relu)   # noqa: E731
        got = run(x.to(dev), res.to(dev) if res is not None else None)
        desc = (kind, sub, B, Ci, Co, D, H, W, relu, res is not None)
        err = (got.cpu() - ref).abs().max().item() if ref.numel() else 0.0
        if kind == "twobatch" and B > 1:
            # the same items one at a time: another tile pick / kernel form per launch; the two evaluations of an item may differ by
            # FP32 roundings of the sum (split-K forms), never by more than the distance either keeps from the reference
            for i in range(B):
                one = run(x[i:i + 1].to(dev), res[i:i + 1].to(dev) if res is not None else None)
                err = max(err, (one - got[i:i + 1]).abs().max().item())
        return desc, err, TOL, tuple(got.shape) == tuple(ref.shape)
    if kind == "padlevel":   # the hourglass's deepest level on rows padded to a 16-byte multiple (W % 4 == 2): Hourglass.forward
        Co = rng.choice([32, 64])
        D, H, W = rng.randint(1, 5), rng.randint(1, 12), 4 * rng.randint(2, 30) + 2
        B, D = _shrink(B, D, H, W, 64, 64 + 8 * Co)
        x = _rnd((B, 64, D, H, W), g)
        w4, w5 = _rnd((64, 64, 3, 3, 3), g, 1.0 / math.sqrt(64 * 27)), _rnd((64, Co, 3, 3, 3), g, 1.0 / math.sqrt(64 * 27 / 8))
        res = _rnd((B, Co, 2 * D, 2 * H, 2 * W), g)
        ref = F.relu(F.conv_transpose3d(F.relu(F.conv3d(x, w4, None, padding=1)), w5, None, stride=2, padding=1, output_padding=1) + res)
        mid = ops.conv3d_k3(ops.copy_window(x.to(dev), (W + 3) // 4 * 4, 0), ops.pack_conv3d_weights(w4.to(dev)), 64, None, None, None, 1, True)
        ops.zero_columns_(mid, W)
        got = ops.deconv3d_k3s2(mid, ops.pack_deconv3d_weights(w5.to(dev)), Co, None, None, res.to(dev), True, out_width=2 * W)
        return (kind, B, Co, D, H, W), (got.cpu() - ref).abs().max().item(), TOL, tuple(got.shape) == tuple(ref.shape)
    if kind == "c1":
        D, H = rng.randint(1, 19), rng.randint(1, 19)
        W = rng.choice([rng.randint(1, 130), 60, 64, 120, 124, 128, 240])
        Ci = rng.choice([1, 2, 5, 32, 32])
        B, D = _shrink(B, D, H, W, Ci, 8)
        x, w = _rnd((B, Ci, D, H, W), g), _rnd((1, Ci, 3, 3, 3), g, 1.0 / math.sqrt(Ci * 27))
        res = _rnd((B, 1, D, H, W), g) if rng.random() < 0.5 else None
        ref = F.conv3d(x, w, None, padding=1) + 0.25 + (res if res is not None else 0.0)
        got = ops.conv3d_k3_c1(x.to(dev), w.to(dev), 0.25, res.to(dev) if res is not None else None)
        return (kind, B, Ci, D, H, W, res is not None), (got.cpu() - ref).abs().max().item(), TOL, tuple(got.shape) == tuple(ref.shape)
    if kind == "gwc":        # group-wise correlation: matrix-core form (0 <= d <= 64) and the fallback
        G, CG = rng.choice([1, 2, 5, 8]), rng.choice([2, 4, 8, 16])
        H, W = rng.randint(1, 7), rng.choice([rng.randint(2, 300), 64, 240, 256, 260, 312])
        start, dil, md = rng.choice([0, 0, 0, -3, 2]), rng.choice([1, 1, 2]), rng.randint(1, 70)
        B = min(B, 2)
        idx = ops.disp_index_list(md, start, dil)
        L, R = _rnd((B, G * CG, H, W), g), _rnd((B, G * CG, H, W), g)
        ref = torch.zeros(B, G, len(idx), H, W)
        for k, d in enumerate(idx):
            if abs(d) < W:
                xs, xt = slice(max(d, 0), W + min(d, 0)), slice(max(-d, 0), W - max(d, 0))
                ref[:, :, k, :, xs] = (L[..., xs] * R[..., xt]).view(B, G, CG, H, -1).mean(2)
        got = ops.gwc_fms(L.to(dev), R.to(dev), idx, G)
        return (kind, B, G, CG, H, W, md, start, dil), (got.cpu() - ref).abs().max().item(), TOL, tuple(got.shape) == tuple(ref.shape)
    if kind == "catfirst":   # first layer on the concatenation / difference volume without the volume
        C, Co, D = rng.choice([4, 16, 32]), 32, rng.choice([4, 8, 12, 16, 24])
        H, W = rng.randint(1, 9), rng.choice([D + 8, D + 12, 64, 72, 100, 120, 128])
        B = min(B, 3)
        kd = rng.choice(["cat", "dif"])
        L, R = _rnd((B, C, H, W), g), _rnd((B, C, H, W), g)
        idx = list(range(D))
        Cin = 2 * C if kd == "cat" else C
        w = _rnd((Co, Cin, 3, 3, 3), g, 1.0 / math.sqrt(Cin * 27))
        sc, sh = 0.5 + torch.rand(Co, generator=g), torch.rand(Co, generator=g) - 0.5
        if not ops.catconv_applicable(L.to(dev), R.to(dev), idx, Co):
            return None
        vol = (ops.cat_fms if kd == "cat" else ops.dif_fms)(L.to(dev), R.to(dev), idx).cpu()
        ref = F.relu(F.conv3d(vol, w, None, padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1))
        got = ops.catconv_first(L.to(dev), R.to(dev), D, ops.catconv_pack(w.to(dev), kd), sc.to(dev), sh.to(dev), True)
        return (kind, kd, B, C, D, H, W), (got.cpu() - ref).abs().max().item(), TOL, tuple(got.shape) == tuple(ref.shape)
    # 2-D backbone convolutions
    H = rng.randint(1, 40)
    W = rng.choice([rng.randint(1, 100), 16, 48, 52, 96, 100, 128])
    k, stride, dil = rng.choice([(1, 1, 1), (3, 1, 1), (3, 1, 1), (3, 1, 2), (3, 2, 1), (1, 2, 1), (5, 2, 1), (3, 1, 4), (3, 1, 8)])
    Co = rng.choice([1, 32] if (dil > 2 or k == 5) else ([32, 64] if stride == 2 else [1, 32, 64, 128]))
    Ci = rng.choice([1, 3, 4, 8, 20, 32, 64, 128])
    B, _ = _shrink(B, 1, H, W, Ci, Co, k * k)
    relu, use_res = rng.random() < 0.5, rng.random() < 0.5
    sc, sh = 0.5 + torch.rand(Co, generator=g), torch.rand(Co, generator=g) - 0.5
    x = _rnd((B, Ci, H, W), g)
    w = _rnd((Co, Ci, k, k), g, 1.0 / math.sqrt(Ci * k * k))
    ref = F.conv2d(x, w, None, stride=stride, padding=dil * (k // 2), dilation=dil) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    res = _rnd(ref.shape, g) if use_res else None
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    got = ops.conv2d(x.to(dev), ops.pack_conv2d_weights(w.to(dev)), Co, k, stride, dil, sc.to(dev), sh.to(dev),
                     res.to(dev) if res is not None else None, relu)
    return (kind, B, Ci, Co, H, W, k, stride, dil, relu, use_res), (got.cpu() - ref).abs().max().item(), TOL, tuple(got.shape) == tuple(ref.shape)


@pytest.mark.parametrize("chunk", range(CHUNKS))
def test_random_shapes_and_batches_against_torch_cpu(dev, chunk):
    from densematchingbenchmark_amd import ops
    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    failures, ran = [], 0
    try:
        for i in range(CASES_PER_CHUNK):
            seed = SEED_BASE + chunk * 1000 + i
            try:
                out = _case(seed, dev, ops)
            except Exception as e:  # noqa: BLE001  (a shape the library refuses is a failure too: every drawn shape is legal)
                failures.append((seed, "EXC", repr(e)[:300]))
                continue
            if out is None:
                continue
            ran += 1
            desc, err, tol, shape_ok = out
            if not shape_ok or not err <= tol:
                failures.append((seed, desc, err))
    finally:
        torch.set_num_threads(threads)
    assert not failures, "%d of %d cases failed:\n%s" % (len(failures), ran, "\n".join(map(str, failures)))
    assert ran >= CASES_PER_CHUNK - 6
