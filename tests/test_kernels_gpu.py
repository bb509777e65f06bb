"""Parity of every C-ABI entry point (through densematchingbenchmark_amd.ops) against the CPU oracle.

Integer / copy / index paths are compared bit-exactly; floating-point kernels with the tolerance written at
each assert (FP32 fma-chain vs the oracle's FP32 evaluation in a different summation order)."""
import math

import pytest
import torch
import numpy as np
import torch.nn.functional as F

from oracle import dmb_oracle as O
from tests._util import golden, maxdiff, sha

pytestmark = pytest.mark.gpu


def _lib_ws_ints():
    from densematchingbenchmark_amd import _lib
    return _lib.DECONV3D_WORKSPACE_BYTES // 4


def _ops():
    from densematchingbenchmark_amd import ops

    return ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _misaligned(t, dev):
    """The same values on the device at an address that is only 4-byte aligned.  The library picks its 16-byte (vector) paths
    from the operands' alignment, so a misaligned operand is how a caller reaches the dword (scalar) paths -- the release library
    has no switch for it (kernel-variant options exist only in the development build, which the tests never load)."""
    buf = torch.empty(t.numel() + 1, dtype=torch.float32, device=dev)
    v = buf[1:].view(t.shape)
    v.copy_(t)
    assert v.data_ptr() % 16 == 4 and v.is_contiguous()
    return v


# ------------------------------------------------------------------------------------------- volumes (bit-exact)
@pytest.mark.parametrize("shape,md,sd,dil", [
    ((1, 1, 3, 4), 5, -2, 2),       # reference known-answer shape (tests/.../test_cat_fms.py:30-40)
    ((2, 32, 16, 32), 8, 0, 1),
    ((1, 5, 7, 61), 12, -3, 1),     # W % 4 != 0 -> scalar path
    ((1, 8, 9, 64), 6, 0, 2),       # dilation 2 -> indices [0, 2, 5]
    ((1, 4, 5, 12), 20, 0, 1),      # disparities beyond the width
])
def test_cat_dif_bit_exact(dev, shape, md, sd, dil):
    ops = _ops()
    L, R = _rand(shape, 1), _rand(shape, 2)
    idx = ops.disp_index_list(md, sd, dil)
    assert idx == O.disp_index_list(md, sd, dil)
    got = ops.cat_fms(L.to(dev), R.to(dev), idx).cpu()
    assert torch.equal(got, O.cat_fms(L, R, md, sd, dil))
    got = ops.dif_fms(L.to(dev), R.to(dev), idx).cpu()
    assert torch.equal(got, O.dif_fms(L, R, md, sd, dil))


def test_cat_known_answer(dev):
    ops = _ops()
    L = torch.arange(1, 13, dtype=torch.float32).view(1, 1, 3, 4)
    R = torch.arange(13, 25, dtype=torch.float32).view(1, 1, 3, 4)
    out = ops.cat_fms(L.to(dev), R.to(dev), ops.disp_index_list(5, -2, 2)).cpu()
    assert out.shape == (1, 2, 3, 3, 4)
    assert out[0, 0, :, 0].tolist() == [[1, 2, 0, 0], [1, 2, 3, 4], [0, 0, 3, 4]]
    assert out[0, 1, :, 0].tolist() == [[15, 16, 0, 0], [13, 14, 15, 16], [0, 0, 13, 14]]


def _fast_case(row):
    shape, D, seed = tuple(int(v) for v in row[:4]), int(row[4]), int(row[5])
    a, b = _rand(shape, seed), _rand(shape, seed + 1000)
    g = torch.Generator().manual_seed(seed + 2000)
    ds = torch.rand((shape[0], D, shape[2], shape[3]), generator=g) * shape[3] * 0.6 - 2.0
    return a, b, ds


def test_fast_mode_volumes_bit_exact(dev):
    """csrc/warp_volume.hip against the REFERENCE's fast_cat_fms / fast_dif_fms outputs (tests/golden/fast_volumes.npz):
    the known-answer case of its test, per-pixel non-integer samples, the builder's own linspace samples -- bit for bit."""
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.cat_fms import CAT_FUNCS
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.dif_fms import DIF_FUNCS
    fast_cat, fast_dif = CAT_FUNCS["fast_mode"], DIF_FUNCS["fast_mode"]
    g = golden("fast_volumes.npz")
    L = torch.arange(1, 13, dtype=torch.float32).view(1, 1, 3, 4).to(dev)
    R = torch.arange(13, 25, dtype=torch.float32).view(1, 1, 3, 4).to(dev)
    assert np.array_equal(fast_cat(L, R, 5, -2, 2).cpu().numpy(), g["ka_cat"])
    ka_samples = torch.linspace(-2, 2, 3).repeat(1, 3, 4, 1).permute(0, 3, 1, 2).contiguous().to(dev)
    assert np.array_equal(fast_cat(L, R, 5, -2, 2, ka_samples).cpu().numpy(), g["ka_cat_samples"])
    assert np.array_equal(fast_dif(L, R, 5, -2, 2).cpu().numpy(), g["ka_dif"])
    for i, row in enumerate(g["cases"]):
        a, b, ds = _fast_case(row)
        c = fast_cat(a.to(dev), b.to(dev), disp_sample=ds.to(dev))
        d = fast_dif(a.to(dev), b.to(dev), disp_sample=ds.to(dev))
        assert sha(c) == str(g["cat_sha_%d" % i]) and sha(d) == str(g["dif_sha_%d" % i])
        assert torch.equal(c.cpu(), O.fast_cat_fms(a, b, disp_sample=ds))
        for p_, key in ((1.0, "dif_norm1_%d"), (2.0, "dif_norm2_%d")):
            n = fast_dif(a.to(dev), b.to(dev), disp_sample=ds.to(dev), normalize=True, p=p_)
            assert maxdiff(n, g[key % i]) <= 1e-5 * max(1.0, float(np.abs(g[key % i]).max()))
        n3 = fast_dif(a.to(dev), b.to(dev), disp_sample=ds.to(dev), normalize=True, p=3.0)
        assert maxdiff(n3, O.fast_dif_fms(a, b, disp_sample=ds, normalize=True, p=3.0)) <= 2e-5 * max(1.0, n3.abs().max().item())
        c = fast_cat(a.to(dev), b.to(dev), 24, -3, 2)
        assert sha(c) == str(g["cat_default_sha_%d" % i])


def test_wrappers_refuse_mismatched_operands(dev):
    """Sizes a kernel indexes with come from ONE operand; the wrappers refuse the others when they disagree instead of
    letting the device read out of bounds."""
    from densematchingbenchmark_amd import _lib
    ops = _ops()
    a, b = torch.randn(1, 4, 6, 16, device=dev), torch.randn(1, 4, 6, 12, device=dev)
    idx = ops.disp_index_list(4, 0, 1)
    for call in (lambda: ops.cat_fms(a, b, idx), lambda: ops.dif_fms(a, b, idx), lambda: ops.gwc_fms(a, b, idx, 2),
                 lambda: ops.correlation1d(a, b, 4), lambda: ops.fast_cat_fms(a, b, torch.zeros(4, device=dev)),
                 lambda: ops.cat_fms(a[0], a[0], idx)):
        with pytest.raises(_lib.DmbLibraryError):
            call()
    x = torch.randn(1, 8, 4, 6, 16, device=dev)
    wp = ops.pack_conv3d_weights(torch.randn(32, 8, 3, 3, 3, device=dev))
    ops.conv3d_k3(x, wp, 32)
    with pytest.raises(_lib.DmbLibraryError):
        ops.conv3d_k3(x, wp, 64)                                            # weights packed for 32 output channels
    with pytest.raises(_lib.DmbLibraryError):
        ops.conv3d_k3(x, wp, 32, scale=torch.ones(8, device=dev))
    with pytest.raises(_lib.DmbLibraryError):
        ops.conv3d_k3_c1(x, torch.randn(1, 4, 3, 3, 3, device=dev))
    cost = torch.randn(1, 4, 6, 16, device=dev)
    with pytest.raises(_lib.DmbLibraryError):
        ops.soft_argmin_sampled(cost, torch.zeros(1, 4, 6, 12, device=dev))
    with pytest.raises(_lib.DmbLibraryError):
        ops.stereo_focal_loss_fwd(cost, torch.zeros(2, 1, 6, 16, device=dev), 1.0, [0.0, 1.0, 2.0, 3.0], 0, 4, 0, 3, 0.0)
    with pytest.raises(_lib.DmbLibraryError):
        ops.map_loss_fwd(torch.zeros(1, 1, 6, 16, device=dev), torch.zeros(1, 1, 6, 12, device=dev), 0, 4, 0)


def test_fast_mode_argument_errors(dev):
    from densematchingbenchmark_amd import _lib
    ops = _ops()
    a = torch.randn(1, 2, 4, 6, device=dev)
    with pytest.raises(_lib.DmbLibraryError):
        ops.fast_cat_fms(a, a, torch.zeros(1, 3, 4, 5, device=dev))     # samples do not match the features
    with pytest.raises(_lib.DmbLibraryError):
        ops.fast_cat_fms(a, a, torch.zeros(1, device=dev))               # D = 1: the reference divides by D - 1
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.cat_fms import fast_cat_fms
    with pytest.raises(NotImplementedError):   # a gradient for the samples exists for per-pixel samples only
        fast_cat_fms(a, a, 4, disp_sample=torch.zeros(3, device=dev).requires_grad_())
    with pytest.raises(_lib.DmbLibraryError):
        ops.fast_fms_bwd(a, a, torch.zeros(3, device=dev), torch.zeros(1, 4, 3, 4, 6, device=dev), wrt_samples=True)
    with pytest.raises(_lib.DmbLibraryError):  # the normalised form's grad_output is [B, D, H, W]
        ops.fast_fms_bwd(a, a, torch.zeros(3, device=dev), torch.zeros(1, 2, 3, 4, 6, device=dev), True, norm_out=torch.zeros(1, 3, 4, 6, device=dev))


def test_fast_volume_builders_backward(dev):
    """Backward of fast_cat_fms / fast_dif_fms (csrc/warp_volume.hip, the sampler's adjoint) through the module-level builders
    under torch.autograd, against gradients of the REFERENCE's builders under its own autograd (tests/golden/
    fast_volumes_grad.npz): per-pixel samples and the builders' linspace samples; and against an FP64 evaluation of the oracle
    (error no larger than 4x the FP32 reference's own)."""
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.cat_fms import fast_cat_fms
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.dif_fms import fast_dif_fms
    g = golden("fast_volumes_grad.npz")
    for i, row in enumerate(g["cases"]):
        sh, D, seed = tuple(int(v) for v in row[:4]), int(row[4]), int(row[5])
        a, b = _rand(sh, seed), _rand(sh, seed + 1000)
        gen = torch.Generator().manual_seed(seed + 2000)
        ds = torch.rand((sh[0], D, sh[2], sh[3]), generator=gen) * sh[3] * 0.6 - 2.0
        for kind, fn in (("cat", fast_cat_fms), ("dif", fast_dif_fms)):
            ch = 2 * sh[1] if kind == "cat" else sh[1]
            for mode in ("pixel", "default"):
                nd = D if mode == "pixel" else 12
                up = _rand((sh[0], ch, nd, sh[2], sh[3]), seed + 3000 + (0 if kind == "cat" else 1))
                L, R = a.to(dev).requires_grad_(), b.to(dev).requires_grad_()
                kw = dict(disp_sample=ds.to(dev)) if mode == "pixel" else dict(max_disp=24, start_disp=-3, dilation=2)
                vol = fn(L, R, **kw)
                vol.backward(up.to(dev))
                okw = dict(disp_sample=ds) if mode == "pixel" else dict(max_disp=24, start_disp=-3, dilation=2)
                t64 = O.fast_volume_grads(a, b, up, kind=kind, dtype=torch.float64, **okw)
                for got, name, truth in ((L.grad, "dL", t64[0]), (R.grad, "dR", t64[1])):
                    ref = torch.as_tensor(g["%s_%s_%s_%d" % (kind, mode, name, i)])
                    scale = max(1.0, ref.abs().max().item())
                    assert (got.cpu() - ref).abs().max().item() <= 2e-5 * scale, (kind, mode, name, i)
                    e_got = (got.cpu().double() - truth).abs().max().item()
                    e_ref = (ref.double() - truth).abs().max().item()
                    assert e_got <= max(4 * e_ref, 1e-6 * scale), (kind, mode, name, i, e_got, e_ref)


def test_fast_volume_builders_backward_through_the_samples(dev):
    """Per-pixel samples that require a gradient (what AnyNet.py:60-73 and DeepPruner.py:192 feed the builders): d disp_sample,
    d reference_fm and d target_fm of the module-level builders under torch.autograd -- the plain builders and fast_dif_fms's
    p-norm over the channels (p = 0.5, 1, 2, 3) -- against the REFERENCE's builders under its own autograd
    (tests/golden/fast_volumes_grad.npz) and against an FP64 evaluation of the oracle (error <= 4x the FP32 reference's own)."""
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.cat_fms import fast_cat_fms
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.dif_fms import fast_dif_fms
    g = golden("fast_volumes_grad.npz")
    forms = [("cat", "cat", {}), ("dif", "dif", {}), ("difn1", "dif", dict(normalize=True, p=1.0)), ("difn2", "dif", dict(normalize=True, p=2.0)),
             ("difn3", "dif", dict(normalize=True, p=3.0)), ("difnh", "dif", dict(normalize=True, p=0.5))]
    for i, row in enumerate(g["cases"]):
        sh, D, seed = tuple(int(v) for v in row[:4]), int(row[4]), int(row[5])
        a, b = _rand(sh, seed), _rand(sh, seed + 1000)
        gen = torch.Generator().manual_seed(seed + 2000)
        ds = torch.rand((sh[0], D, sh[2], sh[3]), generator=gen) * sh[3] * 0.6 - 2.0
        for name, kind, kw in forms:
            shape = (sh[0], D, sh[2], sh[3]) if kw else (sh[0], (2 if kind == "cat" else 1) * sh[1], D, sh[2], sh[3])
            up = _rand(shape, seed + 3000 + (0 if name == "cat" else 1))
            L, R, S = a.to(dev).requires_grad_(), b.to(dev).requires_grad_(), ds.to(dev).requires_grad_()
            vol = (fast_cat_fms if kind == "cat" else fast_dif_fms)(L, R, disp_sample=S, **kw)
            vol.backward(up.to(dev))
            t64 = O.fast_volume_grads(a, b, up, kind=kind, disp_sample=ds, dtype=torch.float64, wrt_samples=True, **kw)
            if kw:
                want = torch.as_tensor(g["%s_samples_out_%d" % (name, i)])     # (p = 0.5 sums square roots: values of 20 - 30)
                assert (vol.detach().cpu() - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item())
            for got, key, truth in ((L.grad, "dL", t64[0]), (R.grad, "dR", t64[1]), (S.grad, "dS", t64[2])):
                ref = torch.as_tensor(g["%s_samples_%s_%d" % (name, key, i)])
                scale = max(1.0, ref.abs().max().item())
                assert got.shape == ref.shape
                assert (got.cpu() - ref).abs().max().item() <= 2e-5 * scale, (name, key, i)
                e_got = (got.cpu().double() - truth).abs().max().item()
                e_ref = (ref.double() - truth).abs().max().item()
                assert e_got <= max(4 * e_ref, 1e-6 * scale), (name, key, i, e_got, e_ref)
    # only the samples want a gradient (the features detached, DeepPruner's refinement of the range predictors)
    S = ds.to(dev).requires_grad_()
    fast_cat_fms(a.to(dev), b.to(dev), disp_sample=S).backward(_rand((sh[0], 2 * sh[1], D, sh[2], sh[3]), seed + 3000).to(dev))
    assert (S.grad.cpu() - torch.as_tensor(g["cat_samples_dS_%d" % i])).abs().max().item() <= 2e-5 * max(1.0, float(np.abs(g["cat_samples_dS_%d" % i]).max()))


@pytest.mark.parametrize("C,G", [(16, 2), (32, 8), (12, 12), (320, 40)])
@pytest.mark.parametrize("W,md,sd", [(40, 9, 0), (300, 48, 0), (70, 12, -3)])
def test_gwc(dev, C, G, W, md, sd):
    """MFMA form (0 <= d <= 64, even channels/group) and VALU form (everything else: negative disparities, one channel per
    group) against the oracle."""
    ops = _ops()
    L, R = _rand((2, C, 5, W), 3), _rand((2, C, 5, W), 4)
    idx = ops.disp_index_list(md, sd, 1)
    got = ops.gwc_fms(L.to(dev), R.to(dev), idx, G).cpu()
    ref = O.gwc_fms(L, R, md, sd, 1, G)
    assert (got - ref).abs().max().item() <= 3e-6  # <= C/G FP32 products, different summation order


@pytest.mark.parametrize("C,W,D", [(8, 40, 9), (32, 300, 48), (4, 23, 9)])
def test_gwc_second_witness(dev, C, W, D):
    """The group-wise correlation kernels against two statements that share no code with the gwc oracle (which has no
    reference implementation to be pinned to): (i) G = C is the element-wise product of the two halves of cat_fms's volume --
    the HIP cat_fms is bit-exact against the reference's; (ii) G = 1 rescaled by C is correlation1d_cost's channels in
    disparity order (correlation1d_cost.py:12-25), leaky-ReLU applied to both."""
    ops = _ops()
    L, R = _rand((2, C, 5, W), 13), _rand((2, C, 5, W), 14)
    Ld, Rd = L.to(dev), R.to(dev)
    idx = ops.disp_index_list(D, 0, 1)
    cat = ops.cat_fms(Ld, Rd, idx)
    per_channel = ops.gwc_fms(Ld, Rd, idx, C)
    assert torch.equal(per_channel, cat[:, :C] * cat[:, C:])                 # one product per element: exact
    assert torch.equal(per_channel.cpu(), O.gwc_fms(L, R, D, 0, 1, C))
    dot = ops.gwc_fms(Ld, Rd, idx, 1)[:, 0] * C
    cor = ops.correlation1d(Ld, Rd, D, 0.1)
    assert (F.leaky_relu(dot, 0.1) - cor.flip(1)).abs().max().item() <= 1e-5 * max(1.0, math.sqrt(C))


# ------------------------------------------------------------------------------------------- conv family
def _affine(C, seed):
    g = torch.Generator().manual_seed(seed)
    return 0.5 + torch.rand(C, generator=g), torch.rand(C, generator=g) - 0.5


@pytest.mark.parametrize("Ci,Co,stride", [(32, 32, 1), (64, 32, 1), (64, 64, 1), (32, 64, 1), (32, 64, 2), (64, 64, 2)])
@pytest.mark.parametrize("shape", [(2, 6, 10, 70), (1, 5, 9, 13), (1, 6, 11, 96), (2, 3, 5, 48)])
def test_conv3d_k3(dev, Ci, Co, stride, shape):
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 5)
    w = _rand((Co, Ci, 3, 3, 3), 6, 1.0 / math.sqrt(Ci * 27))
    sc, sh = _affine(Co, 7)
    ref = F.conv3d(x, w, None, stride=stride, padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    res = _rand(ref.shape, 8)
    wp = ops.pack_conv3d_weights(w.to(dev))
    got = ops.conv3d_k3(x.to(dev), wp, Co, sc.to(dev), sh.to(dev), None, stride, False).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-5  # |values| ~ 1, K = Ci*27 FP32 fma chain
    got = ops.conv3d_k3(x.to(dev), wp, Co, sc.to(dev), sh.to(dev), res.to(dev), stride, True).cpu()
    assert (got - F.relu(ref + res)).abs().max().item() <= 2e-5
    got = ops.conv3d_k3(x.to(dev), wp, Co, None, None, None, stride, False).cpu()
    assert (got - F.conv3d(x, w, None, stride=stride, padding=1)).abs().max().item() <= 2e-5


@pytest.mark.parametrize("Ci,Co,stride,shape", [
    (5, 32, 1, (1, 4, 6, 50)),      # odd channel count: zero-padded weight fragments + bounds-checked copies
    (40, 64, 1, (1, 3, 5, 48)),     # row-pair width with 64 output channels -> flattened path
    (24, 32, 1, (2, 5, 7, 96)),     # W % 48 == 0 -> row-pair tiles
    (12, 32, 2, (1, 6, 9, 21)),
    (7, 64, 2, (1, 5, 8, 33)),
    (32, 32, 1, (1, 3, 6, 156)),    # StereoNet width: TX = 52 tiles
    (64, 64, 1, (1, 5, 9, 80)),     # W % 40 == 0 with 64 output channels -> row-quad tiles (8 columns x 4 rows)
    (20, 64, 1, (2, 3, 6, 120)),
    (64, 64, 1, (1, 2, 3, 40)),
    (64, 64, 1, (1, 4, 7, 72)),     # W % 24 == 0 only -> 24-column row-quad tiles
    (64, 64, 1, (4, 24, 20, 120)),  # both widths possible: picked by the rounds x columns model
    (64, 64, 1, (1, 4, 9, 60)),     # few tiles, W % 30 == 0 -> 30-column flattened tiles
    (32, 32, 1, (2, 5, 7, 128)),    # training-crop widths: 32-column row-pair tiles ...
    (64, 32, 1, (1, 4, 6, 64)),
    (64, 64, 1, (1, 5, 9, 64)),     # ... and 32-column row-quad tiles
    (32, 64, 1, (2, 4, 8, 32)),
    (32, 64, 2, (1, 6, 8, 128)),    # stride 2 at crop widths: 22-column tiles
    (64, 64, 2, (2, 4, 9, 64)),
    (64, 64, 2, (1, 4, 8, 240)),    # ... and the 30-column ones
])
def test_conv3d_any_channels(dev, Ci, Co, stride, shape):
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 31)
    w = _rand((Co, Ci, 3, 3, 3), 32, 1.0 / math.sqrt(Ci * 27))
    sc, sh = _affine(Co, 33)
    ref = F.conv3d(x, w, None, stride=stride, padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    wp = ops.pack_conv3d_weights(w.to(dev))
    got = ops.conv3d_k3(x.to(dev), wp, Co, sc.to(dev), sh.to(dev), None, stride, True).cpu()
    assert (got - F.relu(ref)).abs().max().item() <= 2e-5
    res = _rand(ref.shape, 34)
    got = ops.conv3d_k3(x.to(dev), wp, Co, sc.to(dev), sh.to(dev), res.to(dev), stride, False).cpu()
    assert (got - (ref + res)).abs().max().item() <= 2e-5


@pytest.mark.parametrize("Ci,Co", [(64, 64), (64, 32), (20, 32), (9, 64)])
@pytest.mark.parametrize("shape", [(2, 3, 5, 35), (1, 4, 6, 61), (1, 5, 7, 64), (2, 3, 4, 120)])
def test_deconv3d(dev, Ci, Co, shape):
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 9)
    w = _rand((Ci, Co, 3, 3, 3), 10, 1.0 / math.sqrt(Ci * 27 / 8))
    sc, sh = _affine(Co, 11)
    ref = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1)
    ref = ref * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    res = _rand(ref.shape, 12)
    wp = ops.pack_deconv3d_weights(w.to(dev))
    got = ops.deconv3d_k3s2(x.to(dev), wp, Co, sc.to(dev), sh.to(dev), None, False).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-5
    got = ops.deconv3d_k3s2(x.to(dev), wp, Co, sc.to(dev), sh.to(dev), res.to(dev), True).cpu()
    assert (got - F.relu(ref + res)).abs().max().item() <= 2e-5


@pytest.mark.parametrize("Ci", [32, 5])
def test_conv3d_c1(dev, Ci):
    ops = _ops()
    x = _rand((2, Ci, 6, 9, 70), 13)
    w = _rand((1, Ci, 3, 3, 3), 14, 1.0 / math.sqrt(Ci * 27))
    res = _rand((2, 1, 6, 9, 70), 15)
    ref = F.conv3d(x, w, torch.tensor([0.25]), padding=1) + res
    got = ops.conv3d_k3_c1(x.to(dev), w.to(dev), 0.25, res.to(dev)).cpu()
    assert (got - ref).abs().max().item() <= 1e-5


@pytest.mark.parametrize("shape", [(2, 6, 9, 72), (1, 17, 10, 240), (1, 3, 19, 64), (1, 9, 8, 124)])
@pytest.mark.usefixtures("single_chain")
def test_conv3d_c1_vector_rows(dev, shape):
    """Rows that are 16-byte aligned (W % 4 == 0) take the register-staged kernel (16-byte fetches, one LDS word per input
    row and thread, halo columns by lane shifts): same accumulation order as the dword kernel -> BIT-identical to it, and
    within 1e-5 of the CPU convolution; tile edges in every direction (8 z x 8 y x 60 x tiles)."""
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, 32, D, H, W), 16)
    w = _rand((1, 32, 3, 3, 3), 17, 1.0 / math.sqrt(32 * 27))
    res = _rand((B, 1, D, H, W), 18)
    ref = F.conv3d(x, w, torch.tensor([-0.5]), padding=1) + res
    got = ops.conv3d_k3_c1(x.to(dev), w.to(dev), -0.5, res.to(dev))
    assert (got.cpu() - ref).abs().max().item() <= 1e-5
    old = ops.conv3d_k3_c1(_misaligned(x, dev), w.to(dev), -0.5, res.to(dev))     # a misaligned input takes the dword kernel
    assert torch.equal(got, old)
    assert (ops.conv3d_k3_c1(x.to(dev), w.to(dev), 0.0, None).cpu() - (ref - res + 0.5)).abs().max().item() <= 1e-5


# ------------------------------------------------------------------------------------------- upsampling
@pytest.mark.parametrize("ins,outs", [((4, 6, 10), (16, 24, 40)), ((3, 5, 7), (11, 17, 30))])
def test_trilinear(dev, ins, outs):
    ops = _ops()
    x = _rand((2,) + ins, 16)
    ref = F.interpolate(x.unsqueeze(1), list(outs), mode="trilinear", align_corners=True).squeeze(1)
    got = ops.trilinear_ac(x.to(dev), outs).cpu()
    assert (got - ref).abs().max().item() <= 2e-6  # same index/weight arithmetic, fma contraction may differ


def test_trilinear_weights_round_like_the_reference(dev):
    """The interpolation weight must be lambda = fl(fl(scale * dst) - i0) -- the product rounded BEFORE the subtraction, as in
    the ATen CPU path the reference reaches -- not an fma of the two (csrc/interp.h).  A 0/1 comb makes the kernel output the
    weights themselves, which are compared bit for bit with the same arithmetic done by torch in FP32."""
    ops = _ops()
    Wi, Wo = 241, 964
    x = (torch.arange(Wi) % 2).float().view(1, 1, 1, Wi)
    got = ops.trilinear_ac(x.to(dev), (1, 1, Wo)).cpu().view(-1)
    scale = torch.tensor(float(Wi - 1), dtype=torch.float32) / torch.tensor(float(Wo - 1), dtype=torch.float32)
    src = scale * torch.arange(Wo, dtype=torch.float32)          # rounded FP32 products
    i0 = src.floor()
    l1 = (src - i0).clamp(0.0, 1.0)
    i0 = i0.long()
    i1 = torch.where(i0 < Wi - 1, i0 + 1, i0)
    want = torch.where(i1 % 2 == 1, l1, torch.zeros_like(l1)) + torch.where(i0 % 2 == 1, 1.0 - l1, torch.zeros_like(l1))
    assert torch.equal(got, want)
    fused, _ = ops.trilinear_ac_soft_argmin(x.to(dev), (1, 1, Wo), [0.0], 1.0)
    assert torch.equal(fused.cpu().view(-1), want)


def test_deconv_k8s4(dev):
    ops = _ops()
    x = _rand((2, 3, 5, 9), 17)
    w = _rand((1, 1, 8, 8, 8), 18, 0.1)
    ref = F.conv_transpose3d(x.unsqueeze(1), w, None, stride=4, padding=2).squeeze(1)
    got = ops.deconv3d_k8s4_c1(x.to(dev), w.view(8, 8, 8).to(dev)).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 1e-5


# ------------------------------------------------------------------------------------------- regression
@pytest.mark.parametrize("D,HW", [(192, (6, 40)), (24, (5, 13)), (5, (2, 2))])
@pytest.mark.parametrize("gain", [1.0, 12.0])
def test_soft_argmin(dev, D, HW, gain):
    ops = _ops()
    cost = _rand((2, D) + HW, 19, gain)
    vals = ops.disp_sample_values(D, 0, 1)
    got = ops.soft_argmin(cost.to(dev), vals, 1.0, True).cpu()
    truth = O.soft_argmin_f64(cost, D).float()
    ref = O.faster_soft_argmin(cost, D)
    # our FP64-accumulated result is within FP32 rounding of the truth ...
    assert (got - truth).abs().max().item() <= 2e-5 * max(1.0, D / 64)
    # ... and the reference's own FP32 evaluation is no closer to the truth than ~1e-4 at D=192
    assert (got - ref).abs().max().item() <= 1.5e-4
    got = ops.soft_argmin(cost.to(dev), vals, 0.5, False).cpu()
    assert (got - O.soft_argmin(cost, D, alpha=0.5, normalize=False)).abs().max().item() <= 1e-3 * gain


def test_soft_argmin_known_answer(dev):
    ops = _ops()
    cost = torch.ones(1, 5, 2, 2)
    vals = ops.disp_sample_values(9, -4, 2)
    assert vals == [-4.0, -2.0, 0.0, 2.0, 4.0]
    got = ops.soft_argmin(cost.to(dev), vals, 1.0, True).cpu()
    assert got.abs().max().item() <= 1e-7  # reference prints -5.96e-8 (test_disp_predictors.py:42-102)
    got = ops.local_soft_argmin(cost.to(dev), 2, 1, -4, 2, 1.0).cpu()
    assert torch.equal(got, torch.full((1, 1, 2, 2), -2.0))


def test_soft_argmin_sampled(dev):
    ops = _ops()
    cost = _rand((2, 12, 5, 8), 20, 3.0)
    samp = _rand((2, 12, 5, 8), 21, 10.0)
    got = ops.soft_argmin_sampled(cost.to(dev), samp.to(dev), 1.0, True).cpu()
    ref = O.soft_argmin(cost, 12, disp_sample=samp)
    assert (got - ref).abs().max().item() <= 1e-5


@pytest.mark.parametrize("radius,rd,start,dil", [(2, 1, 0, 1), (3, 2, -4, 2), (0, 1, 0, 1)])
def test_local_soft_argmin(dev, radius, rd, start, dil):
    ops = _ops()
    D = 48
    cost = _rand((2, D, 7, 20), 22, 4.0)
    cost[0, :, 0, 0] = 1.0          # full tie -> index 0
    cost[0, 5, 1, 1] = cost[0, 9, 1, 1] = 100.0  # two-way tie -> lowest index
    cost[1, D - 1, 2, 2] = 50.0     # window clipped at the top
    got, idx = ops.local_soft_argmin(cost.to(dev), radius, rd, start, dil, 1.0, return_index=True)
    ref, ridx = O.local_soft_argmin(cost, D * dil, radius, start, dil, rd, 1.0)
    assert torch.equal(idx.cpu(), ridx)  # index path: bit-exact
    assert (got.cpu() - ref).abs().max().item() <= 3e-5


def test_trilinear_soft_argmin_fused(dev):
    ops = _ops()
    x = _rand((2, 6, 8, 12), 23, 3.0)
    outs = (24, 32, 48)
    vals = ops.disp_sample_values(24, 0, 1)
    up = ops.trilinear_ac(x.to(dev), outs)
    a = ops.soft_argmin(up, vals, 1.0, True)
    b = ops.trilinear_soft_argmin(x.to(dev), outs, vals, 1.0)
    assert (a - b).abs().max().item() <= 2e-5  # identical logits; only the max-rescale grouping differs (few ulp at disp ~ 20)


# ------------------------------------------------------------------------------------------- conf head / EPE
@pytest.mark.parametrize("D,Cm", [(48, 16), (192, 64), (20, 6)])
def test_conf_head(dev, D, Cm):
    ops = _ops()
    cost = _rand((2, D, 11, 70), 24)
    w1 = _rand((Cm, D, 3, 3), 25, 1.0 / math.sqrt(D * 9))
    w2 = _rand((Cm,), 26, 0.5)
    sc, sh = _affine(Cm, 27)
    h = F.relu(F.conv2d(cost, w1, None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    ref = torch.sigmoid(F.conv2d(h, w2.view(1, Cm, 1, 1)))
    wp = ops.pack_conf_head_weights(w1.to(dev))
    got = ops.conf_head(cost.to(dev), wp, sc.to(dev), sh.to(dev), w2.to(dev)).cpu()
    assert (got - ref).abs().max().item() <= 1e-5


def test_epe_accumulate(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(28)
    gt = torch.rand((3, 1, 20, 32), generator=g) * 220 - 10   # some <= 0 and some >= 192 -> masked out
    est = gt + torch.randn((3, 1, 20, 32), generator=g) * 3
    gt[2] = -1.0  # an image with an empty mask contributes zeros but still counts (pixel_error.py:48-55)
    acc = torch.zeros(6, dtype=torch.float64, device=dev)
    ops.epe_accumulate(est.to(dev), gt.to(dev), acc, (17, 30), 0, 192)
    ref, n = O.dataset_metrics([est], [gt], (17, 30), 0, 192)
    acc = acc.cpu()
    assert acc[0].item() == 3 and n == 3
    for i, k in enumerate(("epe", "1px", "2px", "3px", "5px")):
        assert abs(acc[1 + i].item() / 3 - ref[k]) <= 1e-5 * max(1.0, abs(ref[k]))


def test_epe_accumulate_multi_matches_single(dev):
    """dmb_epe_accum_multi_f64 (the disparity maps of one forward against the same ground truth in one pass) against one
    dmb_epe_accum_f64 call per map, over two updates, and through EpeAccumulator.update, which picks the one-pass form for
    2 .. 4 maps: bit-identical accumulators (no atomics: 64 slice sums per image added in ascending order, images by a fixed
    butterfly)."""
    ops = _ops()
    from densematchingbenchmark_amd.evaluation import EpeAccumulator
    g = torch.Generator().manual_seed(29)
    acc_m = torch.zeros((3, 6), dtype=torch.float64, device=dev)
    acc_s = torch.zeros((3, 6), dtype=torch.float64, device=dev)
    holder = EpeAccumulator(dev, 3, 0, 192)
    for rep in range(2):
        gt = torch.rand((2, 1, 24, 40), generator=g) * 220 - 10
        ests = [gt + torch.randn((2, 1, 24, 40), generator=g) * (1 + i) for i in range(3)]
        if rep == 1:
            gt[1] = 500.0   # empty mask
        gd, ed = gt.to(dev), [e.to(dev) for e in ests]
        ops.epe_accumulate_multi(ed, gd, acc_m, (21, 37), 0, 192)
        for i in range(3):
            ops.epe_accumulate(ed[i], gd, acc_s[i], (21, 37), 0, 192)
        holder.update(ed, gd, (21, 37))
    # slice sums are written, not added atomically, and combined in a fixed order: the three forms agree bit for bit
    assert acc_m[0, 0].item() == 4 and torch.equal(acc_m, acc_s) and torch.equal(holder.acc, acc_s)


# ------------------------------------------------------------------------------------------- 2-D backbone ops
@pytest.mark.parametrize("Ci,Co,k,stride,dil,shape", [
    (3, 32, 3, 2, 1, (2, 20, 70)),       # firstconv.0: 3 input channels (zero-padded fragments), stride 2
    (32, 32, 3, 1, 1, (1, 17, 96)),
    (32, 64, 3, 2, 1, (1, 18, 50)),      # layer2.0.conv1
    (32, 64, 1, 2, 1, (1, 18, 50)),      # layer2.0.downsample
    (64, 128, 3, 1, 1, (2, 9, 48)),
    (128, 128, 3, 1, 2, (1, 12, 61)),    # layer4: dilation 2
    (128, 32, 1, 1, 1, (1, 1, 2)),       # SPP branch after 64x64 pooling
    (320, 128, 3, 1, 1, (1, 8, 30)),     # lastconv.0
    (20, 32, 3, 1, 1, (1, 6, 13)),
    (3, 32, 5, 2, 1, (2, 21, 75)),       # StereoNet down-sampling head 0: 5x5, stride 2
    (32, 32, 5, 2, 1, (1, 48, 156)),
    (32, 32, 3, 2, 1, (1, 9, 101)),
    (32, 64, 1, 2, 1, (1, 7, 97)),
    (16, 32, 1, 2, 1, (1, 1, 1)),
])
def test_conv2d(dev, Ci, Co, k, stride, dil, shape):
    ops = _ops()
    B, H, W = shape
    x = _rand((B, Ci, H, W), 51)
    w = _rand((Co, Ci, k, k), 52, 1.0 / math.sqrt(Ci * k * k))
    sc, sh = _affine(Co, 53)
    pad = dil * (k // 2)
    ref = F.conv2d(x, w, None, stride=stride, padding=pad, dilation=dil) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    res = _rand(ref.shape, 54)
    wp = ops.pack_conv2d_weights(w.to(dev))
    got = ops.conv2d(x.to(dev), wp, Co, k, stride, dil, sc.to(dev), sh.to(dev), None, False).cpu()
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-5
    got = ops.conv2d(x.to(dev), wp, Co, k, stride, dil, sc.to(dev), sh.to(dev), res.to(dev), True).cpu()
    assert (got - F.relu(ref + res)).abs().max().item() <= 2e-5
    got = ops.conv2d(x.to(dev), wp, Co, k, stride, dil).cpu()
    assert (got - F.conv2d(x, w, None, stride=stride, padding=pad, dilation=dil)).abs().max().item() <= 2e-5


@pytest.mark.parametrize("Ci,Co,k,stride,dil,shape", [
    (3, 32, 3, 2, 1, (2, 20, 72)),
    (32, 32, 3, 1, 1, (1, 17, 96)),
    (32, 64, 3, 2, 1, (1, 18, 56)),
    (32, 64, 1, 2, 1, (1, 18, 56)),
    (64, 64, 3, 1, 1, (2, 19, 100)),
    (64, 128, 3, 1, 1, (2, 9, 48)),
    (128, 128, 3, 1, 2, (1, 12, 64)),
    (128, 32, 1, 1, 1, (1, 3, 4)),
    (64, 128, 1, 1, 1, (1, 9, 52)),
    (320, 128, 3, 1, 1, (1, 8, 32)),
    (20, 32, 3, 1, 1, (1, 6, 12)),
    (3, 32, 5, 2, 1, (2, 21, 80)),
    (32, 32, 5, 2, 1, (1, 48, 152)),
    (32, 32, 3, 1, 4, (2, 21, 72)),
    (32, 32, 3, 1, 8, (1, 40, 96)),
    (32, 1, 3, 1, 1, (2, 19, 52)),
])
@pytest.mark.usefixtures("single_chain")
def test_conv2d_vector_path(dev, Ci, Co, k, stride, dil, shape):
    """Widths that are multiples of 4 take the 16-byte staging + transposed-store path; it must agree bit for bit with the
    scalar path (same MFMA sequence, same epilogue arithmetic) and with torch within the FP32 tolerance."""
    ops = _ops()
    B, H, W = shape
    x = _rand((B, Ci, H, W), 151)
    w = _rand((Co, Ci, k, k), 152, 1.0 / math.sqrt(Ci * k * k))
    sc, sh = _affine(Co, 153)
    pad = dil * (k // 2)
    ref = F.conv2d(x, w, None, stride=stride, padding=pad, dilation=dil) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    res = _rand(ref.shape, 154)
    wp = ops.pack_conv2d_weights(w.to(dev))
    args = (x.to(dev), wp, Co, k, stride, dil, sc.to(dev), sh.to(dev), res.to(dev), True)
    got = ops.conv2d(*args)
    scalar = ops.conv2d(_misaligned(x, dev), *args[1:])      # a misaligned input takes the scalar path
    assert torch.equal(got, scalar)
    assert (got.cpu() - F.relu(ref + res)).abs().max().item() <= 2e-5
    got = ops.conv2d(x.to(dev), wp, Co, k, stride, dil).cpu()
    assert (got - F.conv2d(x, w, None, stride=stride, padding=pad, dilation=dil)).abs().max().item() <= 2e-5


@pytest.mark.parametrize("shape", [(8, 136, 240), (2, 20, 48), (3, 9, 100)])
def test_conv2d_tile_height_is_picked_per_launch(dev, shape):
    """64-channel 3x3 layers pick their tile height per launch (4 or 2 output rows per wave, by rounds x rows on the persistent
    grid): launch sizes on either side of the choice against the CPU convolution -- with folded BatchNorm, skip operand and
    ReLU -- and a batch against its own items one at a time (a different launch size, possibly a different height: the same
    MFMA sequence per output, so not a bit may change)."""
    ops = _ops()
    B, H, W = shape
    x = _rand((B, 64, H, W), 301).to(dev)
    w = _rand((64, 64, 3, 3), 302, 1.0 / math.sqrt(64 * 9))
    sc, sh = _affine(64, 303)
    res = _rand((B, 64, H, W), 304).to(dev)
    wp = ops.pack_conv2d_weights(w.to(dev))
    whole = ops.conv2d(x, wp, 64, 3, 1, 1, sc.to(dev), sh.to(dev), res, True)
    for b in range(B):
        one = ops.conv2d(x[b:b + 1].contiguous(), wp, 64, 3, 1, 1, sc.to(dev), sh.to(dev), res[b:b + 1].contiguous(), True)
        assert torch.equal(one[0], whole[b])
    ref = F.relu(F.conv2d(x.cpu(), w, None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res.cpu())
    assert (whole.cpu() - ref).abs().max().item() <= 2e-5


def test_conv2d_unsupported_combinations_fail_loudly(dev):
    ops = _ops()
    from densematchingbenchmark_amd._lib import DmbLibraryError
    x = _rand((1, 64, 8, 8), 1).to(dev)
    for Co, k, stride, dil in ((64, 3, 2, 2), (128, 3, 2, 1), (64, 3, 1, 4), (64, 5, 2, 1), (32, 5, 1, 1)):
        wp = ops.pack_conv2d_weights(_rand((Co, 64, k, k), 2).to(dev))
        with pytest.raises(DmbLibraryError):
            ops.conv2d(x, wp, Co, k, stride, dil)


def test_conv2d_channel_windows(dev):
    """Input, output and residual may be channel windows of wider tensors (the 320-channel SPP buffer)."""
    ops = _ops()
    xw = _rand((2, 100, 9, 50), 61)
    w = _rand((64, 32, 3, 3), 62, 1.0 / math.sqrt(32 * 9))
    sc, sh = _affine(64, 63)
    resw = _rand((2, 80, 9, 50), 64)
    out = torch.full((2, 150, 9, 50), 7.0)
    ref = F.relu(F.conv2d(xw[:, 40:72], w, None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + resw[:, 10:74])
    wp = ops.pack_conv2d_weights(w.to(dev))
    o = out.to(dev)
    ops.conv2d(xw.to(dev), wp, 64, 3, 1, 1, sc.to(dev), sh.to(dev), resw.to(dev), True, in_window=(40, 32), out=o,
               out_ch_offset=50, res_ch_offset=10)
    o = o.cpu()
    assert (o[:, 50:114] - ref).abs().max().item() <= 2e-5
    assert torch.equal(o[:, :50], out[:, :50]) and torch.equal(o[:, 114:], out[:, 114:])   # neighbours untouched


@pytest.mark.parametrize("k,shape", [(8, (2, 5, 24, 40)), (64, (1, 3, 64, 128)), (16, (1, 4, 40, 50))])
def test_avgpool2d(dev, k, shape):
    ops = _ops()
    x = _rand(shape, 71)
    got = ops.avgpool2d(x.to(dev), k).cpu()
    ref = F.avg_pool2d(x, k, k)
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= 1e-6
    wide = _rand((shape[0], shape[1] + 5, shape[2], shape[3]), 72)
    got = ops.avgpool2d(wide.to(dev), k, in_window=(3, shape[1])).cpu()
    assert (got - F.avg_pool2d(wide[:, 3:3 + shape[1]], k, k)).abs().max().item() <= 1e-6


@pytest.mark.parametrize("ins,outs", [((1, 2), (64, 128)), ((4, 8), (64, 128)), ((3, 5), (17, 31)), ((2, 2), (2, 2))])
def test_bilinear_ac(dev, ins, outs):
    ops = _ops()
    x = _rand((2, 6) + ins, 81)
    ref = F.interpolate(x, outs, mode="bilinear", align_corners=True)
    assert (ops.bilinear_ac(x.to(dev), outs).cpu() - ref).abs().max().item() <= 2e-6
    out = torch.zeros((2, 10) + outs).to(dev)
    ops.bilinear_ac(x.to(dev), outs, out=out, out_ch_offset=3)
    assert (out.cpu()[:, 3:9] - ref).abs().max().item() <= 2e-6 and out.cpu()[:, :3].abs().max().item() == 0


@pytest.mark.parametrize("dil,shape", [(4, (2, 21, 70)), (8, (1, 40, 96)), (8, (1, 5, 7))])
def test_conv2d_dilation_4_8(dev, dil, shape):
    ops = _ops()
    B, H, W = shape
    x = _rand((B, 32, H, W), 91)
    w = _rand((32, 32, 3, 3), 92, 1.0 / math.sqrt(32 * 9))
    sc, sh = _affine(32, 93)
    ref = F.relu(F.conv2d(x, w, None, padding=dil, dilation=dil) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    wp = ops.pack_conv2d_weights(w.to(dev))
    got = ops.conv2d(x.to(dev), wp, 32, 3, 1, dil, sc.to(dev), sh.to(dev), None, True).cpu()
    assert (got - ref).abs().max().item() <= 2e-5


def test_conv2d_single_output_channel_with_skip_window(dev):
    """conv_res of the refinement: 32 -> 1, bias, skip read from channel 0 of the 4-channel mixed input, ReLU."""
    ops = _ops()
    x = _rand((2, 32, 19, 50), 95)
    w = _rand((1, 32, 3, 3), 96, 1.0 / math.sqrt(32 * 9))
    bias = _rand((1,), 97)
    mixed = _rand((2, 4, 19, 50), 98)
    ref = F.relu(F.conv2d(x, w, bias, padding=1) + mixed[:, :1])
    got = ops.conv2d(x.to(dev), ops.pack_conv2d_weights(w.to(dev)), 1, 3, 1, 1, None, bias.to(dev), mixed.to(dev), True).cpu()
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("ins,outs,mult", [((24, 40), (192, 320), 8.0), ((5, 7), (13, 30), 30.0 / 7), ((9, 9), (9, 9), 1.0),
                                            ((48, 156), (384, 1248), 8.0)])
def test_bilinear_half_pixel_scaled(dev, ins, outs, mult):
    ops = _ops()
    x = _rand((2, 1) + ins, 99, 10.0)
    ref = F.interpolate(x, size=outs, mode="bilinear", align_corners=False) * mult
    got = ops.bilinear_scale(x.to(dev), outs, mult).cpu()
    assert (got - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item() / 8)
    if ins == outs:
        assert torch.equal(got, x)


@pytest.mark.parametrize("ins,outs", [((6, 9, 20), (24, 36, 80)), ((5, 7, 9), (19, 25, 33)), ((3, 4, 5), (9, 13, 17))])
def test_trilinear_with_fused_regression_is_bit_identical(dev, ins, outs):
    """The producer-fused up-sampling writes the same volume as trilinear_ac and the same disparity as a separate
    soft-argmin pass over that volume (block order of the online soft-max included), bit for bit."""
    ops = _ops()
    x = _rand((2,) + ins, 171, 3.0).to(dev)
    vals = ops.disp_sample_values(outs[0], 0, 1)
    cost, disp = ops.trilinear_ac_soft_argmin(x, outs, vals, 1.0)
    ref_cost = ops.trilinear_ac(x, outs)
    assert torch.equal(cost, ref_cost)
    assert torch.equal(disp, ops.soft_argmin(ref_cost, vals, 1.0, True))
    # the hint is honoured only for the exact tensor state and parameters it was computed for
    ops.RegressionHint.attach(cost, vals, 1.0, disp)
    assert ops.RegressionHint.lookup(cost, vals, 1.0, True) is disp
    assert ops.RegressionHint.lookup(cost, vals, 2.0, True) is None
    assert ops.RegressionHint.lookup(cost, vals[:-1] + [0.0], 1.0, True) is None
    assert ops.RegressionHint.lookup(cost, vals, 1.0, False) is None
    cost.add_(1.0)
    assert ops.RegressionHint.lookup(cost, vals, 1.0, True) is None


# ------------------------------------------------------------------------------------------- GC-Net primitives
@pytest.mark.parametrize("Ci,stride,shape", [(128, 1, (1, 3, 5, 21)), (64, 2, (1, 6, 9, 20)), (192, 1, (1, 2, 4, 64))])
def test_conv3d_128_output_channels(dev, Ci, stride, shape):
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 181)
    w = _rand((128, Ci, 3, 3, 3), 182, 1.0 / math.sqrt(Ci * 27))
    sc, sh = _affine(128, 183)
    ref = F.relu(F.conv3d(x, w, None, stride=stride, padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1))
    got = ops.conv3d_k3(x.to(dev), ops.pack_conv3d_weights(w.to(dev)), 128, sc.to(dev), sh.to(dev), None, stride, True).cpu()
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("Ci,Co,shape", [(32, 1, (2, 3, 5, 24)), (32, 1, (1, 4, 6, 35)), (16, 7, (1, 2, 3, 8))])
def test_deconv3d_few_output_channels(dev, Ci, Co, shape):
    """GC-Net's output layer: ConvTranspose3d(32, 1, 3, 2, 1, 1) with bias on zero-padded weight rows."""
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 185)
    w = _rand((Ci, Co, 3, 3, 3), 186, 1.0 / math.sqrt(Ci * 27 / 8))
    bias = _rand((Co,), 187)
    ref = F.conv_transpose3d(x, w, bias, stride=2, padding=1, output_padding=1)
    got = ops.deconv3d_k3s2(x.to(dev), ops.pack_deconv3d_weights(w.to(dev)), Co, None, bias.to(dev), None, False).cpu()
    assert got.shape == ref.shape and (got - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("kind,shape", [("conv", (1, 4, 6, 48)), ("conv", (1, 3, 5, 21)), ("deconv", (1, 3, 4, 24)),
                                        ("deconv", (2, 2, 3, 13)), ("conv64", (1, 4, 8, 40))])
def test_conv3d_relu_before_skip(dev, kind, shape):
    """relu='pre': y = relu(bn(conv(x))) + skip -- GC-Net adds its skip connections to the activated output."""
    ops = _ops()
    B, D, H, W = shape
    Co = 64 if kind == "conv64" else 32
    x = _rand((B, 32, D, H, W), 191)
    sc, sh = _affine(Co, 193)
    if kind == "deconv":
        w = _rand((32, Co, 3, 3, 3), 192, 0.1)
        y = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1)
    else:
        w = _rand((Co, 32, 3, 3, 3), 192, 1.0 / math.sqrt(32 * 27))
        y = F.conv3d(x, w, None, padding=1)
    y = F.relu(y * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1))
    skip = _rand(y.shape, 194)
    if kind == "deconv":
        got = ops.deconv3d_k3s2(x.to(dev), ops.pack_deconv3d_weights(w.to(dev)), Co, sc.to(dev), sh.to(dev), skip.to(dev), "pre")
    else:
        got = ops.conv3d_k3(x.to(dev), ops.pack_conv3d_weights(w.to(dev)), Co, sc.to(dev), sh.to(dev), skip.to(dev), 1, "pre")
    assert (got.cpu() - (y + skip)).abs().max().item() <= 2e-5


# ------------------------------------------------------- experimental, opt-in: FP32 convolution on 3-way bf16 splits
@pytest.mark.parametrize("Ci,Co,shape", [(32, 32, (1, 6, 10, 96)), (64, 32, (2, 5, 7, 48)), (5, 32, (1, 4, 6, 48)),
                                         (32, 32, (1, 9, 13, 240)), (64, 64, (1, 6, 10, 72)), (32, 64, (2, 5, 7, 24)),
                                         (64, 64, (1, 9, 13, 120))])
@pytest.mark.usefixtures("single_chain")
def test_conv3d_bf16x6_is_as_accurate_as_fp32(dev, Ci, Co, shape):
    """The split kernel against an FP64 convolution: its error must not exceed the exact FP32 kernel's by more than
    noise, and both stay inside the FP32 tolerance of the other conv tests."""
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 201)
    w = _rand((Co, Ci, 3, 3, 3), 202, 1.0 / math.sqrt(Ci * 27))
    sc, sh = _affine(Co, 203)
    res = _rand((B, Co, D, H, W), 204)
    ref = F.relu(F.conv3d(x.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1, 1)
                 + sh.double().view(1, -1, 1, 1, 1) + res.double())
    xd, scd, shd, resd = x.to(dev), sc.to(dev), sh.to(dev), res.to(dev)
    exact = ops.conv3d_k3(xd, ops.pack_conv3d_weights(w.to(dev)), Co, scd, shd, resd, 1, True).cpu().double()
    split = ops.conv3d_k3_x6(xd, ops.pack_conv3d_x6_weights(w.to(dev)), Co, scd, shd, resd, True).cpu().double()
    e_exact, e_split = (exact - ref).abs(), (split - ref).abs()
    assert e_split.max().item() <= 2e-5 and e_exact.max().item() <= 2e-5
    assert e_split.max().item() <= 1.5 * e_exact.max().item() + 1e-7
    assert e_split.mean().item() <= 1.25 * e_exact.mean().item() + 1e-9
    plain = ops.conv3d_k3_x6(xd, ops.pack_conv3d_x6_weights(w.to(dev)), Co).cpu().double()
    assert (plain - F.conv3d(x.double(), w.double(), None, padding=1)).abs().max().item() <= 2e-5


@pytest.mark.parametrize("kind,shape", [("s2", (1, 6, 9, 48)), ("s2", (2, 5, 8, 120)), ("deconv", (1, 3, 5, 60)), ("deconv", (2, 2, 3, 24)),
                                        ("s1_32", (1, 5, 9, 312)), ("s1_32", (2, 4, 12, 48)), ("s1_64", (1, 4, 8, 156)), ("s1_64", (1, 3, 7, 40))])
@pytest.mark.usefixtures("single_chain")
def test_conv3d_vector_and_scalar_staging_agree(dev, kind, shape):
    """Rows that are 16-byte aligned are staged with 16-byte LDS-DMA words; the result must be bit-identical to the dword
    staging path (same MFMA sequence, same epilogue), which stays in use for other widths and for misaligned operands -- a
    misaligned input is how the dword path is reached here."""
    ops = _ops()
    B, D, H, W = shape
    xc = _rand((B, 32, D, H, W), 211)
    Co = 32 if kind == "s1_32" else 64
    sc, sh = _affine(Co, 213)
    outs = []
    for x in (xc.to(dev), _misaligned(xc, dev)):
        if kind == "deconv":
            w = _rand((32, 64, 3, 3, 3), 212, 0.1).to(dev)
            outs.append(ops.deconv3d_k3s2(x, ops.pack_deconv3d_weights(w), 64, sc.to(dev), sh.to(dev), None, True))
        else:
            w = _rand((Co, 32, 3, 3, 3), 212, 0.03).to(dev)
            outs.append(ops.conv3d_k3(x, ops.pack_conv3d_weights(w), Co, sc.to(dev), sh.to(dev), None, 2 if kind == "s2" else 1, True))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("Ci,shape", [(32, (1, 8, 16, 120)), (64, (2, 5, 9, 60)), (32, (1, 6, 8, 240)), (32, (1, 4, 4, 124))])
@pytest.mark.parametrize("mode", ["plain", "skip_relu", "relu_then_skip"])
@pytest.mark.usefixtures("single_chain")
def test_conv3d_s2_pair_epilogue_is_bit_identical(dev, Ci, shape, mode):
    """The stride-2 kernel's 8-byte epilogue (S2Cfg VEP: 32 x 32 accumulator tiles through a per-wave LDS scratch, a lane stores two
    adjacent columns of one channel) against the dword epilogue (taken when the output is not 8-byte aligned: the caller's ``out``
    here): the same FP32 operations per output, so bit-identical -- with folded BatchNorm, with the skip operand before / after the
    ReLU, over partial tiles in x, y and z -- and within 2e-5 of the CPU convolution."""
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 221)
    w = _rand((64, Ci, 3, 3, 3), 222, 1.0 / math.sqrt(Ci * 27))
    sc, sh = _affine(64, 223)
    Do, Ho, Wo = (D - 1) // 2 + 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1
    res = None if mode == "plain" else _rand((B, 64, Do, Ho, Wo), 224)
    relu = {"plain": False, "skip_relu": True, "relu_then_skip": "pre"}[mode]
    ref = F.conv3d(x, w, None, stride=2, padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    if mode == "skip_relu":
        ref = F.relu(ref + res)
    elif mode == "relu_then_skip":
        ref = F.relu(ref) + res
    wp = ops.pack_conv3d_weights(w.to(dev))
    outs = []
    for out in (None, _misaligned(torch.zeros(B, 64, Do, Ho, Wo), dev)):
        outs.append(ops.conv3d_k3(x.to(dev), wp, 64, sc.to(dev), sh.to(dev), None if res is None else res.to(dev), 2, relu, out=out))
    assert torch.equal(outs[0], outs[1])
    assert (outs[0].cpu() - ref).abs().max().item() <= 2e-5


# ------------------------------------------------------------------------------- first conv on a cat volume, 2-D form
@pytest.mark.parametrize("B,C,Co,D,H,W", [(2, 32, 32, 16, 12, 64), (1, 32, 32, 48, 9, 240), (1, 8, 32, 4, 5, 12),
                                          (2, 6, 20, 8, 7, 16), (1, 32, 32, 2 * 4, 6, 20)])
@pytest.mark.usefixtures("single_chain")
def test_catconv_first_layer(dev, B, C, Co, D, H, W):
    """dres0[0] on the concatenation volume WITHOUT the volume (csrc/catconv.hip) against (a) an FP64 evaluation of the
    reference arithmetic -- F.conv3d on the oracle's cat_fms volume -- and (b) this library's own 3-D kernel on the
    materialised volume.  Same FP32 products, per-dz grouping of the sums: both stay within 2e-5 of the FP64 value on
    O(1) outputs, every border included (z = 0 / D - 1, the four columns next to x == z, x == W - 1)."""
    ops = _ops()
    L, R = _rand((B, C, H, W), 301), _rand((B, C, H, W), 302)
    w = _rand((Co, 2 * C, 3, 3, 3), 303, 1.0 / math.sqrt(2 * C * 27 / 8))
    sc, sh = _affine(Co, 304)
    idx = ops.disp_index_list(D, 0, 1)
    assert ops.catconv_applicable(L.to(dev), R.to(dev), idx, Co)
    vol = O.cat_fms(L, R, D, 0, 1)
    ref = F.conv3d(vol.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1)
    for relu in (False, True):
        want = F.relu(ref) if relu else ref
        got = ops.catconv_first(L.to(dev), R.to(dev), D, ops.catconv_pack(w.to(dev)), sc.to(dev), sh.to(dev), relu).cpu()
        assert got.shape == want.shape
        assert (got.double() - want).abs().max().item() <= 2e-5
    if Co == 32:
        mat = ops.conv3d_k3(ops.cat_fms(L.to(dev), R.to(dev), idx), ops.pack_conv3d_weights(w.to(dev)), Co, sc.to(dev), sh.to(dev),
                            None, 1, True).cpu()
        assert (got - mat).abs().max().item() <= 4e-5      # two FP32 evaluations, each within 2e-5 of the FP64 value
    # the five 2-D convolutions as one multi-job launch (default), on three streams, and one after the other: the same
    # arithmetic on the same operands
    for mode in ("streams", "serial"):
        ops.set_first_layer_mode(mode)
        try:
            other = ops.catconv_first(L.to(dev), R.to(dev), D, ops.catconv_pack(w.to(dev)), sc.to(dev), sh.to(dev), True).cpu()
        finally:
            ops.set_first_layer_mode("merged")
        assert torch.equal(other, got), mode


def test_catconv_not_applicable_shapes(dev):
    ops = _ops()
    L = _rand((1, 4, 5, 16), 1).to(dev)
    assert not ops.catconv_applicable(L, L, ops.disp_index_list(12, 0, 1), 32)      # image narrower than the band + 8
    assert not ops.catconv_applicable(L, L, ops.disp_index_list(8, -2, 1), 32)      # start_disp != 0
    assert not ops.catconv_applicable(L, L, ops.disp_index_list(8, 0, 2), 32)       # dilation 2
    assert not ops.catconv_applicable(L, L, ops.disp_index_list(4, 0, 1), 64)       # more than 32 output channels
    assert not ops.catconv_applicable(L[..., :14].contiguous(), L[..., :14].contiguous(), ops.disp_index_list(4, 0, 1), 32)


@pytest.mark.parametrize("B,C,H,W,D", [(2, 32, 7, 40, 9), (1, 5, 3, 21, 33), (1, 16, 4, 64, 16)])
def test_correlation1d_cost_reference_semantics(dev, B, C, H, W, D):
    """COR_FUNCS['default'] = the reference's correlation1d_cost (correlation1d_cost.py:7-27): sum over ALL channels, the
    first D of the sampler's 2D-1 offsets (channel j = disparity D-1-j), leaky_relu(0.1), 4-D output.  Oracle = the
    sampler's published semantics (parity UNPINNED: the sampler package is not in the reference tree)."""
    from densematchingbenchmark_amd.modeling.stereo.cost_processors.utils.gwc_fms import COR_FUNCS
    L, R = _rand((B, C, H, W), 71), _rand((B, C, H, W), 72)
    got = COR_FUNCS["default"](L.to(dev), R.to(dev), max_disp=D, start_disp=0, dilation=1, disp_sample=None).cpu()
    want = O.correlation1d_cost(L, R, D)
    assert got.shape == (B, D, H, W)
    assert (got - want).abs().max().item() <= 1e-5 * max(1.0, math.sqrt(C))
    # channel D-1 is disparity 0: the plain per-pixel dot product
    dot = F.leaky_relu((L * R).sum(1), 0.1)
    assert (got[:, D - 1] - dot).abs().max().item() <= 1e-5 * max(1.0, math.sqrt(C))
    with pytest.raises(NotImplementedError):
        COR_FUNCS["default"](L.to(dev), R.to(dev), max_disp=D, kernel_size=3)


@pytest.mark.parametrize("B,C,Co,D,H,W", [(2, 32, 32, 24, 10, 36), (1, 32, 32, 8, 6, 156), (1, 6, 32, 4, 5, 12)])
def test_catconv_first_layer_on_a_difference_volume(dev, B, C, Co, D, H, W):
    """The same 2-D form for dif_fms (StereoNet, BASELINE configs[4]): conv(L - R shifted) = conv_L(w) + conv_R(-w)."""
    ops = _ops()
    L, R = _rand((B, C, H, W), 311), _rand((B, C, H, W), 312)
    w = _rand((Co, C, 3, 3, 3), 313, 1.0 / math.sqrt(C * 27 / 8))
    sc, sh = _affine(Co, 314)
    vol = O.dif_fms(L, R, D, 0, 1)
    ref = F.relu(F.conv3d(vol.double(), w.double(), None, padding=1) * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1))
    got = ops.catconv_first(L.to(dev), R.to(dev), D, ops.catconv_pack(w.to(dev), "dif"), sc.to(dev), sh.to(dev), True).cpu()
    assert got.shape == ref.shape and (got.double() - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("B,Hq,Wq", [(2, 5, 8), (1, 9, 12), (1, 7, 52), (2, 9, 100)])
def test_conf_head_composed_with_learned_upsampling(dev, B, Hq, Wq):
    """AcfNet's confidence head on a cost volume that is the learned 4x up-sampling of a quarter-resolution volume: the
    composed quarter-resolution form (16 phase-wise 3x3 convolutions of that volume + 1x1 + sigmoid, outer pixel ring
    directly) against the head evaluated on the up-sampled volume itself by the CPU oracle and by the direct kernel."""
    ops = _ops()
    Dq, M = 48, 64
    c = _rand((B, Dq, Hq, Wq), 401)
    w8 = _rand((1, 1, 8, 8, 8), 402, 0.2)
    w1 = _rand((M, 4 * Dq, 3, 3), 403, 1.0 / math.sqrt(4 * Dq * 9 / 4))
    w2 = _rand((M,), 404, 0.3)
    sc, sh = _affine(M, 405)
    cost_ref = F.conv_transpose3d(c.unsqueeze(1).double(), w8.double(), stride=4, padding=2).squeeze(1)
    hid = F.relu(F.conv2d(cost_ref, w1.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1))
    want = torch.sigmoid((hid * w2.double().view(1, -1, 1, 1)).sum(1, keepdim=True))
    cq = c.to(dev)
    w8d = w8.to(dev)
    cost = ops.deconv3d_k8s4_c1(cq, w8d.view(8, 8, 8))
    ops.UpsampleSource.attach(cost, cq, w8d)
    assert ops.conf_head_composite_applicable(cost, M)
    comp = ops.conf_head_k8s4_pack(w1.to(dev), w8d, sc.to(dev), sh.to(dev))
    got = ops.conf_head_from_source(cost, comp, sc.to(dev), sh.to(dev), w2.to(dev)).cpu()
    direct = ops.conf_head(cost, ops.pack_conf_head_weights(w1.to(dev)), sc.to(dev), sh.to(dev), w2.to(dev)).cpu()
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() <= 5e-6 and (direct.double() - want).abs().max().item() <= 5e-6
    # the same path with the hidden tensor materialised and reduced by dmb_conf_gather_f32 (a different summation order)
    ops.set_conf_dot_epilogue(False)
    try:
        via_hidden = ops.conf_head_from_source(cost, comp, sc.to(dev), sh.to(dev), w2.to(dev)).cpu()
    finally:
        ops.set_conf_dot_epilogue(True)
    assert (via_hidden.double() - want).abs().max().item() <= 5e-6 and (via_hidden - got).abs().max().item() <= 2e-6
    # ... and with the dword staging path of the convolution kernel (a misaligned quarter-resolution source), which the fused
    # epilogue shares
    cqm = _misaligned(c, dev)
    cost_m = ops.deconv3d_k8s4_c1(cqm, w8d.view(8, 8, 8))
    ops.UpsampleSource.attach(cost_m, cqm, w8d)
    scalar_path = ops.conf_head_from_source(cost_m, comp, sc.to(dev), sh.to(dev), w2.to(dev)).cpu()
    assert torch.equal(scalar_path, got)
    cost.add_(0.0)                                        # a modified tensor no longer matches its note
    assert not ops.conf_head_composite_applicable(cost, M)


@pytest.mark.parametrize("shape", [(2, 6, 5, 12), (1, 48, 9, 16), (1, 3, 7, 260)])
def test_deconv_k8s4_zcol_with_regression_is_bit_identical(dev, shape):
    """The z-column form of AcfNet's learned up-sampling: its volume equals dmb_deconv3d_k8s4_c1_f32's bit for bit (same
    per-output fma order), the folded disparity equals soft_argmin of that volume bit for bit."""
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, D, H, W), 501).to(dev)
    w = _rand((8, 8, 8), 502, 0.2).to(dev)
    vals = ops.disp_sample_values(4 * D, 0, 1)
    ref = ops.deconv3d_k8s4_c1(x, w)
    cost, disp = ops.deconv3d_k8s4_c1_soft_argmin(x, w, vals, 1.0)
    assert torch.equal(cost, ref)
    assert torch.equal(disp, ops.soft_argmin(ref, vals, 1.0, True))
    only, none = ops.deconv3d_k8s4_c1_soft_argmin(x, w, None)
    assert none is None and torch.equal(only, ref)


@pytest.mark.parametrize("Co,shape", [(32, (1, 5, 6, 120)), (64, (2, 3, 5, 60)), (32, (1, 4, 7, 64))])
@pytest.mark.usefixtures("single_chain")
def test_deconv3d_vector_and_scalar_epilogues_agree(dev, Co, shape):
    """The transposed convolution's 16-byte epilogue (two x parities interleaved through a per-wave LDS scratch, residual
    ring) against its 8-byte scattered form: same fma / add / max sequence per output -> bit-identical, with and without the
    skip operand, both ReLU placements.  The scattered form is what a misaligned output takes (the caller's ``out``)."""
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, 64, D, H, W), 601).to(dev)
    w = _rand((64, Co, 3, 3, 3), 602, 0.05).to(dev)
    wp = ops.pack_deconv3d_weights(w)
    sc, sh = _affine(Co, 603)
    res = _rand((B, Co, 2 * D, 2 * H, 2 * W), 604).to(dev)
    for r, relu in ((None, True), (res, True), (res, False), (res, "pre")):
        outs = []
        for out in (None, _misaligned(torch.zeros(B, Co, 2 * D, 2 * H, 2 * W), dev)):
            outs.append(ops.deconv3d_k3s2(x, wp, Co, sc.to(dev), sh.to(dev), r, relu, out=out))
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("Ci,Co,shape", [(64, 32, (1, 5, 6, 120)), (64, 64, (2, 3, 5, 60)), (32, 32, (1, 4, 7, 124)),
                                         (64, 32, (2, 6, 3, 240)), (128, 64, (1, 2, 3, 116)), (48, 32, (1, 1, 1, 60)),
                                         (64, 64, (4, 12, 34, 60)), (64, 32, (2, 24, 68, 120))])
def test_deconv3d_parity_class_items_match_the_two_parity_form(dev, Ci, Co, shape):
    """csrc/deconv3d_zy.hip -- work items (tile, z parity, y parity), four class bodies with their own chunk sizes, items handed
    out through per-XCD counters in the caller's workspace, three workgroups per CU -- against deconv3d_kernel (both y parities
    per item, static tile walk: what ``workspace=None`` selects): the same ascending (channel, tap) fma chain per output, so
    BIT-identical, with and without the skip operand and for both ReLU placements; and against the CPU transposed convolution.
    Shapes: partial tiles in x (124, 116) and z (D = 5, 3, 1), a single row, 2 .. 8 chunks per class, the hourglass's
    quarter-resolution layer.  The workspace must come back zeroed from every launch."""
    ops = _ops()
    B, D, H, W = shape
    xc = _rand((B, Ci, D, H, W), 611)
    wc = _rand((Ci, Co, 3, 3, 3), 612, 1.0 / math.sqrt(Ci * 27 / 8))
    sc, sh = _affine(Co, 613)
    x, w = xc.to(dev), wc.to(dev)
    wp = ops.pack_deconv3d_weights(w)
    res = _rand((B, Co, 2 * D, 2 * H, 2 * W), 614).to(dev)
    ws = torch.zeros(_lib_ws_ints(), dtype=torch.int32, device=dev)
    ref = F.conv_transpose3d(xc, wc, None, stride=2, padding=1, output_padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    for r, relu in ((None, False), (None, True), (res, True), (res, False), (res, "pre")):
        new = ops.deconv3d_k3s2(x, wp, Co, sc.to(dev), sh.to(dev), r, relu, workspace=ws)
        assert int(ws.abs().sum().item()) == 0          # the last workgroup to leave has reset every counter
        old = ops.deconv3d_k3s2(x, wp, Co, sc.to(dev), sh.to(dev), r, relu, workspace=None)
        assert torch.equal(new, old), (r is not None, relu, (new - old).abs().max().item())
        if r is None and relu is False:
            assert (new.cpu() - ref).abs().max().item() <= 2e-5
    # the same call again on the same workspace, and on the host layer's per-stream workspace
    again = ops.deconv3d_k3s2(x, wp, Co, sc.to(dev), sh.to(dev), res, True, workspace=ws)
    auto = ops.deconv3d_k3s2(x, wp, Co, sc.to(dev), sh.to(dev), res, True)
    assert torch.equal(again, auto) and torch.equal(again, ops.deconv3d_k3s2(x, wp, Co, sc.to(dev), sh.to(dev), res, True, workspace=None))


# ------------------------------------------------------------------------------------- spatial propagation scan (dmb.ops.spn)
@pytest.mark.parametrize("horizontal,reverse", [(True, False), (True, True), (False, False), (False, True)])
@pytest.mark.parametrize("shape", [(2, 3, 6, 9), (1, 2, 64, 80), (1, 1, 1100, 5), (1, 2, 7, 1500)])
def test_spn_gaterecurrent2d(dev, horizontal, reverse, shape):
    """GateRecurrent2dnoind (dmb/ops/spn, the reference's only native op; ONE launch per scan here instead of one per scanned
    line) against the oracle's restatement of the recurrence (UNPINNED: the CUDA reference cannot run here), forward and --
    through torch.autograd -- backward against the oracle's own autograd with an FP64 evaluation as yardstick.  Lines across
    the scan longer than 1024 positions take two positions per thread; longer than 2046 are refused (by the forward already: the
    backward's limit)."""
    from densematchingbenchmark_amd.ops import GateRecurrent2dnoind
    N, C, H, W = shape
    if (H if horizontal else W) > 2046:
        pytest.skip("line too long")
    g = torch.Generator().manual_seed(41)
    X = torch.randn(shape, generator=g)
    Gs = [torch.rand(shape, generator=g) * 0.33 for _ in range(3)]     # gates sum below 1: a contraction, as after AnyNet's normalisation
    up = torch.randn(shape, generator=g)
    leaves = [t.clone().to(dev).requires_grad_() for t in [X] + Gs]
    out = GateRecurrent2dnoind(horizontal, reverse)(*leaves)
    ref = O.spn_gaterecurrent2d(X, *Gs, horizontal, reverse)
    assert out.shape == ref.shape and (out.detach().cpu() - ref).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item())
    out.backward(up.to(dev))
    def grads(dtype):
        ls = [t.clone().to(dtype).requires_grad_() for t in [X] + Gs]
        O.spn_gaterecurrent2d(*ls, horizontal, reverse).backward(up.to(dtype))
        return [t.grad for t in ls]
    g32, g64 = grads(torch.float32), grads(torch.float64)
    for got, r32, r64 in zip(leaves, g32, g64):
        scale = max(1.0, r64.abs().max().item())
        e_got = (got.grad.cpu().double() - r64).abs().max().item()
        e_ref = (r32.double() - r64).abs().max().item()
        assert e_got <= max(4 * e_ref, 2e-6 * scale), (e_got, e_ref)


def test_spn_refuses_cpu_tensors_and_long_lines(dev):
    from densematchingbenchmark_amd import _lib
    from densematchingbenchmark_amd.ops import GateRecurrent2dnoind
    x = torch.zeros(1, 1, 4, 4)
    with pytest.raises(_lib.DmbLibraryError):
        GateRecurrent2dnoind(True, False)(x, x, x, x)
    y = torch.zeros(1, 1, 2100, 3, device=dev)
    with pytest.raises(_lib.DmbLibraryError):
        GateRecurrent2dnoind(True, False)(y, y, y, y)


@pytest.mark.parametrize("Ci,shape", [(64, (1, 4, 5, 60)), (64, (2, 3, 34, 60)), (20, (1, 2, 1, 60)), (64, (4, 12, 34, 60)),
                                      (64, (1, 3, 9, 32)), (64, (2, 5, 7, 44)), (64, (1, 2, 24, 64))])
@pytest.mark.usefixtures("single_chain")
def test_conv3d_linear_runs_for_narrow_planes(dev, Ci, shape):
    """The 64-channel stride-1 layer on 64-voxel runs of the (y, x) plane (S1Cfg LIN: rows of 32 .. 64 voxels -- the deepest
    hourglass level; 768 equal workgroups for [4, 64, 12, 34, 60], exactly three per CU) against the flattened dword form (what a
    misaligned input takes) -- the same ascending (channel, tap) fma chain per output, so BIT-identical, with residual and both
    ReLU placements -- and against the CPU convolution.  Runs that cross row ends, a last run shorter than 64, odd plane
    counts, a single row, widths below and at the capacity of the staged rows."""
    ops = _ops()
    B, D, H, W = shape
    xc = _rand((B, Ci, D, H, W), 621)
    wc = _rand((64, Ci, 3, 3, 3), 622, 1.0 / math.sqrt(Ci * 27))
    sc, sh = _affine(64, 623)
    x, wp = xc.to(dev), ops.pack_conv3d_weights(wc.to(dev))
    xm = _misaligned(xc, dev)
    res = _rand((B, 64, D, H, W), 624).to(dev)
    for r, relu in ((None, False), (None, True), (res, True), (res, "pre")):
        outs = [ops.conv3d_k3(xx, wp, 64, sc.to(dev), sh.to(dev), r, 1, relu) for xx in (x, xm)]
        assert torch.equal(outs[0], outs[1]), (r is not None, relu, (outs[0] - outs[1]).abs().max().item())
    ref = F.conv3d(xc, wc, None, padding=1) * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
    got = ops.conv3d_k3(x, wp, 64, sc.to(dev), sh.to(dev), None, 1, False).cpu()
    assert (got - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("Co,shape", [(64, (2, 3, 5, 78)), (32, (1, 4, 6, 130)), (64, (1, 2, 24, 78))])
@pytest.mark.usefixtures("single_chain")
def test_row_padded_hourglass_level_matches_the_unpadded_one(dev, Co, shape):
    """Rows that are not a 16-byte multiple (W % 4 == 2: the KITTI hourglass's deepest level is 78 columns wide) padded with
    zero columns to the next multiple of 4 -- the stride-1 unit over the padded tensor, its padding columns cleared again
    (dmb_zero_columns_f32), and the transposed unit writing only the real 2 W output columns (``Wout``) -- against the same two
    units on the unpadded tensor (dword kernels): the same fma chain per output, so BIT-identical."""
    ops = _ops()
    B, D, H, W = shape
    Wp = (W + 3) // 4 * 4
    xc = _rand((B, 64, D, H, W), 701)
    x = xc.to(dev)
    w4 = _rand((64, 64, 3, 3, 3), 702, 1.0 / math.sqrt(64 * 27)).to(dev)
    w5 = _rand((64, Co, 3, 3, 3), 703, 0.05).to(dev)
    sc4, sh4 = _affine(64, 704)
    sc5, sh5 = _affine(Co, 705)
    res = _rand((B, Co, 2 * D, 2 * H, 2 * W), 706).to(dev)
    wp4, wp5 = ops.pack_conv3d_weights(w4), ops.pack_deconv3d_weights(w5)
    assert ops.padded_rows_applicable(x, Co)
    mid = ops.conv3d_k3(x, wp4, 64, sc4.to(dev), sh4.to(dev), None, 1, True)
    want = ops.deconv3d_k3s2(mid, wp5, Co, sc5.to(dev), sh5.to(dev), res, True)
    xp = ops.copy_window(x, Wp, 0)
    assert tuple(xp.shape) == (B, 64, D, H, Wp) and torch.equal(xp[..., :W], x) and float(xp[..., W:].abs().max()) == 0.0
    midp = ops.conv3d_k3(xp, wp4, 64, sc4.to(dev), sh4.to(dev), None, 1, True)
    ops.zero_columns_(midp, W)
    assert torch.equal(midp[..., :W], mid) and float(midp[..., W:].abs().max()) == 0.0
    got = ops.deconv3d_k3s2(midp, wp5, Co, sc5.to(dev), sh5.to(dev), res, True, out_width=2 * W)
    assert tuple(got.shape) == tuple(want.shape) and torch.equal(got, want)
    # the padded-row form exists only with the workspace kernels: anything else says so
    from densematchingbenchmark_amd._lib import DmbLibraryError
    with pytest.raises(DmbLibraryError):
        ops.deconv3d_k3s2(midp, wp5, Co, sc5.to(dev), sh5.to(dev), res, True, out_width=2 * W, workspace=None)


# ------------------------------------------------------------------------------- round 6: split-K forms for launches that leave the chip idle
@pytest.mark.parametrize("kind,Ci,Co,shape", [("s1", 64, 64, (1, 4, 16, 32)), ("s2", 64, 64, (1, 8, 32, 64)), ("s2", 32, 64, (1, 16, 64, 128)),
                                              ("deconv", 64, 64, (1, 4, 16, 32)), ("deconv", 64, 32, (1, 8, 32, 64)), ("c1", 32, 1, (1, 16, 64, 128)),
                                              ("s1", 32, 32, (1, 3, 10, 44)), ("s2", 48, 32, (2, 5, 9, 52)), ("deconv", 16, 64, (1, 3, 5, 20)),
                                              # 300 work items on 256 workgroups: the persistent walk -- with a second input buffer
                                              # (stride 1: 2 x 74 KB) and with one (stride 2: 138 KB) -- and two batch items
                                              ("s1", 64, 64, (1, 5, 12, 80)), ("s2", 64, 64, (1, 10, 24, 160)), ("s1", 64, 64, (2, 3, 10, 76))])
def test_split_k_forms_of_small_launches(dev, kind, Ci, Co, shape):
    """The layers of the hourglass (hourglass.py:62-86) and the heads (PSMNet.py:46-54) at the sizes ONE 256x512 pair gives them
    (BASELINE configs[0]; dmb/apis/inference.py:191-225 serves one pair per call) take the split-K forms (csrc/conv3d_sk.hip,
    conv3d_c1s_kernel): within 2e-5 of the CPU convolution, within 2e-5 of the single-chain kernels (the same FP32 products, the
    partial sums of a voxel added in another fixed order), NOT bit-identical to them at the configs[0] shapes (that is how the test
    knows the form was taken), and reproducible bit for bit from run to run."""
    ops = _ops()
    B, D, H, W = shape
    x = _rand((B, Ci, D, H, W), 601)
    if kind == "c1":
        w = _rand((1, Ci, 3, 3, 3), 602, 1.0 / math.sqrt(Ci * 27))
        res = _rand((B, 1, D, H, W), 603)
        ref = F.conv3d(x, w, None, padding=1) - 0.5 + res
        run = lambda: ops.conv3d_k3_c1(x.to(dev), w.to(dev), -0.5, res.to(dev))   # noqa: E731
    else:
        sc, sh = _affine(Co, 604)
        if kind == "deconv":
            w = _rand((Ci, Co, 3, 3, 3), 602, 1.0 / math.sqrt(Ci * 27 / 8))
            y = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1)
            wp = ops.pack_deconv3d_weights(w.to(dev))
        else:
            w = _rand((Co, Ci, 3, 3, 3), 602, 1.0 / math.sqrt(Ci * 27))
            y = F.conv3d(x, w, None, stride=2 if kind == "s2" else 1, padding=1)
            wp = ops.pack_conv3d_weights(w.to(dev))
        y = y * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)
        res = _rand(y.shape, 603)
        ref = F.relu(y + res)
        if kind == "deconv":
            run = lambda: ops.deconv3d_k3s2(x.to(dev), wp, Co, sc.to(dev), sh.to(dev), res.to(dev), True)   # noqa: E731
        else:
            run = lambda: ops.conv3d_k3(x.to(dev), wp, Co, sc.to(dev), sh.to(dev), res.to(dev), 2 if kind == "s2" else 1, True)   # noqa: E731
    assert ops.split_k()
    got = run()
    assert (got.cpu() - ref).abs().max().item() <= 2e-5
    assert torch.equal(run(), got)
    ops.set_split_k(False)
    try:
        single = run()
    finally:
        ops.set_split_k(True)
    assert (single.cpu() - ref).abs().max().item() <= 2e-5
    assert (single - got).abs().max().item() <= 2e-5
    if (B, D, H, W) in ((1, 4, 16, 32), (1, 8, 32, 64), (1, 16, 64, 128)):
        assert not torch.equal(single, got)


def test_torch_extension_shim_and_ctypes_bind_the_same_functions(dev):
    """The thin torch extension (csrc/torch_shim.cpp; the reference's native-op pattern, dmb/ops/spn/src/gaterecurrent2dnoind_cuda.cpp:86-89)
    is loaded on a GPU box and gives, call for call, what the ctypes binding of the same C ABI gives -- outputs bit for bit, the
    library's own error type for a host tensor, a wrong shape, a foreign dtype."""
    from densematchingbenchmark_amd import _lib
    ops = _ops()
    shim = _lib.shim()
    assert shim is not None, _lib.shim_state()
    x = _rand((2, 32, 5, 9, 40), 701).to(dev)
    w = _rand((64, 32, 3, 3, 3), 702, 0.05).to(dev)
    wd = _rand((32, 64, 3, 3, 3), 703, 0.05).to(dev)
    sc, sh = (t.to(dev) for t in _affine(64, 704))
    res = _rand((2, 64, 5, 9, 40), 705).to(dev)
    x2 = _rand((2, 16, 20, 36), 706).to(dev)
    w2 = _rand((32, 16, 3, 3), 707, 0.1).to(dev)
    q = _rand((1, 6, 5, 8), 708).to(dev)

    def run():
        wp = ops.pack_conv3d_weights(w)
        outs = [ops.conv3d_k3(x, wp, 64, sc, sh, res, 1, True), ops.conv3d_k3(x, wp, 64, sc, sh, None, 2, "pre"),
                ops.deconv3d_k3s2(x, ops.pack_deconv3d_weights(wd), 64, sc, sh, None, True),
                ops.conv3d_k3_c1(x, w[:1].contiguous(), 0.5, None), ops.copy_window(x, 48, -3),
                ops.conv2d(x2, ops.pack_conv2d_weights(w2), 32, 3, 1, 1, sc[:32].contiguous(), sh[:32].contiguous(), None, True)]
        outs += list(ops.trilinear_ac_soft_argmin(q, (24, 20, 32), ops.disp_sample_values(24, 0, 1), 1.0))
        # (round 6) the per-unit launches of a training step: BatchNorm forward / backward, the 2-D weight gradient
        rm, rv, nbt = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.tensor(2, dtype=torch.int64, device=dev)
        fwd = ops.bn_train_fwd(res, sc, sh, rm, rv, nbt, 0.1, 1e-5, res * 0.5, True)
        bwd = ops.bn_act_bwd(res * 0.3, res, fwd[0], fwd[3], fwd[4], fwd[1], fwd[2], True, True, want_dres=True, dres_acc=res * 0.1)
        outs += list(fwd) + [rm, rv, nbt.float()] + list(bwd) + [ops.conv2d_wgrad(x2, _rand((2, 32, 20, 36), 709).to(dev), 3, 1)]
        return outs

    try:
        a = run()
        _lib._shim = None
        b = run()
    finally:
        _lib._shim = shim
    assert len(a) == len(b) == 21 and all(torch.equal(u, v) for u, v in zip(a, b))
    for bad in (lambda: ops.conv3d_k3(x.cpu(), ops.pack_conv3d_weights(w), 64), lambda: ops.conv3d_k3(x, ops.pack_conv3d_weights(w), 32),
                lambda: ops.conv3d_k3(x.double(), ops.pack_conv3d_weights(w), 64), lambda: ops.conv3d_k3(x, ops.pack_conv3d_weights(w), 64, None, None, res[:1]),
                lambda: ops.bn_train_fwd(res.cpu(), sc, sh), lambda: ops.bn_train_fwd(res, sc[:8].contiguous(), sh),
                lambda: ops.bn_act_bwd(res[:1], res, None, sc, sh, sc, sh, False, True), lambda: ops.conv2d_wgrad(x2, x2[:, :, :5].contiguous(), 3, 1)):
        with pytest.raises(_lib.DmbLibraryError):
            bad()


@pytest.mark.parametrize("Ci,Co,dil,shape", [(64, 64, 1, (1, 64, 128)), (128, 128, 2, (1, 64, 128)), (32, 32, 1, (1, 128, 256)), (64, 128, 1, (2, 9, 44)),
                                             (128, 32, 1, (1, 6, 28)), (16, 64, 2, (3, 11, 36))])
def test_split_k_form_of_small_conv2d_launches(dev, Ci, Co, dil, shape):
    """The 3x3 layers of the backbones (layers/basic_layers.py:12-66, backbones/PSMNet.py:8-129) for ONE small image per view take the
    split-K form (csrc/conv3d_sk.hip with KZ = 1: [1, 64, 64, 128] 30 -> 11 us): within 2e-5 of the CPU convolution with folded
    BatchNorm, skip and ReLU, within 2e-5 of the tile kernel (DMB_CONV_SINGLE_CHAIN), not bit-identical to it at the 256x512 shapes,
    reproducible run to run, and correct into / out of channel windows of wider tensors (the SPP concat is written in place)."""
    ops = _ops()
    B, H, W = shape
    x = _rand((B, Ci + 8, H, W), 801)
    w = _rand((Co, Ci, 3, 3), 802, 1.0 / math.sqrt(Ci * 9))
    sc, sh = _affine(Co, 803)
    res = _rand((B, Co + 32, H, W), 804)
    ref = F.relu(F.conv2d(x[:, 8:], w, None, padding=dil, dilation=dil) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res[:, 32:])
    wp = ops.pack_conv2d_weights(w.to(dev))

    def run():
        out = torch.full((B, Co + 16, H, W), -7.0, device=dev)
        ops.conv2d(x.to(dev), wp, Co, 3, 1, dil, sc.to(dev), sh.to(dev), res.to(dev), True, in_window=(8, Ci), out=out, out_ch_offset=16,
                   res_ch_offset=32)
        return out

    got = run()
    assert (got[:, 16:].cpu() - ref).abs().max().item() <= 2e-5 and float(got[:, :16].min()) == -7.0 == float(got[:, :16].max())
    assert torch.equal(run(), got)
    ops.set_split_k(False)
    try:
        tiles = run()
    finally:
        ops.set_split_k(True)
    assert (tiles[:, 16:].cpu() - ref).abs().max().item() <= 2e-5 and (tiles - got).abs().max().item() <= 2e-5
    if H * W >= 64 * 128:
        assert not torch.equal(tiles, got)


@pytest.mark.parametrize("Co,C,kind", [(32, 32, "cat"), (32, 32, "dif"), (16, 12, "cat"), (8, 5, "dif"), (32, 7, "cat")])
def test_catconv_pack_in_one_launch(dev, Co, C, kind):
    """dmb_catconv_pack_weights_f32 (ABI 8) against the packs spelled out with tensor slicing + dmb_conv2d_pack_weights_f32
    (ops.catconv_pack_torch, the form of rounds 2-5): all five packs bit for bit, for both volume kinds and odd channel counts."""
    from densematchingbenchmark_amd import ops
    g = torch.Generator().manual_seed(Co * 100 + C)
    w = torch.randn((Co, C if kind == "dif" else 2 * C, 3, 3, 3), generator=g).to(dev)
    a, b = ops.catconv_pack(w, kind), ops.catconv_pack_torch(w, kind)
    assert a["Co"] == b["Co"] and a["Cin"] == b["Cin"]
    for k in ("A", "B1", "B2", "HC", "HD"):
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
