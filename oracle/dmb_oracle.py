"""CPU ORACLE for the cost-volume -> 3-D aggregation -> disparity-regression path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``densematchingbenchmark_amd/`` imports this file; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the
reported CPU baseline -- never as the thing shipped or measured as the product.

What it is: an independent, functional (no nn.Module) PyTorch-CPU restatement of the reference's algorithm
for every function on the hot path, each citing the reference file:line it follows
(paths relative to the reference tree).  Parameters are plain dicts keyed by the reference's own
``state_dict`` names, so a reference checkpoint can drive the oracle directly.

Pinning: ``oracle/gen_golden.py`` imports the real reference (in the build container only) and records
inputs/outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here against
those vectors, including the reference's own known-answer cases (tests/.../test_cat_fms.py:30-40 and
test_disp_predictors.py:42-102).  The group-wise correlation volume has NO reference implementation
(README.md:16 only names GwcNet): for that one function parity is UNPINNED and the spec is SURVEY 8-a4.  It has two
witnesses that share no code with it (tests/test_oracle_golden.py::test_oracle_gwc_has_two_witnesses, and the same on the
GPU): with one channel per group it must equal the product of the two halves of cat_fms's volume (cat_fms IS pinned bit
for bit), and with one group, rescaled by C, correlation1d_cost's pre-activation channels in disparity order
(correlation1d_cost.py:12-25; itself unpinned: the sampler package is not in the reference tree).

Training side: the ``*_train_step`` / ``*_backward`` functions differentiate the same restatements with
torch.autograd in training mode (``bn_training()``); they are pinned too -- gen_golden.py section 4h runs one training
iteration of the reference's OWN modules (PSMNet cost path, AcfNet with its confidence network, the PSMNet backbone, the
whole StereoNet model; ``train()`` mode, its loss classes, its autograd) and ``tests/golden/training.npz`` holds the losses
and a fingerprint of every gradient.

The primitive index formulas (conv / transposed conv / trilinear) are additionally restated as plain C loop
nests in ``oracle/dmb_oracle_c.c`` and cross-checked against this file on small shapes.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm3d default, basic_layers.py:75


# ------------------------------------------------------------------------------------------------------------
# disparity samples
# ------------------------------------------------------------------------------------------------------------
def disp_index_list(max_disp, start_disp=0, dilation=1):
    """cost_processors/utils/cat_fms.py:26-35: linspace(start, end, n) then int() (truncation)."""
    end_disp = start_disp + max_disp - 1
    n = (max_disp + dilation - 1) // dilation
    return [int(v) for v in torch.linspace(start_disp, end_disp, n)]


def disp_sample_values(max_disp, start_disp=0, dilation=1):
    """disp_predictors/faster_soft_argmin.py:33-44, soft_argmin.py:41-43."""
    end_disp = start_disp + max_disp - 1
    n = (max_disp + dilation - 1) // dilation
    return torch.linspace(start_disp, end_disp, n)


# ------------------------------------------------------------------------------------------------------------
# cost-volume builders
# ------------------------------------------------------------------------------------------------------------
def _valid_x(W, d):
    """Columns kept for disparity d: cat_fms.py:36-44."""
    if d > 0:
        return slice(d, W), slice(0, W - d)
    if d == 0:
        return slice(0, W), slice(0, W)
    return slice(0, W + d), slice(-d, W)


def cat_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None):
    """cost_processors/utils/cat_fms.py:7-48 (output is always FP32, :32)."""
    N, C, H, W = reference_fm.shape
    idx = disp_index_list(max_disp, start_disp, dilation)
    out = torch.zeros(N, 2 * C, len(idx), H, W, dtype=torch.float32)
    for k, d in enumerate(idx):
        if abs(d) >= W:
            continue
        xs, xt = _valid_x(W, d)
        out[:, :C, k, :, xs] = reference_fm[:, :, :, xs]
        out[:, C:, k, :, xs] = target_fm[:, :, :, xt]
    return out


def dif_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None):
    """cost_processors/utils/dif_fms.py:7-46."""
    N, C, H, W = reference_fm.shape
    idx = disp_index_list(max_disp, start_disp, dilation)
    out = torch.zeros(N, C, len(idx), H, W, dtype=torch.float32)
    for k, d in enumerate(idx):
        if abs(d) >= W:
            continue
        xs, xt = _valid_x(W, d)
        out[:, :, k, :, xs] = reference_fm[:, :, :, xs] - target_fm[:, :, :, xt]
    return out


def fast_disp_samples(max_disp, start_disp=0, dilation=1):
    """cat_fms.py:55-63 / dif_fms.py:54-62: linspace(start, end, D) -- NOT truncated to integers, unlike the default
    builders (max_disp 192, dilation 2 gives a step of 191/95)."""
    D = (max_disp + dilation - 1) // dilation
    return torch.linspace(start_disp, start_disp + max_disp - 1, D).float()


def inverse_warp_3d(img, disp):
    """layers/inverse_warp_3d.py:4-52 for a [B, C, H, W] image expanded over the D planes of disp [B, D, H, W], zero
    padding: a restatement of F.grid_sample's 5-D "bilinear" (tri-linear) algorithm as the reference reaches it on
    torch >= 1.3 -- the grid is normalised with (size - 1) (:41-43) but sampled with align_corners=False, i.e. the source
    index is ((g + 1) * size - 1) / 2 -- in FP32, operation by operation, neighbours accumulated in the sampler's order
    (top/bottom = z, north/south = y, west/east = x)."""
    f = np.float32
    img = img.detach().cpu().numpy().astype(f)
    disp = disp.detach().cpu().numpy().astype(f)
    B, D, H, W = disp.shape
    C = img.shape[1]
    if min(D, H, W) < 2:
        raise ValueError("inverse_warp_3d divides by (size - 1)")
    with np.errstate(all="ignore"):
        gd = np.broadcast_to(np.arange(D, dtype=f).reshape(1, D, 1, 1), disp.shape)
        gh = np.broadcast_to(np.arange(H, dtype=f).reshape(1, 1, H, 1), disp.shape)
        gw = np.arange(W, dtype=f).reshape(1, 1, 1, W) + disp
        gd = (gd / f(D - 1) * f(2)) - f(1)
        gh = (gh / f(H - 1) * f(2)) - f(1)
        gw = (gw / f(W - 1) * f(2)) - f(1)
        ix = ((gw + f(1)) * f(W) - f(1)) / f(2)
        iy = ((gh + f(1)) * f(H) - f(1)) / f(2)
        iz = ((gd + f(1)) * f(D) - f(1)) / f(2)
        x0, y0, z0 = np.floor(ix), np.floor(iy), np.floor(iz)
        x1, y1, z1 = x0 + f(1), y0 + f(1), z0 + f(1)
        # (weight, z index, y index, x index) in accumulation order tnw, tne, tsw, tse, bnw, bne, bsw, bse
        terms = [((x1 - ix) * (y1 - iy) * (z1 - iz), z0, y0, x0), ((ix - x0) * (y1 - iy) * (z1 - iz), z0, y0, x1),
                 ((x1 - ix) * (iy - y0) * (z1 - iz), z0, y1, x0), ((ix - x0) * (iy - y0) * (z1 - iz), z0, y1, x1),
                 ((x1 - ix) * (y1 - iy) * (iz - z0), z1, y0, x0), ((ix - x0) * (y1 - iy) * (iz - z0), z1, y0, x1),
                 ((x1 - ix) * (iy - y0) * (iz - z0), z1, y1, x0), ((ix - x0) * (iy - y0) * (iz - z0), z1, y1, x1)]
        out = np.zeros((B, C, D, H, W), dtype=f)
        bidx = np.arange(B).reshape(B, 1, 1, 1)
        for wgt, zz, yy, xx in terms:
            ok = (zz >= 0) & (zz < D) & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W) & np.isfinite(xx)
            yi = np.clip(np.nan_to_num(yy), 0, H - 1).astype(np.int64)
            xi = np.clip(np.nan_to_num(xx), 0, W - 1).astype(np.int64)
            for c in range(C):
                v = img[:, c][bidx, yi, xi]                      # the expanded image is the same on every plane
                out[:, c] = np.where(ok, out[:, c] + v * wgt, out[:, c])
    return torch.from_numpy(out)


def _fast_samples(reference_fm, max_disp, start_disp, dilation, disp_sample):
    B, C, H, W = reference_fm.shape
    if disp_sample is None:
        ds = fast_disp_samples(max_disp, start_disp, dilation)
        return ds.view(1, -1, 1, 1).expand(B, ds.numel(), H, W)
    return disp_sample.float()


def fast_cat_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None):
    """cat_fms.py:51-82: target features warped by -disp_sample, reference features masked where the warped target is
    not positive (:77), concatenated."""
    ds = _fast_samples(reference_fm, max_disp, start_disp, dilation, disp_sample)
    tgt = inverse_warp_3d(target_fm, -ds)
    ref = reference_fm.float().unsqueeze(2) * (tgt > 0).float()
    return torch.cat((ref, tgt), dim=1)


def fast_dif_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None, normalize=False, p=1.0):
    """dif_fms.py:49-86."""
    ds = _fast_samples(reference_fm, max_disp, start_disp, dilation, disp_sample)
    tgt = inverse_warp_3d(target_fm, -ds)
    dif = reference_fm.float().unsqueeze(2) * (tgt > 0).float() - tgt
    if normalize:
        dif = torch.norm(dif, p=p, dim=1, keepdim=False)
    return dif


def fast_volume_grads(reference_fm, target_fm, grad_out, max_disp=192, start_disp=0, dilation=1, disp_sample=None, kind="cat",
                      dtype=torch.float32, normalize=False, p=1.0, wrt_samples=False):
    """Gradients of fast_cat_fms / fast_dif_fms (``kind``) with respect to the two feature maps -- and, with ``wrt_samples``, the
    per-pixel samples (third return value) -- for an upstream gradient
    ``grad_out``, the way the reference gets them: torch.autograd through the sampler of layers/inverse_warp_3d.py:19-50 (the
    image expanded over the D planes, a (size - 1)-normalised grid, F.grid_sample with its align_corners=False default).  The
    mask ``(warped > 0)`` is a constant (cat_fms.py:77 builds it with ``.type_as``: no gradient path).  Pinned by
    tests/golden/fast_volumes_grad.npz (the reference's own functions under autograd).  ``dtype=torch.float64`` = yardstick."""
    L = reference_fm.detach().to(dtype).requires_grad_()
    R = target_fm.detach().to(dtype).requires_grad_()
    ds = _fast_samples(reference_fm, max_disp, start_disp, dilation, disp_sample).detach().to(dtype)
    if wrt_samples:
        ds = ds.clone().requires_grad_()
    B, D, H, W = ds.shape
    C = R.shape[1]
    img = R.unsqueeze(2).expand(B, C, D, H, W)
    gd = torch.linspace(0, D - 1, D, dtype=dtype).view(1, D, 1, 1).expand(B, D, H, W)
    gh = torch.linspace(0, H - 1, H, dtype=dtype).view(1, 1, H, 1).expand(B, D, H, W)
    gw = torch.linspace(0, W - 1, W, dtype=dtype).view(1, 1, 1, W).expand(B, D, H, W) + (-ds)
    grid = torch.stack(((gw / (W - 1) * 2) - 1, (gh / (H - 1) * 2) - 1, (gd / (D - 1) * 2) - 1), dim=4)
    tgt = F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    ref = L.unsqueeze(2) * (tgt > 0).to(dtype)
    vol = torch.cat((ref, tgt), dim=1) if kind == "cat" else ref - tgt
    if normalize:                                                     # dif_fms.py:82-84
        vol = torch.norm(vol, p=p, dim=1, keepdim=False)
    vol.backward(grad_out.to(dtype))
    return (L.grad, R.grad, ds.grad) if wrt_samples else (L.grad, R.grad)


def correlation1d_cost(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None):
    """cost_processors/utils/correlation1d_cost.py:7-27.  The arithmetic lives in SpatialCorrelationSampler
    (ClementPinard/Pytorch-Correlation-extension, branch fix_1.7 per INSTALL.md:60-66; no version pin in requirements.txt),
    which is NOT in the reference tree: restated from its published semantics -- PARITY UNPINNED.
    sampler(kernel_size=1, patch_size=(1, 2D-1), stride=1, padding=0, dilation_patch=1):
        out[b, 0, pw, y, x] = sum_c in1[b,c,y,x] * in2[b,c,y, x + pw - (D-1)]   (0 outside in2)
    then ``[:, :max_disp]`` keeps pw in [0, D-1] (offsets -(D-1) .. 0) and leaky_relu(0.1) is applied (:21-25)."""
    B, C, H, W = reference_fm.shape
    out = reference_fm.new_zeros((B, max_disp, H, W))
    for j in range(max_disp):
        d = max_disp - 1 - j
        if d < W:
            out[:, j, :, d:] = (reference_fm[:, :, :, d:] * target_fm[:, :, :, :W - d]).sum(dim=1)
    return F.leaky_relu(out, negative_slope=0.1)


def first_layer_from_maps(reference_fm, target_fm, weight, max_disp, kind="cat"):
    """conv3d(V, weight, padding=1) for V = cat_fms(L, R) (or dif_fms: ``kind="dif"``) with unit disparity step, WITHOUT V:
    the decomposition csrc/catconv.hip executes, restated with torch's 2-D convolutions (checker for that file; the
    arithmetic it must reproduce is cat_fms.py:7-48 / dif_fms.py:7-46 followed by aggregators/PSMNet.py:31-33,58).
        out[z] = sum_{dz: 0 <= z+dz-1 < D} ( F_{dz, m}[y, x] + H_dz[y, x - (z + dz - 1)] ),  m = max(0, dz - (x - z))
    F_{dz, m} = 3x3 conv of L with weight[:, :C, dz] restricted to taps dx >= m; H_dz = 3x3 conv of R (zero-extended to
    the left: that IS the mask x >= z' of the right half) with weight[:, C:, dz]; at x = W-1 without the dx = 2 tap."""
    L, R, w = reference_fm, target_fm, weight
    B, C, H, W = L.shape
    D = max_disp
    wl, wr = (w, -w) if kind == "dif" else (w[:, :C], w[:, C:])
    ext = D + 2

    def taps(k, lo, hi):      # keep dx in [lo, hi)
        k = k.clone()
        k[..., :lo] = 0
        k[..., hi:] = 0
        return k

    Rz = F.pad(R, (ext, 0))                                                       # column j <-> n = j - ext
    Fm = [[F.conv2d(L, taps(wl[:, :, dz], m, 3), padding=1) for dz in range(3)] for m in range(3)]
    Hc = [F.conv2d(Rz, wr[:, :, dz], padding=1) for dz in range(3)]
    Hd = [F.conv2d(Rz, taps(wr[:, :, dz], 0, 2), padding=1) for dz in range(3)]
    x = torch.arange(W)
    out = L.new_zeros((B, w.shape[0], D, H, W))
    for z in range(D):
        for dz in range(3):
            zp = z + dz - 1
            if zp < 0 or zp >= D:
                continue
            g = Hc[dz][..., x - zp + ext].clone()
            g[..., W - 1] = Hd[dz][..., W - 1 - zp + ext]
            m = (dz - (x - z)).clamp(min=0)
            f = torch.zeros_like(g)
            for mm in range(3):
                f = torch.where(m == mm, Fm[mm][dz], f)
            out[:, :, z] += f + g
    return out


def gwc_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, num_groups=40, disp_sample=None):
    """Group-wise correlation (GwcNet, "gwc" volume).  NOT IN THE REFERENCE -- parity unpinned; spec SURVEY 8-a4:
    mean over the C/G channels of a group of L[c, y, x] * R[c, y, x - d], zero outside the valid columns, using the
    same shifting convention as cat_fms.py:36-44."""
    N, C, H, W = reference_fm.shape
    assert C % num_groups == 0
    cg = C // num_groups
    idx = disp_index_list(max_disp, start_disp, dilation)
    out = torch.zeros(N, num_groups, len(idx), H, W, dtype=reference_fm.dtype if reference_fm.dtype == torch.float64 else torch.float32)
    for k, d in enumerate(idx):
        if abs(d) >= W:
            continue
        xs, xt = _valid_x(W, d)
        prod = reference_fm[:, :, :, xs] * target_fm[:, :, :, xt]
        out[:, :, k, :, xs] = prod.view(N, num_groups, cg, H, prod.shape[-1]).mean(dim=2)
    return out


# ------------------------------------------------------------------------------------------------------------
# spatial propagation scan: dmb/ops/spn (the reference's only native op; CUDA-only there: PARITY UNPINNED)
# ------------------------------------------------------------------------------------------------------------
def spn_gaterecurrent2d(X, G1, G2, G3, horizontal, reverse):
    """dmb/ops/spn/src/gaterecurrent2dnoind_kernel.cu:130-166 (one scanned line; the other three directions :168-286 differ in
    the index arithmetic only) driven by :535-552: restated with differentiable torch ops, one line at a time --
        H[s, t] = (1 - g1 - g2 - g3) * X[s, t] + g1 * H[s', t-1] + g2 * H[s', t] + g3 * H[s', t+1]
    with s' the line scanned before s and g_k = G_k[s, t] where the neighbour exists, else 0 (get_gate_sf, :85-98: the gate of a
    link is stored at the later of its two positions, :10-66).  The reference cannot be run here (no CUDA; its Python wrapper
    refuses CPU tensors, functions/gaterecurrent2dnoind.py:14-16): this restatement is UNPINNED; its autograd is what the HIP
    backward is checked against (an independent derivation of kernel.cu:288-345)."""
    if not horizontal:   # scan along the rows: the same recurrence on the transposed planes
        return spn_gaterecurrent2d(X.transpose(2, 3), G1.transpose(2, 3), G2.transpose(2, 3), G3.transpose(2, 3), True,
                                   reverse).transpose(2, 3)
    N, C, H, W = X.shape
    order = range(W - 1, -1, -1) if reverse else range(W)
    lines = {}
    prev = None
    for i, w in enumerate(order):
        x = X[..., w]
        if prev is None:
            h = x * 1.0
        else:
            zero = torch.zeros_like(x[..., :1])
            up = torch.cat([zero, prev[..., :-1]], dim=-1)      # H[s', t - 1]
            dn = torch.cat([prev[..., 1:], zero], dim=-1)       # H[s', t + 1]
            g1 = torch.cat([zero, G1[..., 1:, w]], dim=-1)      # no neighbour above row 0
            g2 = G2[..., w]
            g3 = torch.cat([G3[..., :-1, w], zero], dim=-1)     # none below the last row
            h = (((1.0 - g1) - g2) - g3) * x + ((g1 * up + g2 * prev) + g3 * dn)
        lines[w] = h
        prev = h
    return torch.stack([lines[w] for w in range(W)], dim=-1)


# ------------------------------------------------------------------------------------------------------------
# conv + BN (+ReLU) units: layers/basic_layers.py:68-100,160-177
# ------------------------------------------------------------------------------------------------------------
_BN_TRAINING = [False]


class bn_training(object):
    """``with bn_training():`` -- BatchNorm layers use batch statistics and update the running buffers in ``p`` in place
    (nn.BatchNorm default momentum 0.1), as model.train() does for the reference."""

    def __enter__(self):
        _BN_TRAINING.append(True)

    def __exit__(self, *exc):
        _BN_TRAINING.pop()


def _bn_eval(x, p, prefix):
    """BatchNorm with running statistics (eval mode; batch statistics inside ``bn_training()``); prefix names the
    nn.BatchNorm3d/2d module."""
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    mean, var = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    return F.batch_norm(x, mean, var, w, b, training=_BN_TRAINING[-1], momentum=0.1, eps=BN_EPS)


def conv3d_unit(x, p, prefix, stride=1, batch_norm=True, relu=False):
    """conv3d_bn / conv3d_bn_relu (basic_layers.py:68-83,160-177): Sequential(Conv3d k3 p1, [BN], [ReLU]);
    ``prefix`` is the Sequential's name, children are .0 (conv), .1 (BN)."""
    y = F.conv3d(x, p[prefix + ".0.weight"], p.get(prefix + ".0.bias"), stride=stride, padding=1)
    if batch_norm:
        y = _bn_eval(y, p, prefix + ".1")
    return F.relu(y) if relu else y


def deconv3d_unit(x, p, prefix, batch_norm=True):
    """deconv3d_bn (basic_layers.py:86-100) as used by hourglass.py:52-60: k3, stride 2, padding 1, output_padding 1."""
    y = F.conv_transpose3d(x, p[prefix + ".0.weight"], p.get(prefix + ".0.bias"), stride=2, padding=1, output_padding=1)
    if batch_norm:
        y = _bn_eval(y, p, prefix + ".1")
    return y


def hourglass(x, presqu, postsqu, p, prefix, batch_norm=True):
    """cost_processors/utils/hourglass.py:62-86."""
    out = conv3d_unit(x, p, prefix + ".conv1", stride=2, batch_norm=batch_norm, relu=True)  # :64
    pre = conv3d_unit(out, p, prefix + ".conv2", batch_norm=batch_norm)  # :66
    pre = F.relu(pre + postsqu) if postsqu is not None else F.relu(pre)  # :67-70
    out = conv3d_unit(pre, p, prefix + ".conv3", stride=2, batch_norm=batch_norm, relu=True)  # :73
    out = conv3d_unit(out, p, prefix + ".conv4", batch_norm=batch_norm, relu=True)  # :75
    skip = presqu if presqu is not None else pre
    post = F.relu(deconv3d_unit(out, p, prefix + ".conv5", batch_norm=batch_norm) + skip)  # :78-81
    out = deconv3d_unit(post, p, prefix + ".conv6", batch_norm=batch_norm)  # :84
    return out, pre, post


def _trunk(raw_cost, p, prefix, batch_norm):
    """Shared wiring of PSMAggregator.forward (PSMNet.py:55-72) and AcfAggregator.forward (AcfNet.py:59-76)."""
    c0 = conv3d_unit(raw_cost, p, prefix + "dres0.0", batch_norm=batch_norm, relu=True)
    c0 = conv3d_unit(c0, p, prefix + "dres0.1", batch_norm=batch_norm, relu=True)
    t = conv3d_unit(c0, p, prefix + "dres1.0", batch_norm=batch_norm, relu=True)
    cost0 = conv3d_unit(t, p, prefix + "dres1.1", batch_norm=batch_norm) + c0
    out1, pre1, post1 = hourglass(cost0, None, None, p, prefix + "dres2", batch_norm)
    out1 = out1 + cost0
    out2, pre2, post2 = hourglass(out1, pre1, post1, p, prefix + "dres3", batch_norm)
    out2 = out2 + cost0
    out3, pre3, post3 = hourglass(out2, pre2, post2, p, prefix + "dres4", batch_norm)
    out3 = out3 + cost0

    def classif(x, name):
        h = conv3d_unit(x, p, prefix + name + ".0", batch_norm=batch_norm, relu=True)
        return F.conv3d(h, p[prefix + name + ".1.weight"], p.get(prefix + name + ".1.bias"), padding=1)

    cost1 = classif(out1, "classif1")
    cost2 = classif(out2, "classif2") + cost1
    cost3 = classif(out3, "classif3") + cost2
    return cost1, cost2, cost3


def psm_aggregator(raw_cost, p, max_disp, prefix="", batch_norm=True, upsample=True):
    """cost_processors/aggregators/PSMNet.py:55-95.  Returns [cost3, cost2, cost1] (best first)."""
    B, C, D, H, W = raw_cost.shape
    costs = _trunk(raw_cost, p, prefix, batch_norm)
    if not upsample:
        return [c.squeeze(1) for c in reversed(costs)]
    size = [max_disp, H * 4, W * 4]
    up = [F.interpolate(c, size, mode="trilinear", align_corners=True).squeeze(1) for c in costs]  # :77-93
    return [up[2], up[1], up[0]]


def acf_aggregator(raw_cost, p, max_disp, prefix="", batch_norm=True):
    """cost_processors/aggregators/AcfNet.py:59-90 (learned k8/s4 ConvTranspose3d upsampling, :81-83)."""
    B, C, D, H, W = raw_cost.shape
    costs = _trunk(raw_cost, p, prefix, batch_norm)
    up = []
    for i, c in enumerate(costs):
        w = p[prefix + "deconv%d.weight" % (i + 1)]
        up.append(F.conv_transpose3d(c, w, None, stride=4, padding=2).squeeze(1))
    assert up[0].shape[1:] == (max_disp, H * 4, W * 4)
    return [up[2], up[1], up[0]]


def stereonet_aggregator(raw_cost, p, prefix="", batch_norm=True, num=4):
    """cost_processors/aggregators/StereoNet.py:42-55."""
    x = raw_cost
    for i in range(num):
        x = conv3d_unit(x, p, prefix + "classify.%d" % i, batch_norm=batch_norm, relu=True)
    cost = F.conv3d(x, p[prefix + "lastconv.weight"], p.get(prefix + "lastconv.bias"), padding=1)
    return [cost.squeeze(1)]


# ------------------------------------------------------------------------------------------------------------
# disparity predictors
# ------------------------------------------------------------------------------------------------------------
def soft_argmin(cost_volume, max_disp, start_disp=0, dilation=1, alpha=1.0, normalize=True, disp_sample=None):
    """disp_predictors/soft_argmin.py:45-75."""
    c = cost_volume * alpha
    prob = F.softmax(c, dim=1) if normalize else c
    B, D, H, W = c.shape
    if disp_sample is None:
        ds = disp_sample_values(max_disp, start_disp, dilation)
        assert D == ds.numel()
        disp_sample = ds.view(1, D, 1, 1).expand(B, D, H, W)
    return torch.sum(prob * disp_sample, dim=1, keepdim=True)


def faster_soft_argmin(cost_volume, max_disp, start_disp=0, dilation=1, alpha=1.0, normalize=True):
    """disp_predictors/faster_soft_argmin.py:51-75: the weighted sum as a frozen Conv3d(1,1,(D,1,1))."""
    c = cost_volume * alpha
    prob = F.softmax(c, dim=1) if normalize else c
    w = disp_sample_values(max_disp, start_disp, dilation).view(1, 1, -1, 1, 1)
    return F.conv3d(prob.unsqueeze(1), w).squeeze(1)


def soft_argmin_f64(cost_volume, max_disp, start_disp=0, dilation=1, alpha=1.0):
    """FP64 evaluation of the same formula: the 'truth' both FP32 variants of the reference approximate."""
    c = cost_volume.double() * alpha
    prob = F.softmax(c, dim=1)
    ds = disp_sample_values(max_disp, start_disp, dilation).double().view(1, -1, 1, 1)
    return torch.sum(prob * ds, dim=1, keepdim=True)


def local_soft_argmin(cost_volume, max_disp, radius, start_disp=0, dilation=1, radius_dilation=1, alpha=1.0):
    """disp_predictors/local_soft_argmin.py:48-105.  Returns (disp, argmax index)."""
    B, D, H, W = cost_volume.shape
    assert D == (max_disp + dilation - 1) // dilation
    max_index = torch.argmax(cost_volume, dim=1, keepdim=True)  # :65
    interval = torch.linspace(-radius * radius_dilation, radius * radius_dilation, 2 * radius + 1).long()  # :69-71
    index_group = max_index + interval.view(1, -1, 1, 1)  # :76
    mask = ((index_group >= 0) & (index_group <= D - 1)).type_as(cost_volume)  # :81
    index_group = index_group.clamp(0, D - 1)  # :82
    gathered = torch.gather(cost_volume, 1, index_group)  # :86
    disp_sample = start_disp + index_group.type_as(cost_volume) * dilation  # :89-92
    gathered = gathered * alpha  # :97
    prob = F.softmax(gathered * mask + (1 - mask) * (-10000.0 * alpha), dim=1)  # :100
    return (prob * disp_sample).sum(dim=1, keepdim=True), max_index  # :103


# ------------------------------------------------------------------------------------------------------------
# AcfNet confidence head: cmn/cmn.py:10-36,57-69
# ------------------------------------------------------------------------------------------------------------
def conf_head(cost, p, prefix, batch_norm=True):
    """ConfHead.forward (cmn.py:34-36) + sigmoid (cmn.py:65): returns (conf, conf_cost)."""
    h = F.conv2d(cost, p[prefix + ".conf_net.0.0.weight"], None, padding=1)
    if batch_norm:
        h = _bn_eval(h, p, prefix + ".conf_net.0.1")
    h = F.relu(h)
    conf_cost = F.conv2d(h, p[prefix + ".conf_net.1.weight"], None)
    return torch.sigmoid(conf_cost), conf_cost


def cmn_eval(costs, p, alpha, beta, prefix="conf_heads", batch_norm=True):
    """Cmn.get_confidence / forward in eval (cmn.py:57-84): returns (cost_vars, confs)."""
    confs = [conf_head(c, p, "%s.%d" % (prefix, i), batch_norm)[0] for i, c in enumerate(costs)]
    cost_vars = [alpha * (1 - c) + beta for c in confs]
    return cost_vars, confs


# ------------------------------------------------------------------------------------------------------------
# evaluation: data/datasets/evaluation/stereo/{eval.py:12-31, pixel_error.py:6-73}, tools/test.py:304-307
# ------------------------------------------------------------------------------------------------------------
def remove_padding(batch, size):
    pad_top = batch.shape[-2] - size[-2]
    if pad_top >= 0:
        batch = batch[:, :, pad_top:, :size[-1]]
    return batch


def calc_error(est_disp, gt_disp, lb=None, ub=None):
    mask = torch.ones(gt_disp.shape, dtype=torch.bool)
    if lb is not None:
        mask = mask & (gt_disp > lb)
    if ub is not None:
        mask = mask & (gt_disp < ub)
    if mask.float().sum() < 1.0:
        return {"1px": 0.0, "2px": 0.0, "3px": 0.0, "5px": 0.0, "epe": 0.0}
    gt, est = gt_disp[mask], est_disp[mask]
    abs_error = torch.abs(gt - est)
    total = mask.float().sum()
    out = {"%dpx" % k: float(torch.sum(torch.gt(abs_error, k).float()) / total * 100) for k in (1, 2, 3, 5)}
    out["epe"] = float(abs_error.float().mean())
    return out


def dataset_metrics(est_list, gt_list, original_size, lb, ub):
    """Unweighted mean over images of the per-image error dicts (mmcv LogBuffer.average, tools/test.py:304-307)."""
    keys = ("epe", "1px", "2px", "3px", "5px")
    acc = {k: 0.0 for k in keys}
    n = 0
    for est, gt in zip(est_list, gt_list):
        for b in range(est.shape[0]):
            e = calc_error(remove_padding(est[b:b + 1], original_size), remove_padding(gt[b:b + 1], original_size), lb, ub)
            for k in keys:
                acc[k] += e[k]
            n += 1
    return {k: acc[k] / max(n, 1) for k in keys}, n


# ------------------------------------------------------------------------------------------------------------
# data-side conventions in front of the path: data/transforms/stereo_trans.py (ToTensor :9-18, CenterCrop :20-44, Normalize :78-90,
# StereoPad :92-119) in the order of data/datasets/stereo/builder.py:22-28 and apis/inference.py:120-129,151-188
# (pinned by tests/golden/demo_sceneflow.npz: the reference's own inference_stereo on its demo pair, oracle/gen_golden_demo.py)
# ------------------------------------------------------------------------------------------------------------
IMAGENET_MEAN = (123.675, 116.28, 103.53)    # apis/inference.py:120-121
IMAGENET_STD = (58.395, 57.12, 57.375)


def image_to_chw(img_hwc_u8):
    """apis/inference.py:153-165 (and stereo/scene_flow/base.py:17-23): imread's uint8 [H, W, C] -> float32 [3, H, W], values 0..255."""
    return torch.from_numpy(np.ascontiguousarray(img_hwc_u8[:, :, :3].astype(np.float32).transpose(2, 0, 1)))


def stereo_pad(img, size):
    """stereo_trans.py:92-119: zeros on the TOP (th - h rows) and on the RIGHT (tw - w columns) of a [C, H, W] image."""
    h, w = img.shape[-2:]
    th, tw = size
    if (h, w) == (th, tw):
        return img
    return F.pad(img, [0, tw - w, th - h, 0], mode="constant", value=0)


def center_crop(img, size):
    """stereo_trans.py:20-44."""
    h, w = img.shape[-2:]
    th, tw = size
    if (h, w) == (th, tw):
        return img
    x1, y1 = (w - tw) // 2, (h - th) // 2
    return img[:, y1:y1 + th, x1:x1 + tw]


def normalize(img, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """stereo_trans.py:78-90 -> torchvision's normalize: (x - mean[c]) / std[c], a subtraction then a division in the tensor's
    dtype (so that a padded pixel, normalised AFTER the padding, holds -mean[c] / std[c])."""
    m = torch.as_tensor(mean, dtype=img.dtype).view(-1, 1, 1)
    s = torch.as_tensor(std, dtype=img.dtype).view(-1, 1, 1)
    return img.clone().sub_(m).div_(s)


def prepare_image(img_hwc_u8, pad_to_shape=None, crop_shape=None):
    """The image side of apis/inference.py:_prepare_data: -> [1, 3, th, tw] float32."""
    x = image_to_chw(img_hwc_u8)
    if pad_to_shape is not None:
        x = stereo_pad(x, pad_to_shape)
    if crop_shape is not None:
        x = center_crop(x, crop_shape)
    return normalize(x).unsqueeze(0)


def psmnet_model(left_img, right_img, p, max_disp):
    """GeneralizedStereoModel.forward in eval mode for PSMNet (models/general_stereo_model.py:42-90): images [B, 3, H, W] ->
    ([disp3, disp2, disp1], [cost3, cost2, cost1])."""
    lf, rf = psmnet_backbone(left_img, p), psmnet_backbone(right_img, p)
    return psmnet_path(lf, rf, p, max_disp)


# ------------------------------------------------------------------------------------------------------------
# whole path (what bench.py's cpu_baseline times): SURVEY 8-a1/a13
# ------------------------------------------------------------------------------------------------------------
def psmnet_path(ref_fms, tgt_fms, p, max_disp, scale=4, alpha=1.0, prefix="cost_processor.aggregator."):
    """CatCostProcessor.forward (cost_processors/builder.py:33-40) + [FasterSoftArgmin(c) for c in costs]
    (models/general_stereo_model.py:51-54) for the PSMNet config (configs/PSMNet/scene_flow.py:19-52)."""
    raw = cat_fms(ref_fms, tgt_fms, max_disp // scale, 0, 1)
    costs = psm_aggregator(raw, p, max_disp, prefix)
    disps = [faster_soft_argmin(c, max_disp, 0, 1, alpha, True) for c in costs]
    return disps, costs


def acfnet_path(ref_fms, tgt_fms, p, max_disp, cmn_alpha=1.0, cmn_beta=1.0, scale=4, alpha=1.0):
    """AcfNet eval forward of the path (configs/AcfNet/scene_flow_adaptive.py through general_stereo_model.py:48-90):
    cat volume -> AcfAggregator -> FasterSoftArgmin per level -> Cmn confidences.  ``p`` uses model-level names."""
    raw = cat_fms(ref_fms, tgt_fms, max_disp // scale, 0, 1)
    costs = acf_aggregator(raw, p, max_disp, "cost_processor.aggregator.")
    disps = [faster_soft_argmin(c, max_disp, 0, 1, alpha, True) for c in costs]
    cost_vars, confs = cmn_eval(costs, p, cmn_alpha, cmn_beta, prefix="cmn.conf_heads")
    return disps, costs, confs


def stereonet_path(ref_fms, tgt_fms, p, max_disp, scale=8, alpha=1.0, num=4):
    """StereoNet cost path (configs/StereoNet/scene_flow_8x_2stage.py): dif volume at 1/scale -> StereoNetAggregator
    -> FasterSoftArgmin over max_disp // scale samples (the refinement stages are outside the path)."""
    raw = dif_fms(ref_fms, tgt_fms, max_disp // scale, 0, 1)
    costs = stereonet_aggregator(raw, p, "cost_processor.aggregator.", num=num)
    disps = [faster_soft_argmin(c, max_disp // scale, 0, 1, alpha, True) for c in costs]
    return disps, costs


def gwcnet_path(ref_fms, tgt_fms, p, max_disp, num_groups=40, scale=4, alpha=1.0):
    """GwcNet-style path (no reference implementation; spec SURVEY 8-a4): [gwc(G groups) | cat(2 x 12)] volume ->
    PSMAggregator(in_planes = G + 24) -> FasterSoftArgmin.  ref_fms / tgt_fms = (correlation feats, concat feats)."""
    (lg, lc), (rg, rc) = ref_fms, tgt_fms
    raw = torch.cat([gwc_fms(lg, rg, max_disp // scale, 0, 1, num_groups), cat_fms(lc, rc, max_disp // scale, 0, 1)], dim=1)
    costs = psm_aggregator(raw, p, max_disp, "cost_processor.aggregator.")
    disps = [faster_soft_argmin(c, max_disp, 0, 1, alpha, True) for c in costs]
    return disps, costs


# ------------------------------------------------------------------------------------------------------------
# "next" row: PSMNet feature backbone (backbones/PSMNet.py:8-129, layers/basic_layers.py:31-46,105-123,219-243)
# ------------------------------------------------------------------------------------------------------------
def conv2d_unit(x, p, prefix, stride=1, dilation=1, ksize=3, batch_norm=True, relu=False):
    pad = dilation * (ksize // 2)
    y = F.conv2d(x, p[prefix + ".0.weight"], p.get(prefix + ".0.bias"), stride=stride, padding=pad, dilation=dilation)
    if batch_norm:
        y = _bn_eval(y, p, prefix + ".1")
    return F.relu(y) if relu else y


def basic_block(x, p, prefix, stride, dilation, has_down, batch_norm=True):
    """basic_layers.py:219-243."""
    out = conv2d_unit(x, p, prefix + ".conv1", stride, dilation, 3, batch_norm, relu=True)
    out = conv2d_unit(out, p, prefix + ".conv2", 1, dilation, 3, batch_norm)
    skip = conv2d_unit(x, p, prefix + ".downsample", stride, 1, 1, batch_norm) if has_down else x
    return out + skip


def psmnet_backbone(img, p, prefix="backbone.", batch_norm=True):
    """PSMNetBackbone._forward (backbones/PSMNet.py:82-123) for one image batch [B, 3, H, W] -> [B, 32, H/4, W/4]."""
    x = conv2d_unit(img, p, prefix + "firstconv.0", 2, 1, 3, batch_norm, True)
    x = conv2d_unit(x, p, prefix + "firstconv.1", 1, 1, 3, batch_norm, True)
    x = conv2d_unit(x, p, prefix + "firstconv.2", 1, 1, 3, batch_norm, True)
    for i in range(3):
        x = basic_block(x, p, prefix + "layer1.%d" % i, 1, 1, False, batch_norm)
    for i in range(16):
        x = basic_block(x, p, prefix + "layer2.%d" % i, 2 if i == 0 else 1, 1, i == 0, batch_norm)
    out4 = x
    for i in range(3):
        x = basic_block(x, p, prefix + "layer3.%d" % i, 1, 1, i == 0, batch_norm)
    for i in range(3):
        x = basic_block(x, p, prefix + "layer4.%d" % i, 1, 2, False, batch_norm)
    out8 = x
    H, W = out8.shape[-2:]
    branches = []
    for i, k in ((1, 64), (2, 32), (3, 16), (4, 8)):
        b = F.avg_pool2d(out8, (k, k), stride=(k, k))
        b = conv2d_unit(b, p, prefix + "branch%d.1" % i, 1, 1, 1, batch_norm, True)
        branches.append(F.interpolate(b, (H, W), mode="bilinear", align_corners=True))
    feat = torch.cat((out4, out8, branches[3], branches[2], branches[1], branches[0]), 1)
    y = conv2d_unit(feat, p, prefix + "lastconv.0", 1, 1, 3, batch_norm, True)
    return F.conv2d(y, p[prefix + "lastconv.1.weight"])


def stereonet_backbone(img, p, prefix="backbone.", batch_norm=True, downsample_num=3, residual_num=6):
    """StereoNetBackbone._forward (backbones/StereoNet.py:83-93) for one image batch [B, 3, H, W] -> [B, 32, H/8, W/8]."""
    x = img
    for i in range(downsample_num):
        x = F.conv2d(x, p[prefix + "downsample.%d.downsample.weight" % i], p[prefix + "downsample.%d.downsample.bias" % i],
                     stride=2, padding=2)
    for i in range(residual_num):
        x = basic_block(x, p, prefix + "residual_blocks.%d" % i, 1, 1, False, batch_norm)
    return F.conv2d(x, p[prefix + "lastconv.weight"], p[prefix + "lastconv.bias"], padding=1)


def gc_aggregator(raw_cost, p, prefix="", batch_norm=True):
    """GCAggregator.forward (aggregators/GCNet.py:70-120): raw [B, 64, D/2, H/2, W/2] -> [cost [B, D, H, W]]."""
    def conv(x, name, stride=1):
        return conv3d_unit(x, p, prefix + name, stride, batch_norm, relu=True)

    def deconv(x, name):
        return F.relu(deconv3d_unit(x, p, prefix + name, batch_norm))

    c18 = raw_cost
    c20 = conv(conv(c18, "layer19"), "layer20")
    c21 = conv(torch.cat([c18, c20], 1), "layer21", 2)
    c23 = conv(conv(c21, "layer22"), "layer23")
    c24 = conv(torch.cat([c21, c23], 1), "layer24", 2)
    c26 = conv(conv(c24, "layer25"), "layer26")
    c27 = conv(torch.cat([c24, c26], 1), "layer27", 2)
    c29 = conv(conv(c27, "layer28"), "layer29")
    c30 = conv(torch.cat([c27, c29], 1), "layer30", 2)
    c32 = conv(conv(c30, "layer31"), "layer32")
    c33 = deconv(c32, "layer33")
    c34 = deconv(c33 + c29, "layer34")
    c35 = deconv(c34 + c26, "layer35")
    c36 = deconv(c35 + c23, "layer36")
    c37 = F.conv_transpose3d(c36 + c20, p[prefix + "layer37.weight"], p[prefix + "layer37.bias"], stride=2, padding=1,
                             output_padding=1)
    return [c37.squeeze(1)]


def gcnet_backbone(img, p, prefix="backbone.", batch_norm=True):
    """GCNetBackbone (backbones/GCNet.py:26-38) for one image batch [B, 3, H, W] -> [B, 32, H/2, W/2]."""
    q = prefix + "backbone."
    x = conv2d_unit(img, p, q + "0", 2, 1, 5, batch_norm, True)
    for i in range(1, 9):
        x = basic_block(x, p, q + "%d" % i, 1, 1, False, batch_norm)
    return F.conv2d(x, p[q + "9.weight"], p[q + "9.bias"], padding=1)


def gcnet_path(ref_fms, tgt_fms, p, max_disp, alpha=1.0):
    """GC-Net after the backbone: cat volume at 1/2 resolution -> GCAggregator -> FasterSoftArgmin."""
    raw = cat_fms(ref_fms, tgt_fms, max_disp // 2, 0, 1)
    costs = gc_aggregator(raw, p, "cost_processor.aggregator.")
    return [faster_soft_argmin(c, max_disp, 0, 1, alpha, True) for c in costs], costs


def edge_aware_refinement(disp, left_image, p, prefix, batch_norm=True):
    """EdgeAwareRefinement.forward (disp_refinement/utils/edge_aware.py:44-68)."""
    h, w = left_image.shape[-2:]
    scale = w / disp.shape[-1]
    up = F.interpolate(disp, size=(h, w), mode="bilinear", align_corners=False) * scale
    x = conv2d_unit(torch.cat((up, left_image), 1), p, prefix + "conv_mix", 1, 1, 3, batch_norm, True)
    for i, d in enumerate((1, 2, 4, 8, 1, 1)):
        x = basic_block(x, p, prefix + "residual_dilation_blocks.%d" % i, 1, d, False, batch_norm)
    res = F.conv2d(x, p[prefix + "conv_res.weight"], p[prefix + "conv_res.bias"], padding=1)
    return F.relu(res + up)


def stereonet_refinement(disps, left_image, p, num=1, prefix="disp_refinement.", batch_norm=True):
    """StereoNetRefinement.forward (disp_refinement/StereoNet.py:39-61): best map first."""
    h, w = left_image.shape[-2:]
    init = disps[-1]
    scale = w / init.shape[-1]
    out = [F.interpolate(init, size=(h, w), mode="bilinear", align_corners=False) * scale]
    for i in range(num):
        out.append(edge_aware_refinement(out[-1], left_image, p, prefix + "refine_blocks.%d." % i, batch_norm))
    out.reverse()
    return out


def stereo_focal_loss(cost, gt, variance, max_disp, start_disp=0, dilation=1, coefficient=0.0):
    """StereoFocalLoss.loss_per_level at the cost volume's resolution with LaplaceDisp2Prob
    (losses/stereo_focal_loss.py:63-101, losses/utils/disp2prob.py:107-173); differentiable (torch autograd)."""
    lower, upper = start_disp, start_disp + max_disp
    end_disp = start_disp + max_disp - 1
    m1 = ((gt > lower) & (gt < upper)).to(gt.dtype)
    g = gt * m1
    n = (max_disp + dilation - 1) // dilation
    samples = torch.linspace(start_disp, end_disp, n).view(1, n, 1, 1)
    m2 = ((g > start_disp) & (g < end_disp)).to(gt.dtype)
    g = g * m2
    prob = F.softmax(-torch.abs(samples - g) / variance, dim=1) * m2 + 1e-40
    logq = F.log_softmax(cost, dim=1)
    weight = (1.0 - prob).pow(-coefficient)
    return -((prob * logq) * weight * m1).sum() / m1.sum().clamp(min=1.0)


def conf_nll_loss(conf_logits, gt, max_disp, start_disp=0):
    """ConfidenceNllLoss.loss_per_level (losses/conf_nll_loss.py:35-56)."""
    mask = ((gt > start_disp) & (gt < max_disp)).to(gt.dtype)
    return (-1.0 * F.logsigmoid(conf_logits) * mask).sum() / mask.sum().clamp(min=1.0)


def disp_smooth_l1_loss(est, gt, max_disp, start_disp=0):
    """DispSmoothL1Loss.loss_per_level (losses/smooth_l1_loss.py:36-58)."""
    mask = (gt > start_disp) & (gt < max_disp)
    if mask.sum() < 1:
        return (torch.abs(est - gt) * mask.float()).mean()
    return F.smooth_l1_loss(est[mask], gt[mask], reduction="mean")


# ---------------------------------------------------------------------------------------------- backward passes
# The reference trains through torch.autograd of the nn.Sequential(Conv3d | ConvTranspose3d, BatchNorm3d[, ReLU])
# factories (dmb/modeling/stereo/layers/basic_layers.py:68-100,160-177); the restatement differentiates the same
# functional ops on the CPU.  "dc" = gradient w.r.t. the raw convolution output.
def conv3d_backward(x, w, dc, stride=1, dtype=torch.float32):
    """(dx, dw) of F.conv3d(x, w, stride=stride, padding=1)."""
    x = x.detach().to(dtype).requires_grad_(True)
    w = w.detach().to(dtype).requires_grad_(True)
    y = F.conv3d(x, w, None, stride=stride, padding=1)
    dx, dw = torch.autograd.grad(y, (x, w), dc.to(dtype))
    return dx, dw


def deconv3d_backward(x, w, dy, dtype=torch.float32):
    """(dx, dw) of F.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1); w is [Ci, Co, 3, 3, 3]."""
    x = x.detach().to(dtype).requires_grad_(True)
    w = w.detach().to(dtype).requires_grad_(True)
    y = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1)
    dx, dw = torch.autograd.grad(y, (x, w), dy.to(dtype))
    return dx, dw


def bn_act_train(c, gamma, beta, residual=None, relu=0, eps=1e-5, dy=None, dtype=torch.float32, training=True,
                 running_mean=None, running_var=None, momentum=0.1):
    """nn.BatchNorm (training mode: batch statistics) + skip add + ReLU of a convolution unit, forward and -- when dy is
    given -- backward.  relu: 0 none, 1 after the skip add (hourglass.py:67-81), 2 before it (aggregators/GCNet.py:108-116).
    Returns dict(y[, dc, dgamma, dbeta, dres], running_mean, running_var)."""
    c = c.detach().to(dtype).requires_grad_(True)
    g = gamma.detach().to(dtype).requires_grad_(True)
    b = beta.detach().to(dtype).requires_grad_(True)
    r = residual.detach().to(dtype).requires_grad_(True) if residual is not None else None
    rm = running_mean.detach().to(dtype).clone() if running_mean is not None else None
    rv = running_var.detach().to(dtype).clone() if running_var is not None else None
    v = F.batch_norm(c, rm, rv, g, b, training=training, momentum=momentum, eps=eps)
    if relu == 2:
        v = F.relu(v)
    if r is not None:
        v = v + r
    if relu == 1:
        v = F.relu(v)
    out = dict(y=v.detach(), running_mean=rm, running_var=rv)
    if dy is not None:
        ins = (c, g, b) + ((r,) if r is not None else ())
        grads = torch.autograd.grad(v, ins, dy.to(dtype))
        out.update(dc=grads[0], dgamma=grads[1], dbeta=grads[2], dres=grads[3] if r is not None else None)
    return out


class _no_context(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def psmnet_train_step(ref_fms, tgt_fms, p, max_disp, gt, level_weights=(1.0, 0.7, 0.5), loss_weight=1.0, dtype=torch.float32,
                      training=True):
    """One training forward/backward of the PSMNet cost path as the reference runs it (models/general_stereo_model.py:60-77
    with losses/smooth_l1_loss.py): cat_fms -> PSMAggregator (BatchNorm in training mode) -> FasterSoftArgmin ->
    weighted DispSmoothL1Loss per level; gradients by torch.autograd.  Returns (losses, grads, running) where grads maps
    every learnable parameter (and 'ref_fms' / 'tgt_fms') to its gradient and running holds the updated BN buffers."""
    q, leaves = dict(), dict()
    for k, v in p.items():
        v = v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
            leaves[k] = v
        q[k] = v
    L = ref_fms.detach().clone().to(dtype).requires_grad_(True)
    R = tgt_fms.detach().clone().to(dtype).requires_grad_(True)
    leaves["ref_fms"], leaves["tgt_fms"] = L, R
    with (bn_training() if training else _no_context()):   # training=False: eval-mode BatchNorm (running statistics), same graph
        raw = cat_fms(L, R, max_disp // 4, 0, 1).to(dtype)               # the builder's output is FP32 (cat_fms.py:32)
        costs = psm_aggregator(raw, q, max_disp, "cost_processor.aggregator.")
        if dtype == torch.float32:
            disps = [faster_soft_argmin(c, max_disp, 0, 1, 1.0, True) for c in costs]
        else:   # the same formula evaluated in `dtype`
            ds = disp_sample_values(max_disp, 0, 1).to(dtype).view(1, -1, 1, 1)
            disps = [torch.sum(F.softmax(c, dim=1) * ds, dim=1, keepdim=True) for c in costs]
    losses = [level_weights[i] * loss_weight * disp_smooth_l1_loss(d, gt.to(dtype), max_disp) for i, d in enumerate(disps)]
    names = list(leaves)
    grads = torch.autograd.grad(sum(losses), [leaves[k] for k in names], allow_unused=True)
    running = {k: v for k, v in q.items() if "running_" in k}
    return [l.detach() for l in losses], dict(zip(names, grads)), running


def acfnet_train_step(ref_fms, tgt_fms, p, max_disp, gt, variance=1.2, coefficient=5.0, level_weights=(1.0, 0.7, 0.5),
                      focal_weight=1.0, l1_weight=0.1, adaptive=False, cmn_alpha=1.0, cmn_beta=1.0, nll_weight=8.0,
                      dtype=torch.float32):
    """One training iteration of AcfNet (configs/AcfNet/scene_flow_{uniform,adaptive}.py through
    models/general_stereo_model.py:60-77): cat_fms -> AcfAggregator (biased convolutions, BatchNorm in training mode, learned
    k8/s4 up-sampling) -> FasterSoftArgmin; StereoFocalLoss on the three cost volumes + DispSmoothL1Loss on the three
    disparity maps.  ``adaptive``: the confidence network (cmn/cmn.py:62-84) supplies the per-pixel variance
    alpha * (1 - sigmoid(conf_cost)) + beta and adds ConfidenceNllLoss on the confidence costs; otherwise the variance is the
    fixed number.  ``p``: model-level names.  Returns (dict of losses, grads, running)."""
    q, leaves = dict(), dict()
    for k, v in p.items():
        v = v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
            leaves[k] = v
        q[k] = v
    L = ref_fms.detach().clone().to(dtype).requires_grad_(True)
    R = tgt_fms.detach().clone().to(dtype).requires_grad_(True)
    leaves["ref_fms"], leaves["tgt_fms"] = L, R
    g = gt.to(dtype)
    losses = dict()
    with bn_training():
        raw = cat_fms(L, R, max_disp // 4, 0, 1).to(dtype)
        costs = acf_aggregator(raw, q, max_disp, "cost_processor.aggregator.")
        ds = disp_sample_values(max_disp, 0, 1).to(dtype).view(1, -1, 1, 1)
        disps = [torch.sum(F.softmax(c, dim=1) * ds, dim=1, keepdim=True) for c in costs]
        variances = [variance] * len(costs)
        if adaptive:
            conf_costs = [conf_head(c, q, "cmn.conf_heads.%d" % i)[1] for i, c in enumerate(costs)]
            variances = [cmn_alpha * (1 - torch.sigmoid(cc)) + cmn_beta for cc in conf_costs]
            for i, cc in enumerate(conf_costs):
                losses["conf_loss_lvl%d" % i] = nll_weight * level_weights[i] * conf_nll_loss(cc, g, max_disp)
    for i, (c, d) in enumerate(zip(costs, disps)):
        losses["stereo_focal_loss_lvl%d" % i] = focal_weight * level_weights[i] * stereo_focal_loss(c, g, variances[i], max_disp, 0, 1, coefficient)
        losses["l1_loss_lvl%d" % i] = l1_weight * level_weights[i] * disp_smooth_l1_loss(d, g, max_disp)
    names = list(leaves)
    grads = torch.autograd.grad(sum(losses.values()), [leaves[k] for k in names])
    return {k: v.detach() for k, v in losses.items()}, dict(zip(names, grads)), {k: v for k, v in q.items() if "running_" in k}


def acfnet_uniform_train_step(ref_fms, tgt_fms, p, max_disp, gt, **kw):
    return acfnet_train_step(ref_fms, tgt_fms, p, max_disp, gt, adaptive=False, **kw)


def psmnet_backbone_train_step(l_img, r_img, p, dl, dr, dtype=torch.float32, prefix="backbone."):
    """Training-mode forward / backward of the PSMNet backbone as the reference runs it (backbones/PSMNet.py:127-131: the left
    view, then the right view through the shared weights, BatchNorm statistics per call): gradients of sum(fl * dl) + sum(fr * dr)
    w.r.t. every learnable parameter.  Returns ((fl, fr), grads, running)."""
    q, leaves = dict(), dict()
    for k, v in p.items():
        v = v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
            leaves[k] = v
        q[k] = v
    with bn_training():
        fl = psmnet_backbone(l_img.to(dtype), q, prefix)
        fr = psmnet_backbone(r_img.to(dtype), q, prefix)
    names = list(leaves)
    grads = torch.autograd.grad((fl * dl.to(dtype)).sum() + (fr * dr.to(dtype)).sum(), [leaves[k] for k in names])
    return (fl.detach(), fr.detach()), dict(zip(names, grads)), {k: v for k, v in q.items() if "running_" in k}


def stereonet_e2e_train_step(l_img, r_img, p, max_disp, gt, level_weights=(1.0, 0.5), dtype=torch.float32, num_refine=1):
    """One training iteration of the WHOLE StereoNet-8x model (configs/StereoNet/scene_flow_8x_2stage.py through
    models/general_stereo_model.py:42-77): backbone on each view (BatchNorm statistics per call), difference volume at 1/8,
    StereoNetAggregator, FasterSoftArgmin, edge-aware refinement cascade, weighted DispSmoothL1Loss on [refined..., up-sampled
    coarse] (sparse ground truth: data.sparse=True -> no interpolation of gt; all maps here are full-size).
    ``p``: model-level names.  Returns (losses, grads, running)."""
    q, leaves = dict(), dict()
    for k, v in p.items():
        v = v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()
        if v.is_floating_point() and "running_" not in k and "disp_regression" not in k:
            v.requires_grad_(True)
            leaves[k] = v
        q[k] = v
    li, ri, g = l_img.to(dtype), r_img.to(dtype), gt.to(dtype)
    with bn_training():
        fl = stereonet_backbone(li, q)
        fr = stereonet_backbone(ri, q)
        raw = dif_fms(fl, fr, max_disp // 8, 0, 1).to(dtype)
        cost = stereonet_aggregator(raw, q, "cost_processor.aggregator.")[0]
        ds = disp_sample_values(max_disp // 8, 0, 1).to(dtype).view(1, -1, 1, 1)
        disp = torch.sum(F.softmax(cost, dim=1) * ds, dim=1, keepdim=True)
        disps = stereonet_refinement([disp], li, q, num=num_refine)
    losses = [level_weights[i] * disp_smooth_l1_loss(d, g, max_disp) for i, d in enumerate(disps)]
    names = list(leaves)
    grads = torch.autograd.grad(sum(losses), [leaves[k] for k in names], allow_unused=True)
    return [l.detach() for l in losses], dict(zip(names, grads)), {k: v for k, v in q.items() if "running_" in k}


def gcnet_e2e_train_step(l_img, r_img, p, max_disp, gt, dtype=torch.float32, training=True):
    """One training iteration of the whole GC-Net model (configs/GCNet/scene_flow.py through models/general_stereo_model.py:42-77):
    backbone on each view (BatchNorm statistics per call), concatenation volume at 1/2 resolution, GCAggregator, FasterSoftArgmin,
    DispSmoothL1Loss.  ``p``: model-level names.  Returns (loss, grads, running)."""
    q, leaves = dict(), dict()
    for k, v in p.items():
        v = v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()
        if v.is_floating_point() and "running_" not in k and "disp_regression" not in k:
            v.requires_grad_(True)
            leaves[k] = v
        q[k] = v
    with (bn_training() if training else _no_context()):   # training=False: the same graph with running statistics
        fl = gcnet_backbone(l_img.to(dtype), q)
        fr = gcnet_backbone(r_img.to(dtype), q)
        raw = cat_fms(fl, fr, max_disp // 2, 0, 1).to(dtype)
        cost = gc_aggregator(raw, q, "cost_processor.aggregator.")[0]
        ds = disp_sample_values(max_disp, 0, 1).to(dtype).view(1, -1, 1, 1)
        disp = torch.sum(F.softmax(cost, dim=1) * ds, dim=1, keepdim=True)
    loss = disp_smooth_l1_loss(disp, gt.to(dtype), max_disp)
    names = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    return loss.detach(), dict(zip(names, grads)), {k: v for k, v in q.items() if "running_" in k}


def stereonet_train_step(ref_fms, tgt_fms, p, max_disp, gt, dtype=torch.float32, num=4):
    """Training forward/backward of the StereoNet cost path at the volume's own resolution (dif_fms -> StereoNetAggregator
    with biased convolutions, BatchNorm in training mode -> soft-argmin -> smooth-L1 against ``gt`` [B, 1, H, W] given at
    that resolution).  ``p``: aggregator-level names.  Returns (loss, grads, running)."""
    q, leaves = dict(), dict()
    for k, v in p.items():
        v = v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
            leaves[k] = v
        q[k] = v
    L = ref_fms.detach().clone().to(dtype).requires_grad_(True)
    R = tgt_fms.detach().clone().to(dtype).requires_grad_(True)
    leaves["ref_fms"], leaves["tgt_fms"] = L, R
    with bn_training():
        raw = dif_fms(L, R, max_disp, 0, 1).to(dtype)
        cost = stereonet_aggregator(raw, q, "", num=num)[0]
        ds = disp_sample_values(max_disp, 0, 1).to(dtype).view(1, -1, 1, 1)
        disp = torch.sum(F.softmax(cost, dim=1) * ds, dim=1, keepdim=True)
    loss = disp_smooth_l1_loss(disp, gt.to(dtype), max_disp)
    names = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    return loss.detach(), dict(zip(names, grads)), {k: v for k, v in q.items() if "running_" in k}


def random_params_psm(seed=0, in_planes=64, classif_gain=10.0, bias=False, acf=False):
    """Seeded default-init parameters with the reference's state_dict names (what nn.Conv3d/BatchNorm3d
    default init produces, drawn with an explicit generator), classifier output convs scaled so that costs
    are peaked (SURVEY 8-c fixture recipe).  BN statistics are randomised away from (0, 1) so that a wrong
    fold shows up."""
    g = torch.Generator().manual_seed(seed)
    p = {}

    def conv(name, co, ci, k=3, with_bias=False, transposed=False):
        fan_in = (co if transposed else ci) * k ** 3
        bound = 1.0 / math.sqrt(fan_in)
        shape = (ci, co, k, k, k) if transposed else (co, ci, k, k, k)
        p[name + ".weight"] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        if with_bias:
            p[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound

    def bn(name, c):
        p[name + ".weight"] = 0.5 + torch.rand(c, generator=g)
        p[name + ".bias"] = (torch.rand(c, generator=g) - 0.5) * 0.2
        p[name + ".running_mean"] = (torch.rand(c, generator=g) - 0.5) * 0.2
        p[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)

    def unit(name, co, ci, with_bias=False, transposed=False):
        conv(name + ".0", co, ci, with_bias=with_bias, transposed=transposed)
        bn(name + ".1", co)

    wb = bias or acf
    unit("dres0.0", 32, in_planes, wb)
    unit("dres0.1", 32, 32, wb)
    unit("dres1.0", 32, 32, wb)
    unit("dres1.1", 32, 32, wb)
    for hg in ("dres2", "dres3", "dres4"):
        unit(hg + ".conv1", 64, 32)
        unit(hg + ".conv2", 64, 64)
        unit(hg + ".conv3", 64, 64)
        unit(hg + ".conv4", 64, 64)
        unit(hg + ".conv5", 64, 64, transposed=True)
        unit(hg + ".conv6", 32, 64, transposed=True)
    for i in (1, 2, 3):
        unit("classif%d.0" % i, 32, 32, wb)
        conv("classif%d.1" % i, 1, 32)
        p["classif%d.1.weight" % i] *= classif_gain
        if acf:
            fan_in = 1 * 8 ** 3
            p["deconv%d.weight" % i] = (torch.rand((1, 1, 8, 8, 8), generator=g) * 2 - 1) / math.sqrt(fan_in) * 8.0
    return p


def with_prefix(p, prefix):
    return {prefix + k: v for k, v in p.items()}
