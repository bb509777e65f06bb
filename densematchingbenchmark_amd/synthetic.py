"""Seeded synthetic weights and inputs for benchmarks and smoke tests (no datasets / checkpoints offline).

Weights: PyTorch default initialisation drawn from an explicit generator, BatchNorm statistics moved away from
(0, 1), and the three classifier output convolutions scaled by ``classif_gain`` so that the regressed cost
volumes are peaked instead of the degenerate near-constant volume default init produces.
Inputs: feature maps ``randn`` seeded by ``1234 + global_pair_index`` and a smooth synthetic
ground-truth disparity field in (1, 150) px."""
import math

import torch


def init_params_(module, seed=0, classif_gain=10.0):
    """In-place, deterministic re-initialisation of every parameter/buffer of a cost-processor style module."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        shape = tuple(v.shape)
        if k.endswith("disp_regression.weight"):
            continue  # frozen disparity samples of FasterSoftArgmin
        if k.endswith("running_var"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif k.endswith("running_mean"):
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        elif v.dim() == 1 and k.endswith(".1.weight"):      # BatchNorm gamma inside a (conv, bn[, relu]) unit
            t = 0.5 + torch.rand(shape, generator=g)
        elif v.dim() == 1:                                   # conv bias / BatchNorm beta
            t = (torch.rand(shape, generator=g) - 0.5) * 0.2
        else:                                                # conv weights: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            fan_in = v[0].numel() if "deconv" not in k and not _is_transposed(module, k) else v[:, 0].numel()
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(max(fan_in, 1))
            if any(k.endswith("classif%d.1.weight" % i) for i in (1, 2, 3)) or k.endswith(("lastconv.weight", "layer37.weight")):
                t = t * classif_gain
            if "deconv" in k:                                # AcfNet's learned 4x up-sampling: keep O(1) gain
                t = t * 8.0
        new[k] = t.to(v.dtype)
    module.load_state_dict(new, strict=False)
    return module


def _is_transposed(module, key):
    name = key.rsplit(".", 1)[0]
    try:
        sub = module.get_submodule(name)
    except AttributeError:
        return False
    return isinstance(sub, (torch.nn.ConvTranspose3d, torch.nn.ConvTranspose2d))


def feature_pair(global_pair_index, channels, height, width, device="cpu"):
    """Left/right feature maps [1, C, H, W] for one stereo pair."""
    g = torch.Generator().manual_seed(1234 + int(global_pair_index))
    left = torch.randn((1, channels, height, width), generator=g)
    right = torch.randn((1, channels, height, width), generator=g)
    return left.to(device), right.to(device)


def feature_batch(first_pair_index, stride, batch, channels, height, width, device="cpu"):
    """Batch of pairs ``first, first+stride, ...`` -- the reference shards pair i to rank i mod world
    (tools/test.py:108), so a rank's local batch is strided by the world size."""
    pairs = [feature_pair(first_pair_index + j * stride, channels, height, width) for j in range(batch)]
    left = torch.cat([p[0] for p in pairs]).to(device)
    right = torch.cat([p[1] for p in pairs]).to(device)
    return left, right


def gt_disparity(global_pair_index, batch, height, width, pad_top=0, device="cpu"):
    """Smooth ground-truth field in (1, 150) px, zero in the padded rows (masked out by lower_bound=0)."""
    g = torch.Generator().manual_seed(4321 + int(global_pair_index))
    ph = torch.rand((batch, 4), generator=g) * 6.283
    ys = torch.linspace(0, 1, height).view(1, height, 1)
    xs = torch.linspace(0, 1, width).view(1, 1, width)
    f = (torch.sin(2.5 * ys + ph[:, 0].view(-1, 1, 1)) * torch.cos(3.5 * xs + ph[:, 1].view(-1, 1, 1))
         + torch.sin(1.5 * (xs + ys) + ph[:, 2].view(-1, 1, 1)))
    d = 75.5 + 37.0 * f
    d = d.clamp(1.5, 149.5)
    if pad_top > 0:
        d[:, :pad_top, :] = 0.0
    return d.unsqueeze(1).contiguous().to(device)


def gt_batch(first_pair_index, stride, batch, height, width, pad_top=0, device="cpu"):
    """Ground truth of pairs ``first, first+stride, ...`` -- a function of the GLOBAL pair index only, so that a job's
    accumulated errors do not depend on how its pairs are sharded over ranks."""
    return torch.cat([gt_disparity(first_pair_index + j * stride, 1, height, width, pad_top) for j in range(batch)]).to(device)


def image_pair(global_pair_index, height, width):
    """SURVEY.md 8-d's synthetic IMAGE pair, as the decoder would hand it over: uint8 [H, W, 3] left / right views.  Left =
    low-pass-filtered uniform texture in [0, 255); right = the left view resampled along x by the pair's smooth ground-truth
    field (``gt_disparity``), R(x) = L(x + d(x)) with linear interpolation -- so the evaluation mask and the two views belong
    together.  Host-side input generation (torch CPU), not part of the path."""
    g = torch.Generator().manual_seed(2468 + int(global_pair_index))
    tex = torch.rand((1, 3, height // 4 + 2, width // 4 + 2), generator=g)
    fine = torch.rand((1, 3, height, width), generator=g)
    left = torch.nn.functional.interpolate(tex, (height, width), mode="bicubic", align_corners=True).clamp(0, 1) * 0.75 + fine * 0.25
    d = gt_disparity(global_pair_index, 1, height, width)[0, 0]                      # [H, W]
    xs = torch.arange(width, dtype=torch.float32).view(1, width) + d
    x0 = xs.floor().clamp(0, width - 1)
    x1 = (x0 + 1).clamp(max=width - 1)
    lam = (xs.clamp(0, width - 1) - x0).clamp(0, 1)
    rows = left[0]                                                                   # [3, H, W]
    gather = lambda ix: torch.gather(rows, 2, ix.long().unsqueeze(0).expand(3, -1, -1))   # noqa: E731
    right = gather(x0) * (1 - lam) + gather(x1) * lam
    to_u8 = lambda t: (t.clamp(0, 1) * 255.0).floor().clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous()   # noqa: E731
    return to_u8(left[0]), to_u8(right)


def image_batch(first_pair_index, stride, batch, height, width, device="cpu"):
    """uint8 [B, H, W, 3] x 2 of pairs ``first, first + stride, ...`` (the decoder's layout: what ops.stereo_pad_normalize takes)."""
    pairs = [image_pair(first_pair_index + j * stride, height, width) for j in range(batch)]
    return torch.stack([p[0] for p in pairs]).to(device), torch.stack([p[1] for p in pairs]).to(device)


def peaked_cost_volume(seed, planes, height, width):
    """A quarter-resolution cost volume [1, planes, height, width] with ground-truth-like peaks: the peak plane
    follows a smooth field that sits near plane 1.25 (full-resolution disparity 5) on the left quarter of the image, near plane
    46.25 (disparity 185) on the right quarter and ramps in between; costs fall from +12 at the peak to -12 four planes away,
    plus 0.3 sigma of noise -- the ends of the disparity range and strongly peaked volumes, which the random-weight networks
    never produce."""
    g = torch.Generator().manual_seed(9000 + int(seed))
    ys = torch.linspace(0, 1, height).view(1, height, 1)
    xs = torch.linspace(0, 1, width).view(1, 1, width)
    t = ((xs - 0.25) / 0.5).clamp(0, 1)
    ramp = t * t * (3 - 2 * t)
    peak = 1.25 + 45.0 * ramp + 0.6 * torch.sin(9.0 * ys) * torch.sin(5.0 * xs)      # in planes
    peak = peak.clamp(0.25, planes - 1.25) * (planes - 1) / 47.0 if planes != 48 else peak.clamp(0.25, planes - 1.25)
    z = torch.arange(planes, dtype=torch.float32).view(planes, 1, 1)
    cost = (12.0 - 6.0 * (z - peak).abs()).clamp(min=-12.0) + 0.3 * torch.randn((planes, height, width), generator=g)
    return cost.unsqueeze(0).contiguous()


def banded_match_pair(seed, h, w, planes, bands=4):
    """Feature maps with EXACT matches: left ~ N(0, 1) per quarter-resolution pixel, right = the left one shifted by an integer
    disparity that is constant in each of ``bands`` horizontal bands (R(y, x') = L(y, x' + d(y)); noise where nothing matches).
    Returns (left, right [1, 32, h, w], ground truth [1, 1, 4 h, 4 w] in full-resolution pixels, 0 = no match).  The training data
    of oracle/train_peaked_reference.py and the input of the peaked full-size fixture (oracle/gen_golden_fullsize.py `peaked`)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(31000 + int(seed))
    left = torch.randn((1, 32, h, w), generator=g)
    right = torch.randn((1, 32, h, w), generator=g)          # noise where nothing matches
    gt = torch.zeros((1, 1, h, w))
    edges = [0] + sorted(torch.randint(1, h, (bands - 1,), generator=g).tolist()) + [h]
    for b in range(bands):
        y0, y1 = edges[b], edges[b + 1]
        if y1 <= y0:
            continue
        d = int(torch.randint(0, min(planes, w - 8), (1,), generator=g))
        right[:, :, y0:y1, :w - d] = left[:, :, y0:y1, d:]
        gt[:, :, y0:y1, d:] = float(d) if d > 0 else 0.25       # (disparity 0 is a legal match: keep it inside the mask)
    gt_full = F.interpolate(gt, scale_factor=4, mode="nearest") * 4.0
    return left, right, gt_full
