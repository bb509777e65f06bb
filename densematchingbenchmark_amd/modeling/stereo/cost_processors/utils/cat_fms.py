"""Concatenation cost volume: drop-in for dmb/modeling/stereo/cost_processors/utils/cat_fms.py (``CAT_FUNCS``)."""
import torch

from ..... import ops
from ...layers import train_fn


def cat_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None):
    """[B, C, H, W] x 2 -> [B, 2C, D, H, W]; same arguments and semantics as cat_fms.py:7-48 (``disp_sample`` is
    ignored there too).  One HIP kernel launch instead of 2*D strided slice copies; output is FP32 (cat_fms.py:32)."""
    idx = ops.disp_index_list(max_disp, start_disp, dilation)
    if torch.is_grad_enabled() and (reference_fm.requires_grad or target_fm.requires_grad):
        return train_fn.CatFmsFn.apply(reference_fm.float().contiguous(), target_fm.float().contiguous(), tuple(idx))
    return ops.cat_fms(reference_fm.float(), target_fm.float(), idx)


class LazyCatVolume:
    """The volume of cat_fms(reference_fm, target_fm, ...) -- or, ``kind="dif"``, of dif_fms -- as a DESCRIPTION: an
    aggregator whose first convolution knows the volume's structure (FusedConv3d on csrc/catconv.hip) consumes it without
    the tensor (1.6 GB at the BASELINE size) ever being written; anything else calls ``materialize()`` and gets exactly the
    builder's tensor.  Built by the cost processor (cost_processors/builder.py).  ``differentiable`` (round 6: the training
    path): the first convolution runs its FORWARD on the 2-D form as well and builds the volume only inside its backward pass
    (train_fn.CatConvUnitFn); ``materialize()`` then goes through the builders' autograd Functions."""

    def __init__(self, reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, kind="cat", differentiable=False, **unused):
        self.reference_fm, self.target_fm, self.kind = reference_fm.float(), target_fm.float(), kind
        self.disp_idx = ops.disp_index_list(max_disp, start_disp, dilation)
        B, C, H, W = reference_fm.shape
        self.shape = torch.Size((B, 2 * C if kind == "cat" else C, len(self.disp_idx), H, W))
        self.device, self.dtype = reference_fm.device, torch.float32
        self.differentiable = bool(differentiable)
        self.requires_grad = self.differentiable and torch.is_grad_enabled() and (reference_fm.requires_grad or target_fm.requires_grad)

    def dim(self):
        return 5

    def materialize(self):
        if self.requires_grad:
            fn = train_fn.CatFmsFn if self.kind == "cat" else train_fn.DifFmsFn
            return fn.apply(self.reference_fm.contiguous(), self.target_fm.contiguous(), tuple(self.disp_idx))
        build = ops.cat_fms if self.kind == "cat" else ops.dif_fms
        return build(self.reference_fm, self.target_fm, self.disp_idx)


def fast_cat_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None):
    """cat_fms.py:51-82: the target features WARPED by ``disp_sample`` ([B, D, H, W], per pixel) -- or by the
    linspace(start, end, D) samples the builder generates itself -- and the reference features masked where the warped
    target is not positive.  Not equivalent to cat_fms: the reference samples a (size - 1)-normalised grid with
    F.grid_sample's align_corners=False default; csrc/warp_volume.hip reproduces that blend bit for bit.
    Under autograd the two feature maps get their gradients from the sampler's adjoint and per-pixel samples that require one
    get theirs from the sampler's column derivative (``ops.fast_fms_bwd``), as the reference's do from F.grid_sample.
    Limit: when a gradient is requested the feature maps may be at most 1024 columns wide (train_fn.FAST_FMS_BWD_MAX_W: the
    adjoint keeps two 8-channel rows in LDS); wider inputs raise NotImplementedError in the forward already."""
    wrt_samples = disp_sample is not None and disp_sample.requires_grad
    if disp_sample is None:
        disp_sample = ops.fast_disp_samples(max_disp, start_disp, dilation)
    if torch.is_grad_enabled() and (reference_fm.requires_grad or target_fm.requires_grad or wrt_samples):
        return train_fn.FastFmsFn.apply(reference_fm.float().contiguous(), target_fm.float().contiguous(),
                                        train_fn.fast_samples_for_autograd(disp_sample, reference_fm), False)
    return ops.fast_cat_fms(reference_fm.float(), target_fm.float(), disp_sample)


CAT_FUNCS = dict(
    default=cat_fms,
    fast_mode=fast_cat_fms,
)
