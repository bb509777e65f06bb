"""``COR_FUNCS``: the reference's own entry (``default`` = correlation1d_cost, cost_processors/utils/
correlation1d_cost.py:7-31) and the group-wise correlation volumes of GwcNet, which the reference does not implement
(README.md:16 names the model only; spec: SURVEY.md section 8-a4) under their own keys ``gwc`` / ``gwc_cat``."""
import torch

from ..... import ops


def correlation1d_cost(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None,
                       kernel_size=1, stride=1, padding=0, dilation_patch=1):
    """correlation1d_cost.py:7-27: full-channel correlation over the first ``max_disp`` of the sampler's
    ``2*max_disp - 1`` patch offsets (channel j = disparity max_disp - 1 - j), no normalisation, then
    ``leaky_relu(0.1)``; output [B, max_disp, H, W].  ``start_disp`` / ``dilation`` / ``disp_sample`` are accepted and
    ignored exactly as there.  The sampler package is not part of the reference tree: parity UNPINNED (include/dmb_hip.h)."""
    if (kernel_size, stride, padding, dilation_patch) != (1, 1, 0, 1):
        raise NotImplementedError("correlation1d_cost on the HIP path: kernel_size=1, stride=1, padding=0, dilation_patch=1 only")
    if torch.is_grad_enabled() and (reference_fm.requires_grad or target_fm.requires_grad):
        raise NotImplementedError("correlation1d_cost has no backward on the HIP path")
    return ops.correlation1d(reference_fm.float(), target_fm.float(), max_disp, 0.1)


def gwc_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None, num_groups=40):
    """[B, C, H, W] x 2 -> [B, G, D, H, W]: per-group mean of L[c, y, x] * R[c, y, x - d]."""
    idx = ops.disp_index_list(max_disp, start_disp, dilation)
    return ops.gwc_fms(reference_fm.float(), target_fm.float(), idx, num_groups)


def gwc_cat_fms(reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, disp_sample=None, num_groups=40):
    """GwcNet "gwc + concat" volume.  ``reference_fm``/``target_fm`` are (correlation_features,
    concat_features) pairs: [B, 320, H, W] and [B, 12, H, W]; output [B, G + 2*Cc, D, H, W] -- 64 channels for
    G=40, Cc=12, i.e. what PSMAggregator(in_planes=64) consumes.  Both parts are written in place into one
    tensor (no torch.cat pass)."""
    (lg, lc), (rg, rc) = reference_fm, target_fm
    idx = ops.disp_index_list(max_disp, start_disp, dilation)
    B, _, H, W = lg.shape
    Cc = lc.shape[1]
    out = torch.empty((B, num_groups + 2 * Cc, len(idx), H, W), dtype=torch.float32, device=lg.device)
    ops.gwc_fms(lg.float(), rg.float(), idx, num_groups, out=out, out_ch_offset=0)
    ops.cat_fms_into(lc.float(), rc.float(), idx, out, num_groups)
    return out


class LazyGwcCatVolume:
    """The "gwc + concat" volume as a DESCRIPTION (eval mode, built by the cost processor): its concat channels have the
    structure csrc/catconv.hip exploits (2-D maps instead of a 3-D convolution over them), its correlation channels do not
    and are materialised on their own.  ``materialize()`` gives exactly gwc_cat_fms's tensor."""

    kind = "gwc_cat"

    def __init__(self, reference_fm, target_fm, max_disp=192, start_disp=0, dilation=1, num_groups=40, **unused):
        (self.lg, self.lc), (self.rg, self.rc) = reference_fm, target_fm
        self.args = dict(max_disp=max_disp, start_disp=start_disp, dilation=dilation, num_groups=num_groups)
        self.num_groups = num_groups
        self.reference_fm, self.target_fm = self.lc.float(), self.rc.float()       # the concat part
        self.disp_idx = ops.disp_index_list(max_disp, start_disp, dilation)
        B, _, H, W = self.lg.shape
        self.shape = torch.Size((B, num_groups + 2 * self.lc.shape[1], len(self.disp_idx), H, W))
        self.device, self.dtype, self.requires_grad = self.lg.device, torch.float32, False

    def dim(self):
        return 5

    def correlation_part(self):
        return ops.gwc_fms(self.lg.float(), self.rg.float(), self.disp_idx, self.num_groups)

    def materialize(self):
        return gwc_cat_fms((self.lg, self.lc), (self.rg, self.rc), **self.args)


COR_FUNCS = dict(
    default=correlation1d_cost,
    gwc=gwc_fms,
    gwc_cat=gwc_cat_fms,
)
