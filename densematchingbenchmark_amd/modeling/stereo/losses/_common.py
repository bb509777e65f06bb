"""Shared host logic of the loss evaluators: ground truth brought to a level's resolution, weights per level."""
import torch
import torch.nn.functional as F

from .... import ops


def scaled_gt(gt_disp, hw, sparse):
    """Ground truth at a cost / disparity level's resolution (stereo_focal_loss.py:66-73): divided by the width ratio and
    pooled (max for sparse ground truth, average otherwise).  Returns (gt, scale)."""
    H, W = hw
    if gt_disp.shape[-2] == H and gt_disp.shape[-1] == W:
        return gt_disp, 1.0
    scale = gt_disp.shape[-1] / (W * 1.0)
    pool = F.adaptive_max_pool2d if sparse else F.adaptive_avg_pool2d
    return pool(gt_disp / scale, (H, W)), scale


def per_level(values, n):
    """A scalar / None / list of per-level weights -> list of n floats (None -> 1.0)."""
    if values is None:
        values = 1.0
    if not isinstance(values, (list, tuple)):
        values = [values] * n
    return list(values)


class MapLoss(torch.autograd.Function):
    """Masked mean over a [B, 1, H, W] map: mode 0 = -logsigmoid(x), mode 1 = smooth_l1(x - gt)."""

    @staticmethod
    def forward(ctx, x, gt, lower, upper, mode):
        out = ops.map_loss_fwd(x.detach(), gt, lower, upper, mode)
        ctx.save_for_backward(x.detach(), gt, out)
        ctx.args = (lower, upper, mode)
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        x, gt, out = ctx.saved_tensors
        lower, upper, mode = ctx.args
        return ops.map_loss_bwd(x, gt, out, grad_out.float().contiguous(), lower, upper, mode), None, None, None, None
