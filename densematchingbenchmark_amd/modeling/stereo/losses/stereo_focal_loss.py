"""StereoFocalLoss with the Laplace target distribution: drop-in for dmb/modeling/stereo/losses/stereo_focal_loss.py:9-140
and the LaplaceDisp2Prob it calls (losses/utils/disp2prob.py:107-173), one forward and one backward kernel per cost level."""
import torch

from .... import ops
from ._common import per_level, scaled_gt


class _FocalLevel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, variance, gt, values, lower, upper, start, end, coefficient):
        var = variance.detach() if torch.is_tensor(variance) else variance
        out, stats = ops.stereo_focal_loss_fwd(cost.detach(), gt, var, values, lower, upper, start, end, coefficient)
        ctx.save_for_backward(cost.detach(), gt, out, stats, *([var] if torch.is_tensor(var) else []))
        ctx.args = (None if torch.is_tensor(var) else var, values, lower, upper, start, end, coefficient)
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        saved = ctx.saved_tensors
        cost, gt, out, stats = saved[:4]
        var_scalar, values, lower, upper, start, end, coefficient = ctx.args
        var = saved[4] if len(saved) > 4 else var_scalar
        want_var = torch.is_tensor(var) and ctx.needs_input_grad[1]
        gcost, gvar = ops.stereo_focal_loss_bwd(cost, gt, var, values, stats, out, grad_out.float().contiguous(), lower,
                                                upper, start, end, coefficient, want_var)
        return gcost, gvar, None, None, None, None, None, None, None


class StereoFocalLoss(object):
    """Same constructor, call signature and returned dict (``stereo_focal_loss_lvl{i}``) as the reference."""

    def __init__(self, max_disp, start_disp=0, dilation=1, weights=None, focal_coefficient=0.0, sparse=False):
        self.max_disp, self.start_disp, self.dilation = max_disp, start_disp, dilation
        self.end_disp = self.max_disp + self.start_disp - 1
        self.weights, self.focal_coefficient, self.sparse = weights, focal_coefficient, sparse

    def loss_per_level(self, estCost, gtDisp, variance, dilation, disp_sample):
        if disp_sample is not None:
            raise NotImplementedError("per-pixel disp_sample (DeepPruner) is outside the HIP path")
        B, C, H, W = estCost.shape
        gt, scale = scaled_gt(gtDisp, (H, W), self.sparse)
        max_disp = int(self.max_disp / scale)                              # stereo_focal_loss.py:79,89
        lower, upper = self.start_disp, self.start_disp + max_disp
        values = ops.disp_sample_values(max_disp, self.start_disp, dilation)
        if len(values) != C:
            raise ValueError("cost volume has %d disparity samples, the loss expects %d" % (C, len(values)))
        var = variance
        if torch.is_tensor(var) and var.numel() == 1:
            if var.requires_grad and torch.is_grad_enabled():
                raise NotImplementedError("StereoFocalLoss: a learnable scalar variance has no backward on the HIP path "
                                          "(pass a [B, 1, H, W] map, as the confidence network does, or a float)")
            var = float(var)   # (one host read; a Python float costs none)
        elif torch.is_tensor(var):
            var = var.expand(B, 1, H, W)
        return _FocalLevel.apply(estCost, var, gt.detach().contiguous(), values, lower, upper, self.start_disp,
                                 self.start_disp + max_disp - 1, self.focal_coefficient)

    def __call__(self, estCost, gtDisp, variance, disp_sample=None):
        if not isinstance(estCost, (list, tuple)):
            estCost = [estCost]
        n = len(estCost)
        weights, dilations = per_level(self.weights, n), per_level(self.dilation, n)
        variances = list(variance) if isinstance(variance, (list, tuple)) else [variance] * n
        samples = list(disp_sample) if isinstance(disp_sample, (list, tuple)) else [disp_sample] * n
        return {"stereo_focal_loss_lvl{}".format(i): weights[i] * self.loss_per_level(c, gtDisp, v, dl, ds)
                for i, (c, v, dl, ds) in enumerate(zip(estCost, variances, dilations, samples))}

    @property
    def name(self):
        return 'StereoFocalLoss'
