"""``GateRecurrent2dnoind``: drop-in for the reference's only native op, dmb/ops/spn (modules/gaterecurrent2dnoind.py:4-13,
functions/gaterecurrent2dnoind.py:8-44), as AnyNet's SPN refinement uses it (disp_refinement/AnyNet.py:54:
``GateRecurrent2dnoind(True, False)``).  One HIP launch per scan instead of one CUDA launch per scanned line; CPU tensors raise
(the reference prints "cpu version is not ready at this time" and returns 0)."""
import torch
import torch.nn as nn


class GateRecurrent2dnoindFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, G1, G2, G3, horizontal, reverse):
        from . import ops
        X, G1, G2, G3 = (t.float().contiguous() for t in (X, G1, G2, G3))
        out = ops.spn_gaterecurrent2d(X, G1, G2, G3, horizontal, reverse)
        ctx.save_for_backward(X, G1, G2, G3, out)
        ctx.horizontal, ctx.reverse = bool(horizontal), bool(reverse)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        from . import ops
        X, G1, G2, G3, out = ctx.saved_tensors
        grads = ops.spn_gaterecurrent2d_bwd(X, G1, G2, G3, out, grad_output.float().contiguous(), ctx.horizontal, ctx.reverse)
        return grads + (None, None)


class GateRecurrent2dnoind(nn.Module):
    """modules/gaterecurrent2dnoind.py:4-13: ``forward(X, G1, G2, G3)`` -> H, all [N, C, H, W]."""

    def __init__(self, horizontal_, reverse_):
        super().__init__()
        self.horizontal = horizontal_
        self.reverse = reverse_

    def forward(self, X, G1, G2, G3):
        return GateRecurrent2dnoindFunction.apply(X, G1, G2, G3, self.horizontal, self.reverse)
