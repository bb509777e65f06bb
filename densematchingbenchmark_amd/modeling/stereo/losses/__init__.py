"""Training-side loss terms of AcfNet's cost filtering on HIP kernels (first part): forward and backward
of the three losses, usable under autograd (``loss.backward()`` yields d loss / d cost, d loss / d variance, d loss /
d disparity, d loss / d confidence logits).  The convolutions' backward passes are not on the HIP path."""
from .builder import CombinedLossEvaluators, make_gsm_loss_evaluator
from .conf_nll_loss import ConfidenceNllLoss
from .smooth_l1_loss import DispSmoothL1Loss
from .stereo_focal_loss import StereoFocalLoss

__all__ = ["StereoFocalLoss", "DispSmoothL1Loss", "ConfidenceNllLoss", "CombinedLossEvaluators", "make_gsm_loss_evaluator"]
