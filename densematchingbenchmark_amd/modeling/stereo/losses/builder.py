"""Loss evaluator assembled from ``cfg.model.losses`` (reference: dmb/modeling/stereo/losses/builder.py:56-113) for the loss
terms that exist on the HIP path: ``l1_loss`` and ``focal_loss`` (plus the confidence loss the AcfNet cmn module owns,
cmn/cmn.py).  ``gerf_loss`` and ``relative_loss`` are not built and raise."""
from .smooth_l1_loss import DispSmoothL1Loss
from .stereo_focal_loss import StereoFocalLoss

_NOT_BUILT = ("gerf_loss", "relative_loss")


def _l1(cfg, node):
    return DispSmoothL1Loss(max_disp=node.get("max_disp", None), weights=node.weights, sparse=cfg.data.sparse)


def _focal(cfg, node):
    return StereoFocalLoss(max_disp=node.get("max_disp", None), start_disp=node.get("start_disp", 0),
                           dilation=node.get("dilation", 1), weights=node.get("weights", None),
                           focal_coefficient=node.get("coefficient", 0.0), sparse=cfg.data.sparse)


_FACTORIES = dict(l1_loss=_l1, focal_loss=_focal)


class CombinedLossEvaluators(object):
    """``evaluator(disps, costs, target, variance=...)`` -> dict of weighted per-level losses, keys as in the reference."""

    def __init__(self, cfg, loss_evaluators, loss_weights):
        self.cfg = cfg.copy()
        self.loss_evaluators, self.loss_weights = loss_evaluators, loss_weights

    def __call__(self, disps, costs, target, **kwargs):
        out = dict()
        for name, evaluator in self.loss_evaluators.items():
            if isinstance(evaluator, StereoFocalLoss):
                terms = evaluator(costs, target, kwargs["variance"])
            else:
                terms = evaluator(disps, target)
            out.update({k: v * self.loss_weights[name] for k, v in terms.items()})
        return out


def make_gsm_loss_evaluator(cfg):
    evaluators, weights = dict(), dict()
    for name, node in cfg.model.losses.items():
        if name in _NOT_BUILT:
            raise NotImplementedError("loss '%s' is outside the HIP path (DESIGN.md, out of scope)" % name)
        if name not in _FACTORIES:
            raise ValueError("{} not implemented.".format(name))
        evaluators[name], weights[name] = _FACTORIES[name](cfg, node), node.weight
    return CombinedLossEvaluators(cfg, evaluators, weights)
