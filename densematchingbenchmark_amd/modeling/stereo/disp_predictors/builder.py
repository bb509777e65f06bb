"""Registry of the disparity predictors on the HIP path (keys as in the reference's disp_predictors/builder.py:5-9)."""
from ...registry import instantiate
from .faster_soft_argmin import FasterSoftArgmin
from .local_soft_argmin import LocalSoftArgmin
from .soft_argmin import SoftArgmin

PREDICTORS = dict(DEFAULT=SoftArgmin, FASTER=FasterSoftArgmin, LOCAL=LocalSoftArgmin)


def build_disp_predictor(cfg):
    """``cfg.model.disp_predictor``: ``type`` defaults to 'FASTER' (reference builder.py:13), the rest are kwargs."""
    return instantiate(PREDICTORS, cfg.model.disp_predictor, "disparity predictor", default_type="FASTER")
