from .builder import PREDICTORS, build_disp_predictor  # noqa: F401
from .faster_soft_argmin import FasterSoftArgmin  # noqa: F401
from .local_soft_argmin import LocalSoftArgmin  # noqa: F401
from .soft_argmin import SoftArgmin  # noqa: F401
