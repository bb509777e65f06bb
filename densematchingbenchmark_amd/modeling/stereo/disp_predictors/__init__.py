from .builder import PREDICTORS, build_disp_predictor  # noqa: F401
