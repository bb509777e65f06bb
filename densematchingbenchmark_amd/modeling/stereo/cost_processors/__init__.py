from .builder import PROCESSORS, build_cost_processor  # noqa: F401
