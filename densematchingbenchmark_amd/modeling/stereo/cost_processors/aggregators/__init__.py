from .AcfNet import AcfAggregator  # noqa: F401
from .GCNet import GCAggregator  # noqa: F401
from .PSMNet import PSMAggregator  # noqa: F401
from .StereoNet import StereoNetAggregator  # noqa: F401
from .builder import AGGREGATORS, build_cost_aggregator  # noqa: F401
