"""Registry of the cost aggregators on the HIP path (keys as in the reference's aggregators/builder.py:8-15)."""
from ....registry import instantiate
from .AcfNet import AcfAggregator
from .GCNet import GCAggregator
from .PSMNet import PSMAggregator
from .StereoNet import StereoNetAggregator

AGGREGATORS = dict(
    PSMNet=PSMAggregator,
    GwcNet=PSMAggregator,   # the gwc + concat volume has 64 channels = PSMAggregator(in_planes=64)
    AcfNet=AcfAggregator,
    StereoNet=StereoNetAggregator,
    GCNet=GCAggregator,
)


def build_cost_aggregator(cfg):
    """``cfg.model.cost_processor.cost_aggregator`` plus the model-wide ``batch_norm`` flag."""
    return instantiate(AGGREGATORS, cfg.model.cost_processor.cost_aggregator, "cost aggregator",
                       off_path=("DeepPruner", "AnyNet"), batch_norm=cfg.model.batch_norm)
