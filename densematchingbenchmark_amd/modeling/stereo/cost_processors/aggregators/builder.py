"""cost_processors/aggregators/builder.py:8-29 for the aggregators on the HIP path."""
from .AcfNet import AcfAggregator
from .GCNet import GCAggregator
from .PSMNet import PSMAggregator
from .StereoNet import StereoNetAggregator

AGGREGATORS = {
    "PSMNet": PSMAggregator,
    "GwcNet": PSMAggregator,  # the gwc+concat volume has 64 channels = PSMAggregator(in_planes=64)
    "AcfNet": AcfAggregator,
    "StereoNet": StereoNetAggregator,
    "GCNet": GCAggregator,
}
_OFF_PATH = ("DeepPruner", "AnyNet")


def build_cost_aggregator(cfg):
    agg_type = cfg.model.cost_processor.cost_aggregator.type
    if agg_type in _OFF_PATH:
        raise NotImplementedError("cost_aggregator '%s' is outside the HIP hot path (SURVEY.md section 2.1 row 6)" % agg_type)
    assert agg_type in AGGREGATORS, "cost_aggregator type not found, excepted: {}," \
                                    "but got {}".format(AGGREGATORS.keys(), agg_type)
    default_args = cfg.model.cost_processor.cost_aggregator.copy()
    default_args.pop('type')
    default_args.update(batch_norm=cfg.model.batch_norm)
    return AGGREGATORS[agg_type](**default_args)
