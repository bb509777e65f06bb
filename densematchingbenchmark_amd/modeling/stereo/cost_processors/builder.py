"""cost_processors/builder.py:21-107: cost processors = volume builder + aggregator, selected by config strings."""
import torch
import torch.nn as nn

from .... import ops
from ..layers import train_fn
from .aggregators import build_cost_aggregator
from .utils.cat_fms import CAT_FUNCS, LazyCatVolume, cat_fms
from .utils.dif_fms import DIF_FUNCS, dif_fms
from .utils.gwc_fms import COR_FUNCS, LazyGwcCatVolume, gwc_cat_fms


class _VolumeThenAggregate(nn.Module):
    """Shared body of Cat/Dif/CorCostProcessor (builder.py:21-86): pick the builder function from ``FUNCS`` by
    ``cost_computation.type``, keep the remaining keys as its kwargs, build the aggregator."""
    FUNCS = None

    def __init__(self, cfg):
        super().__init__()
        comp = cfg.model.cost_processor.cost_computation
        self.vol_func = self.FUNCS[comp.get('type', 'default')]
        self.default_args = comp.copy()
        self.default_args.pop('type')
        self.aggregator = build_cost_aggregator(cfg)

    def forward(self, ref_fms, tgt_fms, disp_sample=None):
        if ((self.vol_func is cat_fms or self.vol_func is dif_fms) and getattr(self.aggregator, "accepts_lazy_cat", False)
                and ops.cat_fusion() and torch.is_tensor(ref_fms)):
            # the aggregator's first convolution consumes the volume's description (csrc/catconv.hip); the raw volume is not part
            # of the result contract (general_stereo_model.py:82-85) and is never written by the forward pass -- in eval mode and
            # (round 6) on the training path, where the first unit's backward builds it for its weight gradient
            raw_cost = LazyCatVolume(ref_fms, tgt_fms, kind="cat" if self.vol_func is cat_fms else "dif",
                                     differentiable=train_fn.wants_grad(self, ref_fms, tgt_fms), **self.default_args)
        elif (self.vol_func is gwc_cat_fms and getattr(self.aggregator, "accepts_lazy_cat", False) and ops.cat_fusion()
              and not train_fn.wants_grad(self, *ref_fms, *tgt_fms)):
            raw_cost = LazyGwcCatVolume(ref_fms, tgt_fms, **self.default_args)   # concat channels as 2-D maps, correlation channels 3-D
        else:
            raw_cost = self.vol_func(ref_fms, tgt_fms, disp_sample=disp_sample, **self.default_args)
        return self.aggregator(raw_cost)


class CatCostProcessor(_VolumeThenAggregate):
    FUNCS = CAT_FUNCS


class DifCostProcessor(_VolumeThenAggregate):
    FUNCS = DIF_FUNCS


class CorCostProcessor(_VolumeThenAggregate):
    FUNCS = COR_FUNCS


PROCESSORS = {
    'Difference': DifCostProcessor,
    'Concatenation': CatCostProcessor,
    'Correlation': CorCostProcessor,
}
_OFF_PATH = ('DeepPruner', 'AnyNet')


def build_cost_processor(cfg):
    proc_type = cfg.model.cost_processor.type
    if proc_type in _OFF_PATH:
        raise NotImplementedError("cost_processor '%s' is outside the HIP hot path" % proc_type)
    assert proc_type in PROCESSORS, "cost_processor type not found, excepted: {}," \
                                    "but got {}".format(PROCESSORS.keys(), proc_type)
    return PROCESSORS[proc_type](cfg=cfg)
