"""The caller of the hot path: models/general_stereo_model.py:14-92 (eval-mode contract; training mode = the same cost path
under autograd on the HIP kernels + the configured disparity losses).

``build_model(cfg)`` means what it means in the reference (dmb/modeling/__init__.py:10, general_stereo_model.py:24): the
backbone ``cfg.model.backbone`` names is built with the model (this package's HIP backbone), so a reference
checkpoint loads with ``strict=True``.  The hot path itself starts at the feature maps: a model WITH a backbone still takes
pre-computed features through ``batch['leftFeature'] / batch['rightFeature']``; ``backbone=None`` builds the path alone (the
benchmarks' and kernel tests' configuration) and any ``nn.Module`` with the reference's ``backbone(left, right) -> (ref_fms,
tgt_fms)`` contract may be passed instead.  A ``disp_refinement`` entry in the config attaches the HIP refinement stage
 exactly where the reference runs it."""
import torch
import torch.nn as nn

from ..cmn import build_cmn
from ..cost_processors import build_cost_processor
from ..disp_predictors import build_disp_predictor
from ..layers import train_fn


class GeneralizedStereoModel(nn.Module):
    def __init__(self, cfg, backbone="auto"):
        super().__init__()
        self.cfg = cfg.copy()
        self.max_disp = cfg.model.max_disp
        if isinstance(backbone, str):
            # "auto" (default) = the reference's behaviour: build what cfg.model.backbone names (general_stereo_model.py:24);
            # a config without that entry builds the path alone.  "hip" insists on the entry.
            if backbone not in ("auto", "hip"):
                raise ValueError("backbone must be 'auto', 'hip', None or an nn.Module, got %r" % (backbone,))
            if backbone == "hip" or 'backbone' in cfg.model:
                from ..backbones import build_backbone
                backbone = build_backbone(cfg)
            else:
                backbone = None
        self.backbone = backbone
        self.cost_processor = build_cost_processor(cfg)
        self.cmn = build_cmn(cfg) if 'cmn' in cfg.model else None
        self.disp_predictor = build_disp_predictor(cfg)
        self.disp_refinement = None
        self.loss_evaluator = None
        if 'disp_refinement' in cfg.model:                           # general_stereo_model.py:35-37
            from ..disp_refinement import build_disp_refinement
            self.disp_refinement = build_disp_refinement(cfg)

    def _forward_train(self, batch, ref_fms, tgt_fms):
        """general_stereo_model.py:60-77: the same forward under autograd, then the configured losses.  Built for the
        cost path (volume builder, aggregator, regression, confidence network), the refinement stage and their losses
        (SURVEY 8-f3)."""
        target = batch.get('leftDisp')
        costs = self.cost_processor(ref_fms, tgt_fms)
        disps = [self.disp_predictor(cost) for cost in costs]
        if self.disp_refinement is not None:                         # general_stereo_model.py:57-58
            if 'leftImage' not in batch:
                raise ValueError("disp_refinement needs batch['leftImage'] (full-resolution left view)")
            disps = self.disp_refinement(disps, ref_fms, tgt_fms, batch['leftImage'], batch.get('rightImage'))
        if self.loss_evaluator is None:
            if 'losses' not in self.cfg.model:
                raise ValueError("training mode needs cfg.model.losses (general_stereo_model.py:40)")
            from ..losses import make_gsm_loss_evaluator
            self.loss_evaluator = make_gsm_loss_evaluator(self.cfg)
        loss_dict = dict()
        variance = None
        if hasattr(self.cfg.model.losses, 'focal_loss'):
            variance = self.cfg.model.losses.focal_loss.get('variance', None)
        if self.cmn is not None:                                     # general_stereo_model.py:66-69
            variance, cm_losses = self.cmn(costs, target)
            loss_dict.update(cm_losses)
        loss_dict.update(self.loss_evaluator(disps, costs, target, variance=variance))
        return {}, loss_dict

    def forward(self, batch):
        if self.training:
            # one gradient-carry scope per forward pass (layers/train_fn.py: the sums autograd would form for tensors with several
            # consumers are taken over by the consumers' own backward kernels)
            with train_fn.carry_scope():
                return self._forward(batch)
        return self._forward(batch)

    def _forward(self, batch):
        if 'leftFeature' in batch:
            ref_fms, tgt_fms = batch['leftFeature'], batch['rightFeature']
        else:
            if self.backbone is None:
                raise ValueError("no backbone attached: provide batch['leftFeature'] and batch['rightFeature']")
            ref_fms, tgt_fms = self.backbone(batch['leftImage'], batch['rightImage'])
        if self.training:
            return self._forward_train(batch, ref_fms, tgt_fms)
        with torch.no_grad():
            costs = self.cost_processor(ref_fms, tgt_fms)            # general_stereo_model.py:51
            disps = [self.disp_predictor(cost) for cost in costs]    # :54
            if self.disp_refinement is not None:                     # :57-58
                if 'leftImage' not in batch:
                    raise ValueError("disp_refinement needs batch['leftImage'] (full-resolution left view)")
                disps = self.disp_refinement(disps, ref_fms, tgt_fms, batch['leftImage'], batch.get('rightImage'))
            results = dict(disps=disps, costs=costs)                 # :82-85
            if self.cmn is not None:
                variance, confs = self.cmn(costs, batch.get('leftDisp'))  # :87-90
                results.update(confs=confs)
        return results, {}
