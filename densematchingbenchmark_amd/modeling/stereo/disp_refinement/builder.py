"""dmb/modeling/stereo/disp_refinement/builder.py:5-25 (StereoNet only: DeepPruner / AnyNet are out of scope)."""
from .StereoNet import StereoNetRefinement

REFINEMENTS = {"StereoNet": StereoNetRefinement}


def build_disp_refinement(cfg):
    refine_type = cfg.model.disp_refinement.type
    if refine_type not in REFINEMENTS:
        raise NotImplementedError("disp refinement type not found, expected: {}, but got {}".format(
            list(REFINEMENTS.keys()), refine_type))
    args = dict(cfg.model.disp_refinement)
    args.pop('type')
    args.update(batch_norm=cfg.model.batch_norm)
    return REFINEMENTS[refine_type](**args)
