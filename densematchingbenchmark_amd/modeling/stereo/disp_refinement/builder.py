"""Registry of the disparity refinements on the HIP path (the reference's disp_refinement/builder.py:5-9 also lists
DeepPruner and AnyNet, which are out of scope here)."""
from ...registry import instantiate
from .StereoNet import StereoNetRefinement

REFINEMENTS = dict(StereoNet=StereoNetRefinement)


def build_disp_refinement(cfg):
    return instantiate(REFINEMENTS, cfg.model.disp_refinement, "disparity refinement", off_path=("DeepPruner", "AnyNet"),
                       batch_norm=cfg.model.batch_norm)
