from .builder import REFINEMENTS, build_disp_refinement
from .StereoNet import StereoNetRefinement

__all__ = ["REFINEMENTS", "build_disp_refinement", "StereoNetRefinement"]
