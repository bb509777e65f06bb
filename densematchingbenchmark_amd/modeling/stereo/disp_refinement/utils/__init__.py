from .edge_aware import EdgeAwareRefinement

__all__ = ["EdgeAwareRefinement"]
