from ...registry import instantiate
from .GCNet import GCNetBackbone
from .PSMNet import PSMNetBackbone
from .StereoNet import StereoNetBackbone

BACKBONES = {"PSMNet": PSMNetBackbone, "StereoNet": StereoNetBackbone, "GCNet": GCNetBackbone}


def build_backbone(cfg):
    """``cfg.model.backbone`` plus the model-wide ``batch_norm`` flag (reference backbones/builder.py)."""
    return instantiate(BACKBONES, cfg.model.backbone, "backbone", off_path=("DeepPruner", "AnyNet"),
                       batch_norm=cfg.model.batch_norm)
