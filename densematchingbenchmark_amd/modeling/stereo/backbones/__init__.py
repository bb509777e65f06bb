from .GCNet import GCNetBackbone
from .PSMNet import PSMNetBackbone
from .StereoNet import StereoNetBackbone

BACKBONES = {"PSMNet": PSMNetBackbone, "StereoNet": StereoNetBackbone, "GCNet": GCNetBackbone}


def build_backbone(cfg):
    """dmb/modeling/stereo/backbones/builder.py: the PSMNet, StereoNet and GC-Net backbones are on the HIP path."""
    b = cfg.model.backbone
    if b.type not in BACKBONES:
        raise NotImplementedError("backbone '%s' is outside the HIP path (attach a stock PyTorch backbone instead)" % b.type)
    args = dict(b)
    args.pop("type")
    args.update(batch_norm=cfg.model.batch_norm)
    return BACKBONES[b.type](**args)
