from .PSMNet import PSMNetBackbone

BACKBONES = {"PSMNet": PSMNetBackbone}


def build_backbone(cfg):
    """dmb/modeling/stereo/backbones/builder.py: only the PSMNet backbone is on the HIP path so far."""
    b = cfg.model.backbone
    if b.type not in BACKBONES:
        raise NotImplementedError("backbone '%s' is outside the HIP path (attach a stock PyTorch backbone instead)" % b.type)
    args = b.copy()
    args.pop("type")
    args.update(batch_norm=cfg.model.batch_norm)
    return BACKBONES[b.type](**args)
