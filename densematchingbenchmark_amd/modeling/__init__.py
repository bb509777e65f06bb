"""Host-side mirror of ``dmb.modeling`` for the cost-volume -> aggregation -> regression path."""
from .stereo import build_stereo_model


def build_model(cfg, backbone="auto"):
    """dmb/modeling/__init__.py:10 -- only the stereo GeneralizedStereoModel meta-architecture is on the path.  As in the
    reference the model comes WITH the backbone ``cfg.model.backbone`` names (a reference checkpoint loads ``strict=True``);
    ``backbone=None`` builds the cost path alone, fed through ``batch['leftFeature'] / ['rightFeature']``."""
    task = cfg.get("task", "stereo")
    if task != "stereo":
        raise NotImplementedError("task '%s' is outside the HIP path (the reference has no flow model either)" % task)
    return build_stereo_model(cfg, backbone=backbone)
