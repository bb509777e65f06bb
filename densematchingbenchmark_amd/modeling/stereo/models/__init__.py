from .general_stereo_model import GeneralizedStereoModel

_META_ARCHITECTURES = {"GeneralizedStereoModel": GeneralizedStereoModel}


def build_stereo_model(cfg, backbone="auto"):
    name = cfg.model.meta_architecture
    if name not in _META_ARCHITECTURES:
        raise NotImplementedError("meta architecture '%s' is outside the HIP hot path" % name)
    return _META_ARCHITECTURES[name](cfg, backbone=backbone)
