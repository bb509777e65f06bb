from .basic_layers import FusedConv3d, HeadConv3d, conv3d_bn, conv3d_bn_relu, deconv3d_bn  # noqa: F401
