from .image_io import imread
from .transforms import CenterCrop, Compose, Normalize, StereoPad, ToTensor, build_transforms

__all__ = ["imread", "CenterCrop", "Compose", "Normalize", "StereoPad", "ToTensor", "build_transforms"]
