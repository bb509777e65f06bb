from .stereo import EpeAccumulator, calc_error, remove_padding  # noqa: F401
