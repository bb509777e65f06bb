from .models import build_stereo_model  # noqa: F401
