from .cmn import Cmn, ConfHead, build_cmn  # noqa: F401
