from . import fmha  # noqa: F401
