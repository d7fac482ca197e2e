from .warp import DVF2DDF, Warp  # noqa: F401
