from . import nets  # noqa: F401
