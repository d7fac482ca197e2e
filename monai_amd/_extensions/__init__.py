from .loader import load_module  # noqa: F401
