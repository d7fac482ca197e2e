from .misc import ensure_tuple, ensure_tuple_rep, fall_back_tuple, look_up_option  # noqa: F401
