"""Fall-through to the reference implementation (boundary B3: "otherwise fall through to the reference / torch
path -- never fail a previously valid call").

The MI355X classes cover the hot path: fp32 tensors resident in HBM, 3-D windows, the common network configurations,
inference.  Everything else the reference accepts -- CPU tensors and numpy arrays, training mode (autograd), 2-D networks, other
norms / activations / up-sampling modes, spline interpolation orders, half / double precision -- is NOT re-implemented here;
when MONAI is importable such a call is handed to the reference's own class / function instead of raising:

  * constructor time: ``OurClass(*args)`` whose configuration the HIP path does not cover (it raises ``NotImplementedError``)
    returns an instance of the reference class built from the same arguments;
  * call time: a call the instance cannot serve (``NotImplementedError``, or ``UnsupportedOnDevice`` from the "must be an fp32
    ROCm tensor" check) goes to a lazily built reference twin.  For networks the twin SHARES the parameters and buffers of the
    MI355X module (same ``Parameter`` objects -- the ``state_dict`` layouts are identical), so ``.train()`` + autograd + an
    optimizer over ``net.parameters()`` work, and ``.eval()`` inference on the GPU is back on the HIP kernels.

The reference object is the one ``monai_amd.patch.install()`` displaced, or -- when the patch is not installed -- whatever the
reference module exports under that name.  Without MONAI the original explicit error is re-raised: there is no other fallback,
and the HIP product path itself never routes through CPU code.
"""

from __future__ import annotations

import functools
import importlib
import warnings

__all__ = ["UnsupportedOnDevice", "reference_object", "reference_fallback", "function_fallback", "fell_through"]


class UnsupportedOnDevice(RuntimeError):
    """The tensor handed to a kernel wrapper is not an fp32 ROCm tensor (``monai_amd._lib.require_device``)."""


_FALLBACK_ERRORS = (NotImplementedError, UnsupportedOnDevice)
_warned: set = set()
_log: list = []          # (component, reason) of every fall-through in this process; tests inspect it


def fell_through() -> list:
    return list(_log)


def reference_object(ref_module: str, name: str):
    """The reference's own ``ref_module.name`` (the displaced object when the patch is installed), or None without MONAI."""
    try:
        from . import patch

        if (ref_module, name) in patch._installed:
            return patch._installed[(ref_module, name)]
        mod = importlib.import_module(ref_module)
    except Exception:
        return None
    obj = getattr(mod, name, None)
    if obj is None or getattr(obj, "_mh_is_product", False):
        return None
    return obj


def _note(component: str, err: BaseException) -> None:
    _log.append((component, str(err)))
    if component not in _warned:
        _warned.add(component)
        warnings.warn(f"monai_amd: {component} falls through to the reference implementation: {err}", stacklevel=3)


def _share_module_state(src, dst) -> None:
    """Make `dst` (reference nn.Module) use the very Parameter / buffer objects of `src` (identical state_dict layout)."""
    import torch

    def owner(root, dotted):
        parts = dotted.split(".")
        m = root
        for p in parts[:-1]:
            m = getattr(m, p)
        return m, parts[-1]

    have = dict(dst.named_parameters())
    for name, p in src.named_parameters():
        if name in have:
            m, leaf = owner(dst, name)
            m._parameters[leaf] = p
    haveb = dict(dst.named_buffers())
    for name, b in src.named_buffers():
        if name in haveb:
            m, leaf = owner(dst, name)
            m._buffers[leaf] = b
    missing = (set(have) - {n for n, _ in src.named_parameters()}) | (set(haveb) - {n for n, _ in src.named_buffers()})
    if missing:
        raise RuntimeError(f"monai_amd: reference twin has state the MI355X module lacks: {sorted(missing)[:5]}")
    assert isinstance(dst, torch.nn.Module)


def reference_fallback(ref_module: str, name: str, methods=("__call__",), share_state: bool = False):
    """Class decorator: see the module docstring.  `methods` are wrapped for call-time fall-through; `share_state` (networks) makes
    the twin share parameters / buffers and follow ``.training``."""

    def deco(cls):
        orig_init = cls.__init__
        cls._mh_is_product = True
        cls._mh_ref = (ref_module, name)

        def __new__(klass, *args, **kwargs):
            self = object.__new__(klass)
            if klass is not cls:                      # subclasses construct normally
                return self
            try:
                orig_init(self, *args, **kwargs)
            except NotImplementedError as e:
                ref = reference_object(ref_module, name)
                if ref is None:
                    raise
                _note(f"{name}(...)", e)
                return ref(*args, **kwargs)
            object.__setattr__(self, "_mh_ctor", (args, kwargs))
            object.__setattr__(self, "_mh_init_done", True)
            return self

        @functools.wraps(orig_init)
        def __init__(self, *args, **kwargs):
            if self.__dict__.get("_mh_init_done"):
                return
            orig_init(self, *args, **kwargs)
            object.__setattr__(self, "_mh_ctor", (args, kwargs))

        def _mh_twin(self):
            twin = self.__dict__.get("_mh_twin_obj")
            if twin is None:
                ref = reference_object(ref_module, name)
                if ref is None:
                    return None
                args, kwargs = self.__dict__.get("_mh_ctor", ((), {}))
                twin = ref(*args, **kwargs)
                if share_state:
                    _share_module_state(self, twin)
                object.__setattr__(self, "_mh_twin_obj", twin)
            if share_state:
                twin.train(self.training)
            else:                                     # transforms: keep the user-visible switches in step
                for attr in ("lazy",):
                    if attr in self.__dict__ or hasattr(type(self), attr):
                        try:
                            setattr(twin, attr, getattr(self, attr))
                        except Exception:
                            pass
            return twin

        def wrap(mname):
            orig = getattr(cls, mname)

            @functools.wraps(orig)
            def method(self, *args, **kwargs):
                try:
                    return orig(self, *args, **kwargs)
                except _FALLBACK_ERRORS as e:
                    twin = self._mh_twin()
                    if twin is None:
                        raise
                    _note(f"{name}.{mname}", e)
                    return getattr(twin, mname)(*args, **kwargs)

            return method

        cls.__new__ = staticmethod(__new__)
        cls.__init__ = __init__
        cls._mh_twin = _mh_twin
        for m in methods:
            if hasattr(cls, m):
                setattr(cls, m, wrap(m))
        return cls

    return deco


def function_fallback(ref_module: str, name: str):
    """Function decorator: a call the MI355X function cannot serve goes to the reference's function of the same name."""

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            try:
                return fn(*args, **kwargs)
            except _FALLBACK_ERRORS as e:
                ref = reference_object(ref_module, name)
                if ref is None:
                    raise
                _note(name, e)
                return ref(*args, **kwargs)

        wrapper._mh_is_product = True
        wrapper._mh_ref = (ref_module, name)
        return wrapper

    return deco
