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
import os
import threading
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
    """The reference's own ``ref_module.name`` (the displaced object when the patch is installed), or None without MONAI
    (or when MONAI_AMD_NO_FALLTHROUGH=1 asks for the explicit errors: strict deployments, and the tests that pin them)."""
    if os.environ.get("MONAI_AMD_NO_FALLTHROUGH") == "1":
        return None
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


_active = threading.local()       # ids of the instances whose wrapped method is running: only the OUTERMOST wrapper falls through


def reference_fallback(ref_module: str, name: str, methods=("__call__",), share_state: bool = False):
    """Class decorator: see the module docstring.  `methods` are wrapped for call-time fall-through; `share_state` (networks) makes
    the twin share parameters / buffers and follow ``.training``.  An unsupported configuration turns the object under
    construction into an instance of the reference class (``__class__`` assignment inside ``__init__``)."""

    def deco(cls):
        cls._mh_is_product = True
        cls._mh_ref = (ref_module, name)

        orig_init = cls.__init__

        @functools.wraps(orig_init)
        def __init__(self, *args, **kwargs):
            try:
                orig_init(self, *args, **kwargs)
            except NotImplementedError as e:
                own = type(self).__dict__.get("_mh_ref")      # a user's subclass must not silently become the reference BASE class
                ref = reference_object(*own) if own else None
                if ref is None or not isinstance(ref, type):
                    raise
                _note(f"{own[1]}(...)", e)
                # the object BECOMES an instance of the reference class, built from the same arguments (copy / pickle never come
                # here); where the two classes' C layouts differ (__slots__ somewhere in the reference's bases) it becomes a proxy
                # instead: every method and attribute goes to a reference instance held in `_mh_delegate`
                self.__dict__.clear()
                try:
                    self.__class__ = ref
                except TypeError:
                    object.__setattr__(self, "_mh_delegate", ref(*args, **kwargs))
                    return
                ref.__init__(self, *args, **kwargs)
                return
            if "_mh_ctor" not in self.__dict__:               # the outermost (most derived) constructor call wins
                object.__setattr__(self, "_mh_ctor", (args, kwargs))

        cls.__init__ = __init__

        def _mh_twin(self):
            own = type(self).__dict__.get("_mh_ref")
            if own is None:
                return None
            twin = self.__dict__.get("_mh_twin_obj")
            if twin is None:
                ref = reference_object(*own)
                if ref is None:
                    return None
                args, kwargs = self.__dict__.get("_mh_ctor", ((), {}))
                twin = ref(*args, **kwargs)
                if share_state:
                    _share_module_state(self, twin)
                object.__setattr__(self, "_mh_twin_obj", twin)
            if share_state:
                twin.train(self.training)
            elif "lazy" in self.__dict__ or hasattr(type(self), "lazy"):     # transforms: keep the user-visible switch in step
                try:
                    twin.lazy = self.lazy
                except Exception:
                    pass
            return twin

        def wrap(mname):
            orig = getattr(cls, mname)

            @functools.wraps(orig)
            def method(self, *args, **kwargs):
                if "_mh_delegate" in self.__dict__:
                    return getattr(self.__dict__["_mh_delegate"], mname)(*args, **kwargs)
                stack = _active.__dict__.setdefault("ids", [])
                outermost = id(self) not in stack
                stack.append(id(self))
                try:
                    return orig(self, *args, **kwargs)
                except _FALLBACK_ERRORS as e:
                    if not outermost:
                        raise
                    twin = self._mh_twin()
                    if twin is None:
                        raise
                    _note(f"{type(self).__name__}.{mname}", e)
                    err = e
                finally:
                    stack.pop()
                del err
                return getattr(twin, mname)(*args, **kwargs)

            return method

        cls._mh_twin = _mh_twin
        for m in methods:
            if m in cls.__dict__ or (hasattr(cls, m) and not getattr(getattr(cls, m), "__wrapped__", None)):
                setattr(cls, m, wrap(m))
        if not share_state and "__getattr__" not in cls.__dict__ and not any("__getattr__" in b.__dict__ for b in cls.__mro__[1:-1]):
            import types

            def delegating(fname, fn):
                @functools.wraps(fn)
                def method(self, *args, **kwargs):
                    d = self.__dict__.get("_mh_delegate")
                    return fn(self, *args, **kwargs) if d is None else getattr(d, fname)(*args, **kwargs)

                return method

            for fname, fn in list(cls.__dict__.items()):      # proxy mode: the class's other public methods go to the delegate too
                if isinstance(fn, types.FunctionType) and not fname.startswith("_") and fname not in methods:
                    setattr(cls, fname, delegating(fname, fn))

            def __getattr__(self, attr):                      # only reached for attributes the (emptied) proxy does not have
                d = self.__dict__.get("_mh_delegate")
                if d is None or attr.startswith("__"):
                    raise AttributeError(f"{type(self).__name__!r} object has no attribute {attr!r}")
                return getattr(d, attr)

            def __setattr__(self, attr, value):
                d = self.__dict__.get("_mh_delegate")
                if d is None:
                    object.__setattr__(self, attr, value)
                else:
                    setattr(d, attr, value)

            cls.__getattr__ = __getattr__
            if "__setattr__" not in cls.__dict__:
                cls.__setattr__ = __setattr__
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


def apply_all() -> None:
    """Decorate every class / function `monai_amd.patch` can install with its fall-through to the reference object of the same
    name (the table of `patch._TARGETS` is the single list of what this package stands in for).  Idempotent."""
    import inspect

    from . import patch

    seen = set()
    for ref_mod, names in patch._TARGETS.items():
        for name, (our_mod, our_name) in names.items():
            mod = importlib.import_module(our_mod)
            obj = getattr(mod, our_name)
            if id(obj) in seen or getattr(obj, "_mh_ref", None) is not None:
                seen.add(id(obj))
                continue
            seen.add(id(obj))
            if inspect.isclass(obj):
                is_module = any(b.__name__ == "Module" and b.__module__.startswith("torch.nn") for b in obj.__mro__)
                if is_module:
                    reference_fallback(ref_mod, name, methods=("forward",), share_state=True)(obj)
                else:
                    methods = tuple(m for m in ("__call__", "inverse", "aggregate", "finalize") if callable(getattr(obj, m, None)))
                    reference_fallback(ref_mod, name, methods=methods)(obj)
            # plain functions are decorated where they are defined (other modules hold direct references to them)
