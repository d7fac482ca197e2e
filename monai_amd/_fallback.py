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


# ---- the reference's image-in / image-out convention of array transforms -------------------------------------------------------
# Every array transform of the reference starts with ``convert_to_tensor(img, track_meta=get_track_meta())`` (e.g.
# monai/transforms/intensity/array.py:1618, spatial/array.py:505): numpy arrays become tensors, and the result is a MetaTensor whenever
# MONAI's global meta tracking is on (the default) -- also for a plain input -- and a plain tensor when it is off -- also for a
# MetaTensor input (tests/transforms/test_spacing.py:300-317, test_gaussian_smooth.py:92 pin both).  Applied around the product's
# ``__call__`` while the patch is installed; used on its own the package keeps plain inputs plain and its own MetaTensor's metadata.
_IMAGE_KWARGS = ("img", "data_array", "data")


def _track_meta_state():
    """(MONAI's MetaTensor class, tracking on?) while this package stands in for the reference (`patch.install()`), else (None, None):
    used on its own the package keeps plain inputs plain."""
    try:
        from . import patch

        if not patch._installed:
            return None, None
        from monai.data.meta_obj import get_track_meta
        from monai.data.meta_tensor import MetaTensor
    except Exception:
        return None, None
    return MetaTensor, bool(get_track_meta())


def _image_in(args, kwargs):
    import numpy as np
    import torch

    key = None
    if args:
        img = args[0]
    else:
        key = next((k for k in _IMAGE_KWARGS if k in kwargs), None)
        if key is None:
            return args, kwargs
        img = kwargs[key]
    new = img
    if isinstance(new, np.ndarray):
        new = torch.as_tensor(np.ascontiguousarray(new))
    if isinstance(new, torch.Tensor):
        meta_cls, track = _track_meta_state()
        if meta_cls is not None and track and type(new) is torch.Tensor:
            new = meta_cls(new)
    if new is img:
        return args, kwargs
    if key is None:
        return (new,) + tuple(args[1:]), kwargs
    return args, dict(kwargs, **{key: new})


def _dict_in(obj, args, kwargs):
    """dictionary transforms: the same convention for the entries named by ``keys`` (the reference's array transform inside the
    dictionary transform applies it per entry)"""
    if not args or not isinstance(args[0], dict):
        return args, kwargs
    d = dict(args[0])
    changed = False
    for k in getattr(obj, "keys", ()) or ():
        if k in d:
            (v2,), _ = _image_in((d[k],), {})
            if v2 is not d[k]:
                d[k], changed = v2, True
    return ((d,) + tuple(args[1:]), kwargs) if changed else (args, kwargs)


def _dict_out(obj, out):
    if isinstance(out, dict):
        for k in getattr(obj, "keys", ()) or ():
            if k in out:
                out[k] = _image_out(out[k])
    return out


def _image_out(out):
    import torch

    meta_cls, track = _track_meta_state()
    if meta_cls is None:
        return out

    def one(t):
        if not isinstance(t, torch.Tensor):
            return t
        is_meta = type(t) is not torch.Tensor and hasattr(t, "as_tensor")
        if track and not is_meta:
            return meta_cls(t)
        if not track and is_meta:
            return t.as_tensor()
        return t

    if isinstance(out, tuple):
        return tuple(one(t) for t in out)
    if isinstance(out, list):
        return [one(t) for t in out]
    return one(out)


def _drop_new_records(img_in, out) -> None:
    """`tracing` off (the reference's inverse methods run their helper transforms under ``trace_transform(False)``, e.g.
    monai/transforms/croppad/array.py:430-434): the call must leave no record of itself on the image."""
    n = len(getattr(img_in, "applied_operations", None) or [])
    for t in (out if isinstance(out, (tuple, list)) else (out,)):
        ops_ = getattr(t, "applied_operations", None)
        if ops_ is not None and len(ops_) > n:
            t.applied_operations = list(ops_[:n])


class _Tracing:
    """The part of the reference's ``TraceableTransform`` interface (monai/transforms/inverse.py:68-110) that its own code uses on the
    transforms it constructs by name -- which are this package's classes once the patch is installed."""

    tracing = True

    def trace_transform(self, to_trace: bool):
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev = self.tracing
            self.tracing = to_trace
            try:
                yield
            finally:
                self.tracing = prev

        return ctx()


def _recorded_by_twin(obj, data) -> bool:
    """True when the most recent applied operation on `data` (an image, or a dictionary of images) carries the `id` of obj's reference
    twin or of one of the twin's member transforms (a dictionary transform records through its array transform) -- i.e. the forward
    call of this object fell through to the reference (monai/transforms/inverse.py:112-160 keys its records by `id(self)`)."""
    twin = obj.__dict__.get("_mh_twin_obj")
    if twin is None:
        return False
    ids = {id(twin)} | {id(v) for v in getattr(twin, "__dict__", {}).values()}
    images = list(data.values()) if isinstance(data, dict) else [data]
    for im in images:
        ops_ = getattr(im, "applied_operations", None)
        if ops_ and isinstance(ops_[-1], dict) and ops_[-1].get("id") in ids:
            return True
    return False


def reference_fallback(ref_module: str, name: str, methods=("__call__",), share_state: bool = False, image_io: bool = False, dict_io: bool = False):
    """Class decorator: see the module docstring.  `methods` are wrapped for call-time fall-through; `share_state` (networks) makes
    the twin share parameters / buffers and follow ``.training``.  An unsupported configuration turns the object under
    construction into an instance of the reference class (``__class__`` assignment inside ``__init__``)."""

    def deco(cls):
        cls._mh_is_product = True
        cls._mh_ref = (ref_module, name)

        orig_init = cls.__init__

        @functools.wraps(orig_init)
        def __init__(self, *args, **kwargs):
            # the constructor call of the most derived class is the one a reference twin is built from: it ENTERS first (the bases'
            # decorated constructors run inside it, through super().__init__) and records its arguments when it returns
            ctors = _active.__dict__.setdefault("ctors", set())
            outermost_ctor = id(self) not in ctors
            ctors.add(id(self))
            try:
                orig_init(self, *args, **kwargs)
            except NotImplementedError as e:
                if not outermost_ctor:
                    raise                                     # the most derived constructor decides what the object becomes
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
            finally:
                if outermost_ctor:
                    ctors.discard(id(self))
            if outermost_ctor:
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
                object.__setattr__(self, "_mh_twin_obj", twin)
            if share_state:
                # on EVERY fetch, not only when the twin is built: nn.Module._apply (.to / .cuda / .half) swaps parameters' .data in place but REPLACES buffer
                # tensors in _buffers, and load_state_dict(assign=True) replaces both -- a twin bound once would keep the old-device / old-dtype objects
                _share_module_state(self, twin)
                twin.train(self.training)
            elif "lazy" in self.__dict__ or hasattr(type(self), "lazy"):     # transforms: keep the user-visible switch in step
                try:
                    twin.lazy = self.lazy
                except Exception:
                    pass
            if not share_state and hasattr(twin, "tracing") and hasattr(self, "tracing"):
                try:
                    twin.tracing = self.tracing      # trace_transform(False) around a call that falls through
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
                    if mname == "inverse" and image_io and args and not (hasattr(args[0], "applied_operations") or isinstance(args[0], dict)):
                        # monai/transforms/inverse.py:343-347 (`get_most_recent_transform`)
                        raise ValueError(f"`data` should be either `MetaTensor` or dictionary, got {type(args[0])}.")
                    if mname == "inverse" and args and _recorded_by_twin(self, args[0]):
                        # the forward call fell through: the record on the image is the reference twin's (its `id`), only it can undo it
                        return self.__dict__["_mh_twin_obj"].inverse(*args, **kwargs)
                    if image_io and mname == "inverse" and args and hasattr(args[0], "applied_operations"):
                        # the reference's inverse POPS the record from the object it is given (inverse.py:353-370, `pop_transform`): callers
                        # -- RandFlip.inverse re-appends the inner record and hands the same object on -- rely on that side effect
                        n_before = len(args[0].applied_operations)
                        out = orig(self, *args, **kwargs)
                        ops_ = args[0].applied_operations
                        if out is not args[0] and len(ops_) == n_before and n_before > 0:
                            ops_.pop()
                        return out
                    if image_io and mname == "__call__":
                        a2, k2 = _image_in(args, kwargs)
                        out = _image_out(orig(self, *a2, **k2))
                        if not getattr(self, "tracing", True):
                            img_in = a2[0] if a2 else next((k2[k] for k in _IMAGE_KWARGS if k in k2), None)
                            _drop_new_records(img_in, out)
                        return out
                    if dict_io and mname == "__call__":
                        a2, k2 = _dict_in(self, args, kwargs)
                        return _dict_out(self, orig(self, *a2, **k2))
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
        if share_state and "__prepare_scriptable__" not in cls.__dict__:
            def __prepare_scriptable__(self):
                """`torch.jit.script(net)` (bundle export, tests/networks/nets/test_basic_unet.py:95-98) asks for a scriptable stand-in: the engine's
                schedule is not TorchScript, the reference twin -- same parameters -- is."""
                twin = self._mh_twin()
                if twin is None:
                    raise NotImplementedError(f"monai_amd.{type(self).__name__}: TorchScript export needs the reference implementation (MONAI is not importable)")
                _note(f"{type(self).__name__}.__prepare_scriptable__", NotImplementedError("the HIP engine is not TorchScript"))
                return twin

            cls.__prepare_scriptable__ = __prepare_scriptable__
        if image_io and not hasattr(cls, "trace_transform"):
            cls.tracing = _Tracing.tracing
            cls.trace_transform = _Tracing.trace_transform
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
                # keywords of this package's own (e.g. the fused-argmax switch of sliding_window_argmax) mean nothing to the reference -- it would forward
                # them to the predictor: they are stripped, and what they stood for is applied to the reference's result
                private = {k: kwargs.pop(k) for k in [k for k in kwargs if k.startswith("_monai_amd_")]}
                result = ref(*args, **kwargs)
                finish = getattr(fn, "_mh_after_reference", None)
                return finish(result, private) if (finish is not None and private) else result

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
            # own attribute only: a subclass of an already decorated class (AvgMerger < Merger, SliceInferer < SlidingWindowInferer)
            # inherits the base's `_mh_ref` and still needs its own
            own_ref = vars(obj).get("_mh_ref") if inspect.isclass(obj) else getattr(obj, "_mh_ref", None)
            if id(obj) in seen or own_ref is not None:
                seen.add(id(obj))
                continue
            seen.add(id(obj))
            if inspect.isclass(obj):
                is_module = any(b.__name__ == "Module" and b.__module__.startswith("torch.nn") for b in obj.__mro__)
                if is_module:
                    reference_fallback(ref_mod, name, methods=("forward",), share_state=True)(obj)
                else:
                    methods = tuple(m for m in ("__call__", "inverse", "aggregate", "finalize") if callable(getattr(obj, m, None)))
                    # array transforms (monai.transforms.<family>.array) follow the reference's image-in / image-out convention
                    image_io = ref_mod.startswith("monai.transforms.") and ref_mod.endswith(".array")
                    dict_io = ref_mod.startswith("monai.transforms.") and ref_mod.endswith(".dictionary")
                    reference_fallback(ref_mod, name, methods=methods, image_io=image_io, dict_io=dict_io)(obj)
            # plain functions are decorated where they are defined (other modules hold direct references to them)
