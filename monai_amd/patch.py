"""Install the MI355X implementations into an installed MONAI, so existing bundles / scripts pick them up unchanged.

    import monai_amd.patch; monai_amd.patch.install()

Bundle configs name components by short name (``"_target_": "SlidingWindowInferer"``); MONAI's ``ComponentLocator``
(monai/bundle/config_item.py:60-112) resolves those by scanning ``sys.modules`` for modules whose name starts with
``monai`` and indexing the classes whose ``__module__`` equals that module.  ``install()`` therefore rebinds the
reference's names IN PLACE -- e.g. ``monai.inferers.inferer.SlidingWindowInferer = monai_amd...SlidingWindowInferer``
with ``__module__ = "monai.inferers.inferer"`` -- and re-exports them from the parent packages.  When MONAI was
installed without its compiled extension (``monai._C`` missing), ``monai_amd._C`` is registered under that name.
Without MONAI installed this module is not needed: use ``"_target_": "monai_amd.inferers.SlidingWindowInferer"``.
"""

from __future__ import annotations

import importlib
import os
import sys

_TARGETS = {
    # reference module                         : {name: (our module, our name)}
    "monai.inferers.inferer": {
        "SlidingWindowInferer": ("monai_amd.inferers.inferer", "SlidingWindowInferer"),
        "SlidingWindowInfererAdapt": ("monai_amd.inferers.inferer", "SlidingWindowInfererAdapt"),
        "SliceInferer": ("monai_amd.inferers.inferer", "SliceInferer"),
        "PatchInferer": ("monai_amd.inferers.inferer", "PatchInferer"),
    },
    "monai.inferers.splitter": {
        "Splitter": ("monai_amd.inferers.splitter", "Splitter"),
        "SlidingWindowSplitter": ("monai_amd.inferers.splitter", "SlidingWindowSplitter"),
    },
    "monai.inferers.merger": {"Merger": ("monai_amd.inferers.merger", "Merger"), "AvgMerger": ("monai_amd.inferers.merger", "AvgMerger")},
    "monai.inferers.utils": {"sliding_window_inference": ("monai_amd.inferers.utils", "sliding_window_inference")},
    "monai.networks.nets.basic_unet": {
        "BasicUNet": ("monai_amd.networks.nets.basic_unet", "BasicUNet"),
        "BasicUnet": ("monai_amd.networks.nets.basic_unet", "BasicUNet"),
        "Basicunet": ("monai_amd.networks.nets.basic_unet", "BasicUNet"),
        "basicunet": ("monai_amd.networks.nets.basic_unet", "BasicUNet"),
    },
    "monai.networks.nets.dynunet": {
        "DynUNet": ("monai_amd.networks.nets.dynunet", "DynUNet"),
        "DynUnet": ("monai_amd.networks.nets.dynunet", "DynUNet"),
        "Dynunet": ("monai_amd.networks.nets.dynunet", "DynUNet"),
    },
    "monai.networks.nets.segresnet": {"SegResNet": ("monai_amd.networks.nets.segresnet", "SegResNet")},
    "monai.networks.nets.unetr": {"UNETR": ("monai_amd.networks.nets.unetr", "UNETR")},
    "monai.networks.nets.swin_unetr": {"SwinUNETR": ("monai_amd.networks.nets.swin_unetr", "SwinUNETR")},
    "monai.networks.nets.unet": {"UNet": ("monai_amd.networks.nets.unet", "UNet"), "Unet": ("monai_amd.networks.nets.unet", "UNet")},
    "monai.transforms.spatial.array": {
        "Spacing": ("monai_amd.transforms.spatial.array", "Spacing"),
        "SpatialResample": ("monai_amd.transforms.spatial.array", "SpatialResample"),
        "Resample": ("monai_amd.transforms.spatial.array", "Resample"),
        "Orientation": ("monai_amd.transforms.spatial.orientation", "Orientation"),
        "Flip": ("monai_amd.transforms.spatial.flip_rotate", "Flip"),
        "Rotate90": ("monai_amd.transforms.spatial.flip_rotate", "Rotate90"),
    },
    "monai.transforms.spatial.dictionary": {
        "Spacingd": ("monai_amd.transforms.spatial.dictionary", "Spacingd"),
        "SpacingD": ("monai_amd.transforms.spatial.dictionary", "Spacingd"),
        "SpacingDict": ("monai_amd.transforms.spatial.dictionary", "Spacingd"),
        "Orientationd": ("monai_amd.transforms.spatial.orientation", "Orientationd"),
        "OrientationD": ("monai_amd.transforms.spatial.orientation", "Orientationd"),
        "OrientationDict": ("monai_amd.transforms.spatial.orientation", "Orientationd"),
        **{n + suffix: ("monai_amd.transforms.spatial.flip_rotate", n + "d") for n in ("Flip", "Rotate90") for suffix in ("d", "D", "Dict")},
    },
    "monai.transforms.intensity.array": {
        "GaussianSmooth": ("monai_amd.transforms.intensity.array", "GaussianSmooth"),
        "ScaleIntensityRange": ("monai_amd.transforms.intensity.array", "ScaleIntensityRange"),
        "NormalizeIntensity": ("monai_amd.transforms.intensity.array", "NormalizeIntensity"),
        "ScaleIntensity": ("monai_amd.transforms.intensity.array", "ScaleIntensity"),
    },
    "monai.transforms.croppad.array": {n: ("monai_amd.transforms.croppad.array", n) for n in
                                       ("CropForeground", "Pad", "SpatialPad", "BorderPad", "DivisiblePad", "Crop", "SpatialCrop", "CenterSpatialCrop")},
    "monai.transforms.croppad.dictionary": {
        "CropForegroundd": ("monai_amd.transforms.croppad.dictionary", "CropForegroundd"),
        "CropForegroundD": ("monai_amd.transforms.croppad.dictionary", "CropForegroundd"),
        "CropForegroundDict": ("monai_amd.transforms.croppad.dictionary", "CropForegroundd"),
        **{n + suffix: ("monai_amd.transforms.croppad.dictionary", n + "d") for n in ("SpatialPad", "BorderPad", "DivisiblePad", "SpatialCrop", "CenterSpatialCrop")
           for suffix in ("d", "D", "Dict")},
    },
    "monai.transforms.post.array": {
        "Activations": ("monai_amd.transforms.post.array", "Activations"),
        "AsDiscrete": ("monai_amd.transforms.post.array", "AsDiscrete"),
    },
    "monai.transforms.post.dictionary": {
        "Activationsd": ("monai_amd.transforms.post.dictionary", "Activationsd"),
        "ActivationsD": ("monai_amd.transforms.post.dictionary", "Activationsd"),
        "ActivationsDict": ("monai_amd.transforms.post.dictionary", "Activationsd"),
        "AsDiscreted": ("monai_amd.transforms.post.dictionary", "AsDiscreted"),
        "AsDiscreteD": ("monai_amd.transforms.post.dictionary", "AsDiscreted"),
        "AsDiscreteDict": ("monai_amd.transforms.post.dictionary", "AsDiscreted"),
    },
    "monai.transforms.intensity.dictionary": {
        "GaussianSmoothd": ("monai_amd.transforms.intensity.dictionary", "GaussianSmoothd"),
        "GaussianSmoothD": ("monai_amd.transforms.intensity.dictionary", "GaussianSmoothd"),
        "GaussianSmoothDict": ("monai_amd.transforms.intensity.dictionary", "GaussianSmoothd"),
        "ScaleIntensityRanged": ("monai_amd.transforms.intensity.dictionary", "ScaleIntensityRanged"),
        "ScaleIntensityRangeD": ("monai_amd.transforms.intensity.dictionary", "ScaleIntensityRanged"),
        "ScaleIntensityRangeDict": ("monai_amd.transforms.intensity.dictionary", "ScaleIntensityRanged"),
        "NormalizeIntensityd": ("monai_amd.transforms.intensity.dictionary", "NormalizeIntensityd"),
        "NormalizeIntensityD": ("monai_amd.transforms.intensity.dictionary", "NormalizeIntensityd"),
        "NormalizeIntensityDict": ("monai_amd.transforms.intensity.dictionary", "NormalizeIntensityd"),
        "ScaleIntensityd": ("monai_amd.transforms.intensity.dictionary", "ScaleIntensityd"),
        "ScaleIntensityD": ("monai_amd.transforms.intensity.dictionary", "ScaleIntensityd"),
        "ScaleIntensityDict": ("monai_amd.transforms.intensity.dictionary", "ScaleIntensityd"),
    },
    "monai.networks.layers.spatial_transforms": {
        "AffineTransform": ("monai_amd.networks.layers.spatial_transforms", "AffineTransform"),
        "grid_pull": ("monai_amd.networks.layers.spatial_transforms", "grid_pull"),
        "grid_push": ("monai_amd.networks.layers.spatial_transforms", "grid_push"),
        "grid_count": ("monai_amd.networks.layers.spatial_transforms", "grid_count"),
        "grid_grad": ("monai_amd.networks.layers.spatial_transforms", "grid_grad"),
    },
    "monai.networks.layers.simplelayers": {"GaussianFilter": ("monai_amd.networks.layers.simplelayers", "GaussianFilter")},
    "monai.networks.blocks.warp": {"Warp": ("monai_amd.networks.blocks.warp", "Warp"), "DVF2DDF": ("monai_amd.networks.blocks.warp", "DVF2DDF")},
}
# parent packages that re-export the names above
_REEXPORT = ["monai.inferers", "monai.networks.nets", "monai.transforms", "monai.networks.layers", "monai.networks.blocks"]

_installed: dict = {}


def install(native_module: bool = True) -> list:
    """Rebind the reference's names to the MI355X implementations.  Returns the list of rebound ``module.name``s."""
    import monai  # noqa: F401  (ImportError if MONAI is not installed: nothing to patch)

    done = []
    for ref_mod_name, names in _TARGETS.items():
        ref_mod = importlib.import_module(ref_mod_name)
        for name, (our_mod_name, our_name) in names.items():
            obj = getattr(importlib.import_module(our_mod_name), our_name)
            if (ref_mod_name, name) not in _installed:
                _installed[(ref_mod_name, name)] = getattr(ref_mod, name, None)
            try:
                obj.__module__ = ref_mod_name      # what ComponentLocator compares against
            except (AttributeError, TypeError):
                pass
            setattr(ref_mod, name, obj)
            for parent in _REEXPORT:
                pm = sys.modules.get(parent)
                if pm is not None and ref_mod_name.startswith(parent) and hasattr(pm, name):
                    if (parent, name) not in _installed:
                        _installed[(parent, name)] = getattr(pm, name)
                    setattr(pm, name, obj)
            done.append(f"{ref_mod_name}.{name}")
    _register_transform_traits()
    if native_module and "monai._C" not in sys.modules:
        try:
            importlib.import_module("monai._C")
        except Exception:
            from . import _C

            sys.modules["monai._C"] = _C
            done.append("monai._C")
    return done


def _register_transform_traits() -> None:
    """MONAI's ``Compose`` / ``Invertd`` / ``allow_missing_keys_mode`` select transforms with ``isinstance`` checks against ``MapTransform`` and
    ``InvertibleTransform`` (monai/transforms/compose.py:600-625, monai/transforms/utils.py:1705-1745).  ``Transform`` is an ABC, so the
    MI355X classes are registered as VIRTUAL subclasses: dictionary transforms as ``MapTransform``, everything with an ``inverse`` as
    ``InvertibleTransform`` -- a bundle's ``Invertd(transform="@preprocessing")`` then walks them like the reference's own classes."""
    from monai.transforms import InvertibleTransform, MapTransform, Transform

    for ref_mod_name, names in _TARGETS.items():
        if not ref_mod_name.startswith("monai.transforms."):
            continue
        for our_mod_name, our_name in set(names.values()):
            cls = getattr(importlib.import_module(our_mod_name), our_name)
            if not isinstance(cls, type):
                continue
            Transform.register(cls)
            if ref_mod_name.endswith(".dictionary"):
                MapTransform.register(cls)
            if callable(getattr(cls, "inverse", None)):
                InvertibleTransform.register(cls)
    # lazy resampling: `Compose(..., lazy=True)` asks `isinstance(t, LazyTrait)` (monai/transforms/lazy/functional.py:185-188,
    # monai/transforms/transform.py:90-134) -- LazyTrait is a plain class, so it becomes a real base of the lazy-capable mixin --
    # and executes pending operations through `resample` (monai/transforms/lazy/utils.py:148-229), rebound to the fused kernel path
    from monai.transforms.traits import LazyTrait

    from .transforms import lazy as _lazy

    if LazyTrait not in _lazy.LazyCapable.__bases__:
        _lazy.LazyCapable.__bases__ = _lazy.LazyCapable.__bases__ + (LazyTrait,)
    import monai.transforms.lazy.functional as _lf
    import monai.transforms.lazy.utils as _lu

    for mod in (_lf, _lu):
        if ("lazy", mod.__name__) not in _installed:
            _installed[("lazy", mod.__name__)] = mod.resample
        mod.resample = _resample_or_reference
    from monai.inferers import Inferer          # an ABC as well: `isinstance(x, Inferer)` checks in user code keep holding

    for our_mod_name, our_name in set(_TARGETS["monai.inferers.inferer"].values()):
        Inferer.register(getattr(importlib.import_module(our_mod_name), our_name))


def _resample_or_reference(data, matrix, kwargs=None):
    """`monai.transforms.lazy.utils.resample` once installed: the fused kernel path, and -- boundary B3 -- the reference's own function
    for what that path does not cover (spline-order interpolation, CPU tensors, other dtypes)."""
    from . import _fallback
    from .transforms import lazy as _lazy

    try:
        return _lazy.resample(data, matrix, kwargs)
    except _fallback._FALLBACK_ERRORS as e:
        ref = _installed.get(("lazy", "monai.transforms.lazy.utils"))
        if ref is None or os.environ.get("MONAI_AMD_NO_FALLTHROUGH") == "1":
            raise
        _fallback._note("lazy.resample", e)
        return ref(data, matrix, kwargs)


def uninstall() -> None:
    """Restore the reference's own objects."""
    for (mod_name, name), obj in list(_installed.items()):
        mod = sys.modules.get(mod_name)
        if mod is not None and obj is not None:
            setattr(mod, name, obj)
    for (kind, mod_name), obj in list(_installed.items()):
        if kind == "lazy" and sys.modules.get(mod_name) is not None:
            sys.modules[mod_name].resample = obj
    _installed.clear()
    try:
        from .transforms import lazy as _lazy

        _lazy.LazyCapable.__bases__ = (_lazy._Root,)
    except Exception:
        pass
    if sys.modules.get("monai._C") is not None and sys.modules["monai._C"].__name__ == "monai_amd._C":
        del sys.modules["monai._C"]
