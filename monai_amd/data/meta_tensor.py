"""A small ``MetaTensor``: a ``torch.Tensor`` subclass that carries ``meta`` (with the 4x4 ``affine``) and the list of
``applied_operations`` -- the parts of monai/data/meta_tensor.py:52-609 that the spatial transforms of this package
read and write.  When the real MONAI is installed its own ``MetaTensor`` works just as well: the transforms only
rely on the duck-typed surface (``.affine``, ``.meta``, ``.applied_operations``, ``.as_tensor()``,
``type(x)(tensor, affine=..., meta=..., applied_operations=...)``)."""

from __future__ import annotations

import copy
from typing import Any

import torch

__all__ = ["MetaTensor", "is_meta", "get_affine", "affine_np"]


class MetaTensor(torch.Tensor):
    @staticmethod
    def __new__(cls, x, affine=None, meta=None, applied_operations=None, *args, **kwargs):
        kw = {k: kwargs[k] for k in ("device", "dtype") if k in kwargs}
        return torch.as_tensor(x, **kw).as_subclass(cls)

    def __init__(self, x, affine=None, meta=None, applied_operations=None, *_a, **_k) -> None:
        super().__init__()
        if meta is not None:
            self.meta = dict(meta)
        elif isinstance(x, MetaTensor):
            self.meta = copy.deepcopy(x.meta)
        else:
            self.meta = {}
        if affine is not None:
            self.meta["affine"] = torch.as_tensor(affine, dtype=torch.float64)
        elif "affine" not in self.meta:
            self.meta["affine"] = torch.eye(4, dtype=torch.float64)
        if applied_operations is not None:
            self.applied_operations = list(applied_operations)
        elif isinstance(x, MetaTensor):
            self.applied_operations = copy.deepcopy(x.applied_operations)
        else:
            self.applied_operations = []

    # -- metadata surface ------------------------------------------------------------------------------
    @property
    def affine(self) -> torch.Tensor:
        return self.meta.get("affine", torch.eye(4, dtype=torch.float64))

    @affine.setter
    def affine(self, value) -> None:
        self.meta["affine"] = torch.as_tensor(value, dtype=torch.float64)

    @property
    def pixdim(self):
        a = self.affine.double()
        return torch.sqrt(torch.sum(a[:3, :3] * a[:3, :3], dim=0)).tolist()

    def as_tensor(self) -> torch.Tensor:
        return self.as_subclass(torch.Tensor)

    # -- pending (lazy) operations: monai/data/meta_obj.py:214-232, meta_tensor.py:470-509 -------------------------------------
    @property
    def pending_operations(self) -> list:
        ops = self.__dict__.get("_pending_operations")
        if ops is None:
            ops = self.__dict__["_pending_operations"] = []
        return ops

    @property
    def has_pending_operations(self) -> bool:
        return len(self.pending_operations) > 0

    def push_pending_operation(self, t) -> None:
        self.pending_operations.append(t)

    def pop_pending_operation(self):
        return self.pending_operations.pop()

    def clear_pending_operations(self) -> None:
        self.__dict__["_pending_operations"] = []

    def push_applied_operation(self, t) -> None:
        self.applied_operations.append(t)

    def peek_pending_shape(self):
        """spatial shape after the pending operations (the last one's ``lazy_shape``), monai/data/meta_tensor.py:470-480"""
        pend = self.pending_operations
        res = pend[-1].get("lazy_shape") if pend else None
        return tuple(int(v) for v in self.shape[1:]) if res is None else tuple(int(v) for v in res)

    def peek_pending_affine(self):
        """affine after the pending operations: ``affine @ A_1 @ A_2 ...`` (meta_tensor.py:482-497)"""
        res = self.affine.double()
        r = len(res) - 1
        for p in self.pending_operations:
            nxt = p.get("lazy_affine")
            if nxt is None:
                continue
            nxt = torch.as_tensor(nxt, dtype=torch.float64)
            if nxt.shape[0] - 1 != r:       # to_affine_nd: embed / crop to rank r
                full = torch.eye(r + 1, dtype=torch.float64)
                d = min(r, nxt.shape[0] - 1)
                full[:d, :d] = nxt[:d, :d]
                full[:d, -1] = nxt[:d, -1]
                nxt = full
            res = res @ nxt
        return res

    def peek_pending_rank(self) -> int:
        a = self.pending_operations[-1].get("lazy_affine") if self.pending_operations else self.affine
        return 1 if a is None else int(max(1, len(a) - 1))

    def copy_meta_from(self, src, copy_attr: bool = True, keys=None):
        self.meta = copy.deepcopy(getattr(src, "meta", {})) if copy_attr else dict(getattr(src, "meta", {}))
        ops = getattr(src, "applied_operations", [])
        self.applied_operations = copy.deepcopy(ops) if copy_attr else list(ops)
        self.__dict__["_pending_operations"] = list(getattr(src, "pending_operations", []))
        return self

    def __repr__(self, **kw):  # pragma: no cover
        return f"meta{self.as_tensor().__repr__()}"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None) -> Any:
        ret = super().__torch_function__(func, types, args, kwargs or {})
        if isinstance(ret, MetaTensor) and not hasattr(ret, "meta"):
            first = next((a for a in args if isinstance(a, MetaTensor) and hasattr(a, "meta")), None)
            if first is not None:
                ret.meta = dict(first.meta)
                ret.applied_operations = list(first.applied_operations)
                ret.__dict__["_pending_operations"] = list(first.pending_operations)
            else:
                ret.meta, ret.applied_operations = {"affine": torch.eye(4, dtype=torch.float64)}, []
        return ret


def is_meta(x) -> bool:
    """MetaTensor of this package or of an installed MONAI (duck-typed)."""
    return isinstance(x, torch.Tensor) and type(x) is not torch.Tensor and hasattr(x, "meta") and hasattr(x, "as_tensor")


def get_affine(x):
    return x.meta.get("affine") if is_meta(x) and "affine" in x.meta else None


def affine_np(x, default_rank: int = 3):
    """the image's affine as a float64 numpy matrix on the host (wherever the MetaTensor's meta keeps it); identity when there is none"""
    import numpy as np

    a = x.meta.get("affine") if is_meta(x) else None
    if a is None:
        return np.eye(default_rank + 1)
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu()
    return np.asarray(a, dtype=np.float64)
