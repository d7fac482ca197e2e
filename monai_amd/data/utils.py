"""Host-side index math of the sliding-window path (tiny; runs once per call on the CPU).

Same results as the reference helpers in monai/data/utils.py -- ``dense_patch_slices`` :166-206,
``get_valid_patch_size`` :343-354, ``compute_importance_map`` :1084-1134 -- but organised around the
per-axis start lists that the blend / gather kernels take (the window grid is their cartesian product).
"""

from __future__ import annotations

import itertools
import math
from collections.abc import Sequence

import numpy as np
import torch

from ..utils.misc import ensure_tuple_rep, look_up_option

__all__ = [
    "window_starts", "dense_patch_slices", "get_valid_patch_size", "compute_importance_map",
    "affine_to_spacing", "to_affine_nd", "zoom_affine", "compute_shape_offset",
]


def get_valid_patch_size(image_size: Sequence[int], patch_size) -> tuple:
    """A patch dimension that is 0/None, missing, or larger than the image becomes the image dimension."""
    ps = tuple(patch_size) if isinstance(patch_size, (Sequence, torch.Size)) else (patch_size,)
    ps = (ps + (0,) * len(image_size))[: len(image_size)]  # ensure_tuple_size semantics: missing axes -> whole axis
    return tuple(min(int(m), int(p) if p else int(m)) for m, p in zip(image_size, ps))


def window_starts(image_size: Sequence[int], patch_size: Sequence[int], scan_interval: Sequence[int]) -> list:
    """Per-axis ascending window start indices.

    Along one axis, window k sits at ``k * interval`` and is pulled back so that it ends inside the image; windows
    are generated up to and including the first one that reaches the image end."""
    patch = get_valid_patch_size(image_size, patch_size)
    steps = tuple(scan_interval) + (0,) * (len(image_size) - len(tuple(scan_interval)))
    axes = []
    for size, p, step in zip(image_size, patch, steps):
        if step == 0:
            count = 1
        else:
            count = 1
            for k in range(int(math.ceil(float(size) / step))):
                if k * step + p >= size:
                    count = k + 1
                    break
        axes.append([min(k * step, size - p) for k in range(count)])
    return axes


def dense_patch_slices(image_size, patch_size, scan_interval, return_slice: bool = True) -> list:
    """All window slices, row-major over the per-axis starts (last axis fastest)."""
    patch = get_valid_patch_size(image_size, patch_size)
    axes = window_starts(image_size, patch, scan_interval)
    if return_slice:
        return [tuple(slice(s, s + patch[d]) for d, s in enumerate(w)) for w in itertools.product(*axes)]
    return [tuple((s, s + patch[d]) for d, s in enumerate(w)) for w in itertools.product(*axes)]


def compute_importance_map(patch_size, mode="constant", sigma_scale=0.125, device="cpu", dtype=torch.float32) -> torch.Tensor:
    """Window weight map.  "constant": ones.  "gaussian": outer product over the axes of
    ``exp(x^2 / (-2 sigma^2))`` with ``x`` centred integer offsets and ``sigma = sigma_scale * size``; clamped from
    below by ``max(min(map), 1e-3)``.

    The 1-D factors are always evaluated with torch on the HOST in fp32 and the outer product / clamp in fp32 as
    well, so the map is bit-identical to the reference's CPU map whatever `device` is (a device-side ``expf`` may
    differ by an ulp; SURVEY.md section 7 "reproducibility quirks")."""
    mode = look_up_option(mode, ("constant", "gaussian"), "mode")
    patch_size = tuple(int(p) for p in patch_size)
    if mode == "constant":
        weights = torch.ones(patch_size, dtype=torch.float)
    else:
        scales = ensure_tuple_rep(sigma_scale, len(patch_size))
        weights = None
        for axis, (n, s) in enumerate(zip(patch_size, scales)):
            sigma = n * s
            offs = torch.arange(start=-(n - 1) / 2.0, end=(n - 1) / 2.0 + 1, dtype=torch.float)
            g = torch.exp(offs**2 / (-2 * sigma**2))
            weights = g if axis == 0 else weights.unsqueeze(-1) * g[(None,) * axis]
    floor = max(torch.min(weights).item(), 1e-3)
    weights = torch.clamp_(weights.to(torch.float), min=floor).to(dtype)
    return weights.to(device)


def importance_map_factors(patch_size, mode="constant", sigma_scale=0.125):
    """The 1-D factors behind ``compute_importance_map`` for a 3-D patch: ``(gz, gy, gx, floor)`` with
    ``map[z, y, x] == max(fl(fl(gz[z] * gy[y]) * gx[x]), floor)`` in fp32 -- the order in which the reference multiplies the axes together
    (monai/data/utils.py:1113-1126).  The blend kernel re-forms the map from them in registers; ``None`` when the patch is not 3-D or the
    factorisation does not reproduce the map bit for bit (checked here, on the host)."""
    patch_size = tuple(int(p) for p in patch_size)
    if len(patch_size) != 3:
        return None
    if look_up_option(mode, ("constant", "gaussian"), "mode") == "constant":
        fac = [torch.ones(n, dtype=torch.float) for n in patch_size]
    else:
        fac = []
        for n, s in zip(patch_size, ensure_tuple_rep(sigma_scale, 3)):
            offs = torch.arange(start=-(n - 1) / 2.0, end=(n - 1) / 2.0 + 1, dtype=torch.float)
            fac.append(torch.exp(offs**2 / (-2 * (n * s) ** 2)))
    full = compute_importance_map(patch_size, mode=mode, sigma_scale=sigma_scale, device="cpu", dtype=torch.float32)
    prod = (fac[0][:, None] * fac[1][None, :])[:, :, None] * fac[2][None, None, :]
    floor = float(max(prod.min().item(), 1e-3))
    if not torch.equal(torch.clamp(prod, min=floor), full):
        return None
    return fac[0], fac[1], fac[2], floor


# --------------------------------------------------------------------------------------------------------
# voxel <-> world affine helpers (fp64, host) -- monai/data/utils.py:737-982

AFFINE_TOL = 1e-3


def _np(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.array(a, dtype=np.float64, copy=True)


def affine_to_spacing(affine, r: int = 3, suppress_zeros: bool = True) -> np.ndarray:
    """Voxel spacing = column norms of the top-left r x r block (:737-761)."""
    a = _np(affine)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise ValueError(f"affine must be a square matrix, got {a.shape}.")
    block = a[:r, :r]
    spacing = np.sqrt(np.sum(block * block, axis=0))
    if suppress_zeros:
        spacing[spacing == 0] = 1.0
    return spacing


def to_affine_nd(r, affine) -> np.ndarray:
    """An (r+1)x(r+1) affine (or one shaped like the matrix `r`) filled from `affine`'s top-left block and last
    column (:938-982)."""
    a = _np(affine)
    if a.ndim != 2:
        raise ValueError(f"affine must have 2 dimensions, got {a.ndim}.")
    new = np.array(r, dtype=np.float64, copy=True)
    if new.ndim == 0:
        sr = int(new)
        if sr < 0:
            raise ValueError(f"r must be positive, got {sr}.")
        new = np.eye(sr + 1, dtype=np.float64)
    d = max(min(len(new) - 1, len(a) - 1), 1)
    new[:d, :d] = a[:d, :d]
    if d > 1:
        new[:d, -1] = a[:d, -1]
    return new


def zoom_affine(affine, scale, diagonal: bool = True) -> np.ndarray:
    """An affine whose column norms are `scale` (no translation): diagonal, or the original rotation times the new
    zooms with shears removed via a Cholesky factor of RZS^T RZS (:823-872)."""
    a = _np(affine)
    if len(a) != len(a[0]):
        raise ValueError(f"affine must be n x n, got {len(a)} x {len(a[0])}.")
    sc = np.array(scale, dtype=float, copy=True).reshape(-1)
    d = len(a) - 1
    norm = affine_to_spacing(a, r=d)
    if len(sc) < d:
        sc = np.append(sc, norm[len(sc):])
    sc = sc[:d]
    sc = np.asarray([s if (s and s > 0) else n for s, n in zip(sc, norm)], dtype=float)
    sc[sc == 0] = 1.0
    if diagonal:
        return np.diag(np.append(sc, [1.0]))
    rzs = a[:-1, :-1]
    zs = np.linalg.cholesky(rzs.T @ rzs).T
    rotation = rzs @ np.linalg.inv(zs)
    s = np.sign(np.diag(zs)) * np.abs(sc)
    out = np.eye(len(a))
    out[:-1, :-1] = rotation @ np.diag(s)
    return out


def compute_shape_offset(spatial_shape, in_affine, out_affine, scale_extent: bool = False):
    """Output shape that keeps the input field of view under `out_affine`, and the world offset that puts it in
    place (:875-935): map the input corners into output voxel space, round the extent, anchor at the corner that
    is minimal in every axis (or centre-align)."""
    shape = np.array(spatial_shape, copy=True, dtype=float)
    sr = len(shape)
    ia, oa = to_affine_nd(sr, in_affine), to_affine_nd(sr, out_affine)
    spans = [(-0.5, dim - 0.5) if scale_extent else (0.0, dim - 1.0) for dim in shape]
    corners = np.asarray(np.meshgrid(*spans, indexing="ij")).reshape((sr, -1))
    corners = np.concatenate((corners, np.ones_like(corners[:1])))
    try:
        corners_out = np.linalg.solve(oa, ia) @ corners
    except np.linalg.LinAlgError as e:
        raise ValueError(f"Affine {oa} is not invertible") from e
    corners = ia @ corners
    all_dist = corners_out[:-1].copy()
    corners_out = corners_out[:-1] / corners_out[-1]
    ptp = np.ptp(corners_out, axis=1)
    out_shape = np.round(ptp) if scale_extent else np.round(ptp + 1.0)
    offset = None
    for i in range(corners.shape[1]):
        min_corner = np.min(all_dist - all_dist[:, i:i + 1], 1)
        if np.allclose(min_corner, 0.0, rtol=AFFINE_TOL):
            offset = corners[:-1, i]
            break
    if offset is None:
        offset = ia[:-1, :-1] @ (shape / 2.0) + ia[:-1, -1] - oa[:-1, :-1] @ (out_shape / 2.0)
    if scale_extent:
        in_offset = np.append(0.5 * (shape / out_shape - 1.0), 1.0)
        offset = np.abs((ia @ in_offset / in_offset[-1])[:-1]) * np.sign(offset)
    return out_shape.astype(int, copy=False), offset


def iter_patch_position(image_size, patch_size, start_pos=(), overlap=0.0, padded: bool = False):
    """Upper-left corners of the patches of a regular grid, row-major (monai/data/utils.py:209-254): step = round(patch *
    (1 - overlap)) for a float overlap, patch - overlap for an int one; the last start is image - patch unless `padded`."""
    from itertools import product, starmap

    ndim = len(image_size)
    patch_size_ = get_valid_patch_size(image_size, patch_size)
    start_pos = tuple(start_pos) + (0,) * (ndim - len(tuple(start_pos)))
    start_pos = start_pos[:ndim]
    overlap = ensure_tuple_rep(overlap, ndim)
    if isinstance(overlap[0], float):
        steps = tuple(round(p * (1.0 - o)) for p, o in zip(patch_size_, overlap))
    else:
        steps = tuple(p - o for p, o in zip(patch_size_, overlap))
    end_pos = image_size if padded else tuple(s - round(p) + 1 for s, p in zip(image_size, patch_size_))
    return product(*starmap(range, zip(start_pos, end_pos, steps)))
