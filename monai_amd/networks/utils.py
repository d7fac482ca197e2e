"""Coordinate-normalisation helpers of the resampling path (host, fp64).

``normalize_transform`` / ``to_norm_affine`` give the same matrices as monai/networks/utils.py:243-326;
``index_matrix`` then folds what the reference does in three separate tensor ops -- normalised theta,
``F.affine_grid`` and the unnormalisation inside ``F.grid_sample`` -- into ONE voxel-space 3x4 matrix, which is all
the resampling kernel needs."""

from __future__ import annotations

from collections.abc import Sequence

import numpy as np

__all__ = ["normalize_transform", "to_norm_affine", "index_matrix"]


def normalize_transform(shape: Sequence[int], align_corners: bool = False, zero_centered: bool = False) -> np.ndarray:
    """(d+1)x(d+1) matrix taking voxel coordinates to [-1, 1]:
    align_corners False/True x zero_centered False/True normalise from [-0.5, d-0.5], [0, d-1], [-(d-1)/2, (d-1)/2],
    [-d/2, d/2] respectively."""
    size = np.asarray(shape, dtype=np.float64)
    norm = size.copy()
    d = len(norm)
    out = np.eye(d + 1, dtype=np.float64)
    if align_corners:
        norm[norm <= 1.0] = 2.0
        scale = 2.0 / (norm if zero_centered else norm - 1.0)
        out[np.arange(d), np.arange(d)] = scale
        if not zero_centered:
            out[:-1, -1] = -1.0
    else:
        norm[norm <= 0.0] = 2.0
        scale = 2.0 / (norm - 1.0 if zero_centered else norm)
        out[np.arange(d), np.arange(d)] = scale
        if not zero_centered:
            out[:-1, -1] = 1.0 / size - 1.0
    return out


def to_norm_affine(affine: np.ndarray, src_size, dst_size, align_corners: bool = False, zero_centered: bool = False) -> np.ndarray:
    """Voxel-space affine (dst voxel -> src voxel) expressed between the two normalised spaces."""
    affine = np.asarray(affine, dtype=np.float64)
    if affine.ndim != 2 or affine.shape[0] != affine.shape[1]:
        raise ValueError(f"affine must be dxd, got {tuple(affine.shape)}.")
    sr = affine.shape[0] - 1
    if sr != len(src_size) or sr != len(dst_size):
        raise ValueError(f"affine suggests {sr}D, got src={len(src_size)}D, dst={len(dst_size)}D.")
    src_x = normalize_transform(src_size, align_corners, zero_centered)
    dst_x = normalize_transform(dst_size, align_corners, zero_centered)
    return src_x @ affine @ np.linalg.inv(dst_x)


def _grid_axis(size: int, align_corners: bool):
    """F.affine_grid base coordinate of output index o along an axis: g = a*o + b."""
    if align_corners:
        return (2.0 / (size - 1), -1.0) if size > 1 else (0.0, 0.0)
    return 2.0 / size, 1.0 / size - 1.0


def _unnorm_axis(size: int, align_corners: bool):
    """F.grid_sample unnormalisation along an axis: i = a*g + b."""
    if align_corners:
        return (size - 1) / 2.0, (size - 1) / 2.0
    return size / 2.0, (size - 1) / 2.0


def index_matrix(theta: np.ndarray, src_size, dst_size, normalized: bool, align_corners: bool, reverse_indexing: bool,
                 zero_centered: bool = False) -> np.ndarray:
    """3x4 matrix M with  source index (z, y, x) = M @ (oz, oy, ox, 1)  for ``AffineTransform.forward(src, theta)``
    (monai/networks/layers/spatial_transforms.py:547-591), composed in fp64:

        normalised theta (to_norm_affine with align_corners=False, as hard-coded at :564-571; axis reversal :572-575)
        o F.affine_grid's base grid for `dst_size`  o  F.grid_sample's unnormalisation for `src_size`.

    2-D problems (sizes of length 2) are embedded with an identity z row."""
    theta = np.array(theta, dtype=np.float64, copy=True)
    sr = len(src_size)
    if theta.shape == (sr, sr + 1):
        theta = np.vstack([theta, np.eye(sr + 1)[-1:]])
    if theta.shape != (sr + 1, sr + 1):
        raise ValueError(f"theta must be {sr + 1}x{sr + 1}, got {theta.shape}.")
    if not normalized:
        theta = to_norm_affine(theta, src_size, dst_size, align_corners=False, zero_centered=zero_centered)
    if reverse_indexing:
        rev = list(range(sr - 1, -1, -1))
        theta[:sr] = theta[rev]
        theta[:, :sr] = theta[:, rev]
    # theta now acts on normalised (x, y[, z]) coordinates; sizes in that order are the reversed tensor sizes
    src_xyz, dst_xyz = list(src_size)[::-1], list(dst_size)[::-1]
    g = np.eye(sr + 1)
    u = np.eye(sr + 1)
    for ax in range(sr):
        g[ax, ax], g[ax, -1] = _grid_axis(int(dst_xyz[ax]), align_corners)
        u[ax, ax], u[ax, -1] = _unnorm_axis(int(src_xyz[ax]), align_corners)
    m_xyz = u @ theta @ g                      # output index (x, y, z, 1) -> source index (x, y, z)
    rev = list(range(sr - 1, -1, -1))
    m = m_xyz[:sr][rev][:, rev + [sr]]         # rows/cols back to tensor order (z, y, x)
    out = np.zeros((3, 4), dtype=np.float64)
    pad = 3 - sr
    for i in range(pad):
        out[i, i] = 1.0                        # identity on the embedded leading axes
    out[pad:, pad:3] = m[:, :sr]
    out[pad:, 3] = m[:, sr]
    return out
