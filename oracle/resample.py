"""Oracle (test infrastructure): affine resampling restated with the ATen CPU operators the reference calls.

Reference followed (paths relative to /root/reference):
  * ``AffineTransform.forward``                monai/networks/layers/spatial_transforms.py:500-592
  * ``normalize_transform`` / ``to_norm_affine``  monai/networks/utils.py:243-326
  * eager branch of ``spatial_resample``       monai/transforms/spatial/functional.py:153-184
    (img.to(dtype) -> AffineTransform(normalized=False, reverse_indexing=True) -> float32)
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def normalize_transform(shape, dtype=torch.float64, align_corners=False, zero_centered=False) -> torch.Tensor:
    shape = torch.as_tensor(shape, dtype=torch.float64)
    norm = shape.clone()
    if align_corners:
        norm[norm <= 1.0] = 2.0
        norm = 2.0 / (norm if zero_centered else norm - 1.0)
        norm = torch.diag(torch.cat((norm, torch.ones(1, dtype=torch.float64))))
        if not zero_centered:
            norm[:-1, -1] = -1.0
    else:
        norm[norm <= 0.0] = 2.0
        norm = 2.0 / (norm - 1.0 if zero_centered else norm)
        norm = torch.diag(torch.cat((norm, torch.ones(1, dtype=torch.float64))))
        if not zero_centered:
            norm[:-1, -1] = 1.0 / shape - 1.0
    return norm.unsqueeze(0).to(dtype)


def to_norm_affine(affine, src_size, dst_size, align_corners=False, zero_centered=False) -> torch.Tensor:
    src_x = normalize_transform(src_size, affine.dtype, align_corners, zero_centered)
    dst_x = normalize_transform(dst_size, affine.dtype, align_corners, zero_centered)
    return src_x @ affine @ torch.as_tensor(np.linalg.inv(dst_x.numpy())).to(affine)


def affine_transform(src, theta, spatial_size=None, normalized=False, mode="bilinear", padding_mode="zeros", align_corners=True,
                     reverse_indexing=True, zero_centered=False) -> torch.Tensor:
    """src (N, C, spatial) CPU tensor; theta dxd / Nxdxd in src's dtype domain."""
    if theta.dim() == 2:
        theta = theta[None]
    theta = theta.clone()
    sr = src.dim() - 2
    if tuple(theta.shape[1:]) in ((2, 3), (3, 4)):
        pad = torch.tensor([0, 0, 1] if sr == 2 else [0, 0, 0, 1]).repeat(theta.shape[0], 1, 1).to(theta)
        theta = torch.cat([theta, pad], dim=1)
    src_size = tuple(src.shape)
    dst_size = src_size if spatial_size is None else src_size[:2] + tuple(spatial_size)
    if not normalized:
        theta = to_norm_affine(theta, src_size[2:], dst_size[2:], align_corners=False, zero_centered=zero_centered)
    if reverse_indexing:
        rev = torch.as_tensor(range(sr - 1, -1, -1))
        theta[:, :sr] = theta[:, rev]
        theta[:, :, :sr] = theta[:, :, rev]
    if theta.shape[0] == 1 and src_size[0] > 1:
        theta = theta.repeat(src_size[0], 1, 1)
    grid = F.affine_grid(theta=theta[:, :sr], size=list(dst_size), align_corners=align_corners)
    return F.grid_sample(input=src.contiguous(), grid=grid, mode=mode, padding_mode=padding_mode, align_corners=align_corners)


def spatial_resample_eager(img, xform, spatial_size, mode="bilinear", padding_mode="border", align_corners=False, dtype=torch.float64):
    """img (C, spatial) -> float32 (C, spatial_size): the non-compiled branch of functional.spatial_resample."""
    x = img.to(dtype)
    out = affine_transform(x.unsqueeze(0), torch.as_tensor(xform).to(x), spatial_size=spatial_size, normalized=False, mode=mode,
                           padding_mode=padding_mode, align_corners=align_corners, reverse_indexing=True).squeeze(0)
    return out.to(torch.float32)
