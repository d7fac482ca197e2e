"""Golden vectors for the transform rows of SURVEY.md 8(a) (a13-a17), produced by the REAL reference.
Run only in the build container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_transforms.py"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

from monai.data import MetaTensor  # noqa: E402
from monai.networks.layers import AffineTransform, GaussianFilter  # noqa: E402
from monai.networks.layers.convutils import gaussian_1d  # noqa: E402
from monai.transforms import GaussianSmooth, Resample, Spacing, Spacingd  # noqa: E402


def rot_affine(seed, spacing):
    rs = np.random.RandomState(seed)
    q, _ = np.linalg.qr(rs.randn(3, 3))
    a = np.eye(4)
    a[:3, :3] = q @ np.diag(spacing)
    a[:3, 3] = rs.randn(3) * 5
    return a


def main():
    out = {}
    # ---- Spacing: the reference's own known-answer cases (tests/transforms/test_spacing.py:30-270, inputs restated)
    cases = [
        (dict(pixdim=(1.0, 1.5), padding_mode="zeros", dtype=float), torch.arange(4).reshape((1, 2, 2)) + 1.0, torch.eye(4), {}),
        (dict(pixdim=1.0, padding_mode="zeros", dtype=float), torch.ones((1, 2, 1, 2)), torch.eye(4), {}),
        (dict(pixdim=2.0, padding_mode="zeros", dtype=float), torch.arange(4).reshape((1, 2, 2)) + 1.0, torch.eye(4), {}),
        (dict(pixdim=(1.0, 0.2, 1.5), diagonal=False, padding_mode="zeros", align_corners=True), torch.ones((1, 2, 1, 2)),
         torch.tensor([[2, 1, 0, 4], [-1, -3, 0, 5], [0, 0, 2.0, 5], [0, 0, 0, 1]]), {}),
        (dict(pixdim=(3.0, 1.0), padding_mode="zeros"), torch.arange(24).reshape((2, 3, 4)), torch.as_tensor(np.diag([-3.0, 0.2, 1.5, 1])), {}),
        (dict(pixdim=(3.0, 1.0), padding_mode="zeros"), torch.arange(24).reshape((2, 3, 4)), torch.eye(4), {}),
        (dict(pixdim=(1.0, 1.0), align_corners=True), torch.arange(24).reshape((2, 3, 4)), torch.eye(4), {}),
        (dict(pixdim=(4.0, 5.0, 6.0)), torch.arange(24).reshape((1, 2, 3, 4)),
         torch.tensor([[-4, 0, 0, 4], [0, 5, 0, -5], [0, 0, 6, -6], [0, 0, 0, 1]]), {}),
        (dict(pixdim=(4.0, 5.0, 6.0), diagonal=True), torch.arange(24).reshape((1, 2, 3, 4)),
         torch.tensor([[-4, 0, 0, 4], [0, 5, 0, -5], [0, 0, 6, -6], [0, 0, 0, 1]]), {}),
        (dict(pixdim=(4.0, 5.0, 6.0), padding_mode="border", diagonal=True), torch.arange(24).reshape((1, 2, 3, 4)),
         torch.tensor([[-4, 0, 0, -4], [0, 5, 0, 0], [0, 0, 6, 0], [0, 0, 0, 1]]), {}),
        (dict(pixdim=(1.0, 2.0, 0.5), padding_mode="border", diagonal=True), torch.arange(24).reshape((1, 2, 3, 4)).float(),
         torch.eye(4), dict(mode="nearest")),
        (dict(pixdim=(1.9, 4.0), padding_mode="zeros", diagonal=True), torch.arange(24).reshape((1, 4, 6)).float(),
         torch.tensor([[-4, 0, 0, 4], [0, 5, 0, -5], [0, 0, 6, -6], [0, 0, 0, 1]]), dict(mode="nearest")),
        (dict(pixdim=(5.0, 3.0), padding_mode="border", diagonal=True, dtype=torch.float32), torch.arange(24).reshape((1, 4, 6)).float(),
         torch.tensor([[-4, 0, 0, 0], [0, 5, 0, 0], [0, 0, 6, 0], [0, 0, 0, 1]]), dict(mode="bilinear")),
        (dict(pixdim=(0.4, 0.7), padding_mode="reflection", diagonal=False), torch.arange(24).reshape((1, 4, 6)).float(),
         torch.eye(4), dict(mode="bilinear", align_corners=True)),
    ]
    for i, (init, data, affine, call) in enumerate(cases):
        y = Spacing(**{k: v for k, v in init.items()})(MetaTensor(data, affine=affine), **call)
        out[f"sp_{i}_out"] = y.numpy()
        out[f"sp_{i}_affine"] = y.affine.numpy()
    out["sp_n"] = np.asarray(len(cases))

    # ---- Spacing on seeded 3-D volumes: rotated affines, every mode / padding / align_corners / dtype
    k = 0
    for seed, shape, spacing, pixdim in [(1, (2, 20, 24, 18), (0.8, 0.8, 1.6), (1.0, 1.0, 1.0)), (2, (1, 17, 13, 22), (1.3, 0.7, 0.9), (0.9, 1.1, 0.6))]:
        torch.manual_seed(seed)
        data = torch.rand(shape)
        aff = rot_affine(seed, spacing)
        for mode in ("bilinear", "nearest"):
            for pad in ("zeros", "border", "reflection"):
                for ac in (False, True):
                    for dt in (np.float64, np.float32):
                        for diag in (False, True):
                            if diag and (ac or dt is np.float32):
                                continue
                            y = Spacing(pixdim=pixdim, diagonal=diag, mode=mode, padding_mode=pad, align_corners=ac, dtype=dt)(
                                MetaTensor(data, affine=aff))
                            out[f"sp3_{k}_cfg"] = np.asarray([seed, int(mode == "nearest"), ("zeros", "border", "reflection").index(pad), int(ac),
                                                              int(dt is np.float32), int(diag)])
                            out[f"sp3_{k}_shape"] = np.asarray(shape)
                            out[f"sp3_{k}_spacing"] = np.asarray(spacing)
                            out[f"sp3_{k}_pixdim"] = np.asarray(pixdim)
                            out[f"sp3_{k}_out"] = y.numpy()
                            out[f"sp3_{k}_affine"] = y.affine.numpy()
                            k += 1
    out["sp3_n"] = np.asarray(k)

    # ---- Spacingd on two keys (image bilinear / label nearest), config-4 style affine
    torch.manual_seed(5)
    img = MetaTensor(torch.rand(1, 30, 28, 20), affine=np.diag([0.8, 0.8, 1.6, 1.0]))
    lab = MetaTensor((torch.rand(1, 30, 28, 20) * 4).floor(), affine=np.diag([0.8, 0.8, 1.6, 1.0]))
    d = Spacingd(keys=("image", "label"), pixdim=(1.0, 1.0, 1.0), mode=("bilinear", "nearest"), padding_mode="border")({"image": img, "label": lab})
    out["spd_image"], out["spd_label"] = d["image"].numpy(), d["label"].numpy()
    out["spd_affine"] = d["image"].affine.numpy()

    # ---- AffineTransform called directly (all flag combinations on one random theta)
    torch.manual_seed(7)
    src = torch.rand(2, 2, 9, 11, 13)
    theta = torch.eye(4) + 0.15 * torch.randn(4, 4)
    theta[3] = torch.tensor([0, 0, 0, 1.0])
    theta[:3, 3] = torch.tensor([0.7, -1.2, 0.4])
    out["at_src"], out["at_theta"] = src.numpy(), theta.numpy()
    k = 0
    for normalized in (False, True):
        for rev in (True, False):
            for ac in (True, False):
                for pad in ("zeros", "border", "reflection"):
                    for mode in ("bilinear", "nearest"):
                        th = theta.clone()
                        if normalized:
                            th[:3, :3] = torch.eye(3) + 0.1 * (theta[:3, :3] - torch.eye(3))
                            th[:3, 3] = theta[:3, 3] * 0.1
                        y = AffineTransform(spatial_size=(7, 12, 10), normalized=normalized, mode=mode, padding_mode=pad, align_corners=ac,
                                            reverse_indexing=rev)(src, th)
                        out[f"at_{k}_cfg"] = np.asarray([int(normalized), int(rev), int(ac), ("zeros", "border", "reflection").index(pad), int(mode == "nearest")])
                        out[f"at_{k}_out"] = y.numpy()
                        k += 1
    y = AffineTransform(normalized=False, zero_centered=True, align_corners=False)(src, theta)
    out["at_zc_out"] = y.numpy()
    src2 = torch.rand(1, 3, 14, 9)
    th2 = torch.tensor([[0.9, 0.2, 1.0], [-0.1, 1.1, -0.5]])
    out["at2d_src"], out["at2d_theta"] = src2.numpy(), th2.numpy()
    out["at2d_out"] = AffineTransform(spatial_size=(10, 12), mode="bilinear", padding_mode="border", align_corners=False)(src2, th2).numpy()
    out["at_n"] = np.asarray(k)
    np.savez_compressed(os.path.join(HERE, "resample.npz"), **out)

    # ---- GaussianSmooth / gaussian_1d / GaussianFilter
    out = {}
    for i, (sigma, approx) in enumerate([(1.5, "erf"), (0.5, "erf"), ([1.5, 0.5], "erf"), (1.0, "erf"), (1.0, "sampled"), (2.0, "scalespace")]):
        nd = 2 if isinstance(sigma, list) or i < 3 else 3
        x = np.array([[[1, 1, 1], [2, 2, 2], [3, 3, 3]], [[4, 4, 4], [5, 5, 5], [6, 6, 6]]], dtype=np.float32) if nd == 2 else None
        if x is None:
            torch.manual_seed(30 + i)
            x = torch.rand(2, 12, 15, 17).numpy()
        y = GaussianSmooth(sigma=sigma, approx=approx)(torch.as_tensor(x))
        out[f"gs_{i}_in"], out[f"gs_{i}_out"] = x, np.asarray(y)
        out[f"gs_{i}_sigma"], out[f"gs_{i}_approx"] = np.asarray(sigma, dtype=np.float64), np.asarray(approx)
    out["gs_n"] = np.asarray(6)
    for i, (sigma, trunc, approx) in enumerate([(0.5, 8.0, "erf"), (1.0, 1.0, "erf"), (1.0, 4.0, "erf"), (2.5, 4.0, "erf"), (1.0, 4.0, "sampled"),
                                                (1.5, 4.0, "scalespace"), (0.1, 4.0, "erf")]):
        out[f"g1d_{i}_cfg"] = np.asarray([sigma, trunc])
        out[f"g1d_{i}_approx"] = np.asarray(approx)
        out[f"g1d_{i}_k"] = gaussian_1d(torch.tensor(sigma), truncated=trunc, approx=approx).numpy()
    out["g1d_n"] = np.asarray(7)
    torch.manual_seed(40)
    x = torch.rand(1, 3, 20, 18, 33)
    out["gf_in"] = x.numpy()
    out["gf_out"] = GaussianFilter(3, [1.0, 2.0, 0.7])(x).numpy()
    np.savez_compressed(os.path.join(HERE, "gaussian.npz"), **out)

    # ---- Resample with explicit grids (torch backend)
    out = {}
    torch.manual_seed(50)
    img = torch.rand(2, 8, 9, 10)
    grid = torch.stack(torch.meshgrid(torch.linspace(-5, 5, 6), torch.linspace(-6, 6, 7), torch.linspace(-7, 7, 5), indexing="ij")) \
        + 0.3 * torch.rand(3, 6, 7, 5)
    out["rs_img"], out["rs_grid"] = img.numpy(), grid.numpy()
    k = 0
    for norm_coords in (True, False):
        for ac in (False, True):
            for pad in ("zeros", "border", "reflection"):
                for mode in ("bilinear", "nearest"):
                    g = grid if norm_coords else grid + torch.tensor([3.5, 4.0, 4.5])[:, None, None, None]
                    y = Resample(mode=mode, padding_mode=pad, norm_coords=norm_coords, align_corners=ac, dtype=np.float64)(img, grid=g)
                    out[f"rs_{k}_cfg"] = np.asarray([int(norm_coords), int(ac), ("zeros", "border", "reflection").index(pad), int(mode == "nearest")])
                    out[f"rs_{k}_out"] = np.asarray(y)
                    k += 1
    out["rs_n"] = np.asarray(k)
    np.savez_compressed(os.path.join(HERE, "resample_grid.npz"), **out)
    print("transform golden vectors written")


if __name__ == "__main__":
    main()
