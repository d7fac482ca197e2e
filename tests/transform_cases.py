"""Parity cases for the resampling / smoothing transforms, shared by the emulator (CPU)
and GPU test modules.  Expected values are the REAL reference's outputs (tests/golden/*.npz,
tests/golden/make_golden_transforms.py); inputs are restated from the reference's own unit tests or seeded."""
from __future__ import annotations

import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PADS = ("zeros", "border", "reflection")
# fp64 interpolation then a float32 cast: the HIP path composes the coordinate as ONE fp64 matrix instead of
# normalise -> affine_grid -> unnormalise, so coordinates differ by ~1e-13 and results by <= a few float32 ulps.
TOL_F64 = 2e-6
# dtype=float32: the reference rounds the normalised grid to fp32 (coordinate error ~1e-5 voxel), this path keeps
# fp64 coordinates and only interpolates in fp32 -> differences up to ~1e-5 * local gradient.
TOL_F32 = 2e-4


def _rot_affine(seed, spacing):
    rs = np.random.RandomState(seed)
    q, _ = np.linalg.qr(rs.randn(3, 3))
    a = np.eye(4)
    a[:3, :3] = q @ np.diag(spacing)
    a[:3, 3] = rs.randn(3) * 5
    return a


def _check_nearest(got, exp, src, src_affine, dst_affine, align_corners, tag):
    """Nearest-neighbour results must equal the reference's wherever the sampling coordinate is not an exact .5 tie.
    At a tie both neighbours are equally near: the reference's pick is decided by the rounding noise of its
    normalise -> affine_grid -> unnormalise chain (1e-16 either way), this path rounds the exact coordinate
    half-to-even.  Mismatches are therefore allowed ONLY on tie voxels, and there `got` must be the other neighbour."""
    from monai_amd.networks.utils import index_matrix

    xform = np.linalg.solve(np.asarray(src_affine, dtype=np.float64), np.asarray(dst_affine, dtype=np.float64))
    m = index_matrix(xform, src.shape[1:], got.shape[1:], normalized=False, align_corners=align_corners, reverse_indexing=True)
    oz, oy, ox = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in got.shape[1:]], indexing="ij")
    coords = [m[r, 0] * oz + m[r, 1] * oy + m[r, 2] * ox + m[r, 3] for r in range(3)]
    tie = np.zeros(got.shape[1:], dtype=bool)
    for c in coords:
        tie |= np.abs(np.abs(c - np.floor(c)) - 0.5) < 1e-7
    bad = got != exp
    assert not (bad & ~tie[None]).any(), (tag, int((bad & ~tie[None]).sum()))
    assert bad.mean() < 0.25, (tag, bad.mean())


def _spacing_table_inputs():
    t = torch
    return [
        (dict(pixdim=(1.0, 1.5), padding_mode="zeros", dtype=float), t.arange(4).reshape((1, 2, 2)) + 1.0, t.eye(4), {}),
        (dict(pixdim=1.0, padding_mode="zeros", dtype=float), t.ones((1, 2, 1, 2)), t.eye(4), {}),
        (dict(pixdim=2.0, padding_mode="zeros", dtype=float), t.arange(4).reshape((1, 2, 2)) + 1.0, t.eye(4), {}),
        (dict(pixdim=(1.0, 0.2, 1.5), diagonal=False, padding_mode="zeros", align_corners=True), t.ones((1, 2, 1, 2)),
         t.tensor([[2, 1, 0, 4], [-1, -3, 0, 5], [0, 0, 2.0, 5], [0, 0, 0, 1]]), {}),
        (dict(pixdim=(3.0, 1.0), padding_mode="zeros"), t.arange(24).reshape((2, 3, 4)), t.as_tensor(np.diag([-3.0, 0.2, 1.5, 1])), {}),
        (dict(pixdim=(3.0, 1.0), padding_mode="zeros"), t.arange(24).reshape((2, 3, 4)), t.eye(4), {}),
        (dict(pixdim=(1.0, 1.0), align_corners=True), t.arange(24).reshape((2, 3, 4)), t.eye(4), {}),
        (dict(pixdim=(4.0, 5.0, 6.0)), t.arange(24).reshape((1, 2, 3, 4)), t.tensor([[-4, 0, 0, 4], [0, 5, 0, -5], [0, 0, 6, -6], [0, 0, 0, 1]]), {}),
        (dict(pixdim=(4.0, 5.0, 6.0), diagonal=True), t.arange(24).reshape((1, 2, 3, 4)),
         t.tensor([[-4, 0, 0, 4], [0, 5, 0, -5], [0, 0, 6, -6], [0, 0, 0, 1]]), {}),
        (dict(pixdim=(4.0, 5.0, 6.0), padding_mode="border", diagonal=True), t.arange(24).reshape((1, 2, 3, 4)),
         t.tensor([[-4, 0, 0, -4], [0, 5, 0, 0], [0, 0, 6, 0], [0, 0, 0, 1]]), {}),
        (dict(pixdim=(1.0, 2.0, 0.5), padding_mode="border", diagonal=True), t.arange(24).reshape((1, 2, 3, 4)).float(), t.eye(4), dict(mode="nearest")),
        (dict(pixdim=(1.9, 4.0), padding_mode="zeros", diagonal=True), t.arange(24).reshape((1, 4, 6)).float(),
         t.tensor([[-4, 0, 0, 4], [0, 5, 0, -5], [0, 0, 6, -6], [0, 0, 0, 1]]), dict(mode="nearest")),
        (dict(pixdim=(5.0, 3.0), padding_mode="border", diagonal=True, dtype=torch.float32), t.arange(24).reshape((1, 4, 6)).float(),
         t.tensor([[-4, 0, 0, 0], [0, 5, 0, 0], [0, 0, 6, 0], [0, 0, 0, 1]]), dict(mode="bilinear")),
        (dict(pixdim=(0.4, 0.7), padding_mode="reflection", diagonal=False), t.arange(24).reshape((1, 4, 6)).float(), t.eye(4),
         dict(mode="bilinear", align_corners=True)),
    ]


def case_spacing_reference_tables(device):
    """tests/transforms/test_spacing.py:30-270 (inputs restated); expected = the reference's own outputs."""
    from monai_amd.data import MetaTensor
    from monai_amd.transforms import Spacing

    g = np.load(os.path.join(GOLDEN, "resample.npz"))
    cases = _spacing_table_inputs()
    assert len(cases) == int(g["sp_n"])
    # first case doubles as the literal table of test_spacing.py:31-38
    for i, (init, data, affine, call) in enumerate(cases):
        y = Spacing(**init)(MetaTensor(data.to(device), affine=affine), **call)
        exp = g[f"sp_{i}_out"]
        assert tuple(y.shape) == exp.shape, (i, y.shape, exp.shape)
        assert y.dtype == torch.float32
        np.testing.assert_allclose(y.cpu().numpy(), exp, atol=2e-5, rtol=2e-6, err_msg=f"spacing table case {i}")
        np.testing.assert_allclose(y.affine.numpy(), g[f"sp_{i}_affine"], atol=1e-9)
    y = Spacing(pixdim=(1.0, 1.5), padding_mode="zeros", dtype=float)(MetaTensor((torch.arange(4).reshape((1, 2, 2)) + 1.0).to(device), affine=torch.eye(4)))
    np.testing.assert_allclose(y.cpu().numpy(), np.array([[[1.0, 1.0], [3.0, 2.0]]]), atol=1e-6)


def case_spacing_3d(device, limit=None):
    from monai_amd.data import MetaTensor
    from monai_amd.transforms import Spacing

    g = np.load(os.path.join(GOLDEN, "resample.npz"))
    n = int(g["sp3_n"])
    worst = 0.0
    for k in range(n if limit is None else min(n, limit)):
        seed, nearest, pad, ac, f32, diag = (int(v) for v in g[f"sp3_{k}_cfg"])
        shape = tuple(int(v) for v in g[f"sp3_{k}_shape"])
        torch.manual_seed(seed)
        data = torch.rand(shape)
        aff = _rot_affine(seed, tuple(g[f"sp3_{k}_spacing"]))
        y = Spacing(pixdim=tuple(g[f"sp3_{k}_pixdim"]), diagonal=bool(diag), mode="nearest" if nearest else "bilinear", padding_mode=PADS[pad],
                    align_corners=bool(ac), dtype=np.float32 if f32 else np.float64)(MetaTensor(data.to(device), affine=aff))
        exp = g[f"sp3_{k}_out"]
        assert tuple(y.shape) == exp.shape, (k, y.shape, exp.shape)
        np.testing.assert_allclose(y.affine.numpy(), g[f"sp3_{k}_affine"], atol=1e-9)
        got = y.cpu().numpy()
        if nearest:
            _check_nearest(got, exp, data.numpy(), aff, g[f"sp3_{k}_affine"], bool(ac), k)
        else:
            err = np.abs(got - exp).max()
            worst = max(worst, err)
            assert err < (TOL_F32 if f32 else TOL_F64), (k, err)
    return worst


def case_spacingd(device):
    from monai_amd.data import MetaTensor
    from monai_amd.transforms import Spacingd

    g = np.load(os.path.join(GOLDEN, "resample.npz"))
    torch.manual_seed(5)
    a = np.diag([0.8, 0.8, 1.6, 1.0])
    img = MetaTensor(torch.rand(1, 30, 28, 20).to(device), affine=a)
    lab = MetaTensor((torch.rand(1, 30, 28, 20) * 4).floor().to(device), affine=a)
    d = Spacingd(keys=("image", "label"), pixdim=(1.0, 1.0, 1.0), mode=("bilinear", "nearest"), padding_mode="border")({"image": img, "label": lab})
    assert tuple(d["image"].shape) == g["spd_image"].shape == tuple(d["label"].shape)
    assert np.abs(d["image"].cpu().numpy() - g["spd_image"]).max() < TOL_F64
    _check_nearest(d["label"].cpu().numpy(), g["spd_label"], lab.as_tensor().cpu().numpy(), a, g["spd_affine"], False, "spacingd label")
    np.testing.assert_allclose(d["image"].affine.numpy(), g["spd_affine"], atol=1e-9)
    inv = Spacingd(keys=("image",), pixdim=(1.0, 1.0, 1.0), padding_mode="border").inverse({"image": d["image"]})
    assert tuple(inv["image"].shape) == (1, 30, 28, 20)


def case_affine_transform(device):
    from monai_amd.networks.layers import AffineTransform

    g = np.load(os.path.join(GOLDEN, "resample.npz"))
    src, theta = torch.from_numpy(g["at_src"]), torch.from_numpy(g["at_theta"])
    for k in range(int(g["at_n"])):
        normalized, rev, ac, pad, nearest = (int(v) for v in g[f"at_{k}_cfg"])
        th = theta.clone()
        if normalized:
            th[:3, :3] = torch.eye(3) + 0.1 * (theta[:3, :3] - torch.eye(3))
            th[:3, 3] = theta[:3, 3] * 0.1
        y = AffineTransform(spatial_size=(7, 12, 10), normalized=bool(normalized), mode="nearest" if nearest else "bilinear",
                            padding_mode=PADS[pad], align_corners=bool(ac), reverse_indexing=bool(rev))(src.to(device), th)
        exp = g[f"at_{k}_out"]
        got = y.cpu().numpy()
        if nearest:
            assert (got == exp).mean() > 0.995, (k, (got == exp).mean())
        else:
            assert np.abs(got - exp).max() < TOL_F32, (k, np.abs(got - exp).max())  # fp32 src -> fp32 pipeline in the reference
    y = AffineTransform(normalized=False, zero_centered=True, align_corners=False)(src.to(device), theta)
    assert np.abs(y.cpu().numpy() - g["at_zc_out"]).max() < TOL_F32
    y = AffineTransform(spatial_size=(10, 12), mode="bilinear", padding_mode="border", align_corners=False)(
        torch.from_numpy(g["at2d_src"]).to(device), torch.from_numpy(g["at2d_theta"]))
    assert np.abs(y.cpu().numpy() - g["at2d_out"]).max() < TOL_F32
    # float64 source: fp64 interpolation, result in fp64 (rounded through fp32 storage)
    y64 = AffineTransform(spatial_size=(7, 12, 10), padding_mode="border", align_corners=False)(src.double().to(device), theta.double())
    assert y64.dtype == torch.float64


# ------------------------------------------------------------------------------------------ grid_pull / Resample
GP_BOUNDS = {"replicate": 0, "dct1": 1, "dct2": 2, "dst1": 3, "dst2": 4, "dft": 5, "zero": 7}


def _gp_inputs(seed, f64):
    shapes = {1: ((2, 2, 7, 6, 5), (4, 5, 6)), 2: ((1, 3, 9, 8), (7, 6)), 3: ((1, 2, 11), (13,))}
    ishape, oshape = shapes[seed]
    torch.manual_seed(seed)
    sd = len(ishape) - 2
    inp32 = torch.randn(ishape, dtype=torch.float32)
    grid32 = (torch.rand((ishape[0],) + oshape + (sd,), dtype=torch.float32) * 3.0 - 1.0) * torch.tensor(ishape[2:], dtype=torch.float32)
    if not f64:
        return inp32, grid32
    inp64 = torch.randn(ishape, dtype=torch.float64)
    grid64 = (torch.rand((ishape[0],) + oshape + (sd,), dtype=torch.float64) * 3.0 - 1.0) * torch.tensor(ishape[2:], dtype=torch.float64)
    return inp64, grid64


def case_grid_pull_vs_reference_build(device):
    """monai_amd._C.grid_pull against outputs of the REFERENCE's compiled CPU resampler (oracle/_ref), 1-D/2-D/3-D,
    fp32 and fp64, all seven boundary conditions, orders 0 and 1, extrapolate on/off."""
    from monai_amd import _C

    g = np.load(os.path.join(GOLDEN, "grid_pull.npz"))
    cache = {}
    worst = 0.0
    for k in range(int(g["gp_n"])):
        seed, f64, b, interp, extrap = (int(v) for v in g[f"gp_{k}_cfg"])
        if (seed, f64) not in cache:
            # the generator draws fp32 first, then fp64, from one stream per seed
            cache[(seed, 0)] = _gp_inputs(seed, False)
            cache[(seed, 1)] = _gp_inputs(seed, True)
        inp, grid = cache[(seed, f64)]
        y = _C.grid_pull(inp.to(device), grid.to(device), [_C.BoundType(b)], [_C.InterpolationType(interp)], bool(extrap))
        exp = g[f"gp_{k}_out"]
        assert tuple(y.shape) == exp.shape
        err = np.abs(y.cpu().numpy() - exp).max()
        worst = max(worst, err)
        assert err < (1e-12 if f64 else 2e-5), (k, seed, f64, b, interp, extrap, err)
    inp, grid = cache[(1, 0)]
    y = _C.grid_pull(inp.to(device), grid.to(device), [_C.BoundType(2), _C.BoundType(7), _C.BoundType(5)], [_C.InterpolationType(1)], True)
    assert np.abs(y.cpu().numpy() - g["gp_mixed_out"]).max() < 2e-5   # per-axis boundary conditions
    # half precision (the reference's GPU build takes it, pushpull_cuda.cu:2195): evaluated in fp32, rounded once -- equal to the half-rounded fp32 result
    # of the half-rounded inputs; push / count / grad go the same way
    hi, hg = inp.half(), grid.half()
    yh = _C.grid_pull(hi.to(device), hg.to(device), [_C.BoundType(2)], [_C.InterpolationType(1)], True)
    assert yh.dtype == torch.float16
    yf = _C.grid_pull(hi.float().to(device), hg.float().to(device), [_C.BoundType(2)], [_C.InterpolationType(1)], True)
    assert torch.equal(yh.cpu(), yf.cpu().half())
    ph = _C.grid_push(yh, hg.to(device), list(inp.shape[2:]), [_C.BoundType(2)], [_C.InterpolationType(1)], True)
    assert ph.dtype == torch.float16 and tuple(ph.shape) == tuple(inp.shape)
    return worst


# ---- the whole monai._C surface (pull / push / count / grad + backward), orders 0-7 ---------------------------------
def _pp_cases():
    cases = []
    for sd in (1, 2, 3):
        mixed = {1: [2], 2: [1, 3], 3: [1, 3, 2]}[sd]
        for f64 in (1, 0):
            orders = [[0], [1], [2], [3], [4], [5], [6], [7], mixed] if f64 else [[1], [3]]
            for b in (0, 1, 2, 3, 4, 5, 7):
                for o in orders:
                    for extrap in ((1, 0) if b in (0, 7) and len(o) == 1 and o[0] in (1, 3) else (1,)):
                        cases.append({"sd": sd, "f64": f64, "bound": [b], "order": o, "extrapolate": extrap, "seed": len(cases)})
    # per-axis boundary conditions
    cases.append({"sd": 3, "f64": 1, "bound": [2, 7, 5], "order": [1], "extrapolate": 1, "seed": len(cases)})
    cases.append({"sd": 3, "f64": 1, "bound": [3, 1, 4], "order": [2, 3, 3], "extrapolate": 1, "seed": len(cases)})
    return cases


PP_CASES = _pp_cases()


def pp_inputs(case):
    """Seeded inputs of one case.  1-D problems use a single image and channel: the reference's 1-D spatial-gradient
    code zeroes the element FOLLOWING each output (`out_ptr_NCX[out_sK] = 0` with out_sK = 1 on an (N,C,X,1) tensor,
    pushpull_cpu.cpp:899-900 and :1362), i.e. the next channel's first value, so only B = C = 1 is well defined there."""
    sd = case["sd"]
    dt = torch.float64 if case["f64"] else torch.float32
    gen = torch.Generator().manual_seed(1000 + case["seed"])
    b, c = (1, 1) if sd == 1 else (2, 2)
    isp, osp = (5, 6, 4)[:sd], (4, 3, 3)[:sd]
    r = lambda *shape: torch.randn(*shape, generator=gen, dtype=dt)  # noqa: E731
    inp = r(b, c, *isp)
    # coordinates from one extent below to one extent above the field of view: every boundary rule is exercised
    grid = (torch.rand(b, *osp, sd, generator=gen, dtype=dt) * 3.0 - 1.0) * torch.tensor(isp, dtype=dt)
    # the sgrad cotangent is a strided view (component stride 2): the reference only evaluates the B-spline second
    # derivatives when that stride is > 1 (`trgt_sK > 1`, pushpull_cpu.cpp:1003) and reads uninitialised memory otherwise
    return {"inp": inp, "grid": grid, "splat": r(b, c, *osp), "g_pull": r(b, c, *osp), "g_push": r(b, c, *isp),
            "g_count": r(b, 1, *isp), "g_sgrad": r(b, c, *osp, sd, 2)[..., 0], "isp": isp}


def pp_run(mod, case, x, device):
    """Every entry point of `mod` (monai_amd._C or the reference build) on one case -> dict of CPU tensors."""
    bd = [mod.BoundType(v) for v in case["bound"]]
    it = [mod.InterpolationType(v) for v in case["order"]]
    ex = bool(case["extrapolate"])
    t = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in x.items()}
    inp_g, grid_g, splat_g = t["inp"].clone().requires_grad_(True), t["grid"].clone().requires_grad_(True), t["splat"].clone().requires_grad_(True)
    res = {
        "pull": mod.grid_pull(t["inp"], t["grid"], bd, it, ex),
        "sgrad": mod.grid_grad(t["inp"], t["grid"], bd, it, ex),
        "push": mod.grid_push(t["splat"], t["grid"], list(x["isp"]), bd, it, ex),
        "count": mod.grid_count(t["grid"], list(x["isp"]), bd, it, ex),
        "count_bwd": mod.grid_count_backward(t["g_count"], grid_g, bd, it, ex),
    }
    a = mod.grid_pull_backward(t["g_pull"], inp_g, grid_g, bd, it, ex)
    res["pull_bwd_input"], res["pull_bwd_grid"] = a[0], a[1]
    a = mod.grid_push_backward(t["g_push"], splat_g, grid_g, bd, it, ex)
    res["push_bwd_input"], res["push_bwd_grid"] = a[0], a[1]
    a = mod.grid_grad_backward(t["g_sgrad"], inp_g, grid_g, bd, it, ex)
    res["sgrad_bwd_input"], res["sgrad_bwd_grid"] = a[0], a[1]
    return {k: v.detach().cpu() for k, v in res.items()}


PP_SCATTER = ("push", "count", "pull_bwd_input", "sgrad_bwd_input")      # atomics: summation order is free


def case_pushpull_vs_reference_build(device):
    """Every monai_amd._C entry point against outputs of the REFERENCE's compiled CPU resampler (oracle/_ref): 1-D / 2-D /
    3-D, fp64 (orders 0-7 and mixed per-axis orders) and fp32 (orders 1 and 3), all seven boundary conditions.
    Gathering outputs (pull, spatial gradients, every gradient with respect to the grid) are BIT-EXACT; scattering
    outputs (push, count and the d/d input results that are pushes) add with atomics in free order: <= 4 ulp-ish."""
    from monai_amd import _C

    g = np.load(os.path.join(GOLDEN, "pushpull.npz"))
    assert int(g["pp_n"]) == len(PP_CASES)
    worst = 0.0
    for k, case in enumerate(PP_CASES):
        got = pp_run(_C, case, pp_inputs(case), device)
        for name, y in got.items():
            exp = g[f"pp_{k}_{name}"]
            assert tuple(y.shape) == exp.shape, (k, name, tuple(y.shape), exp.shape)
            y = y.numpy()
            if name in PP_SCATTER:
                tol = (1e-13 if case["f64"] else 2e-6) * max(1.0, float(np.abs(exp).max()))
                err = float(np.abs(y - exp).max())
                worst = max(worst, err)
                assert err <= tol, (k, case, name, err)
            else:
                assert np.array_equal(y, exp), (k, case, name, float(np.abs(y - exp).max()))
    return worst


def case_pushpull_tiny_extents_wide_coordinates(device):
    """Extents 1-4 per axis and coordinates from -4n to +4n: every branch of every boundary rule, including the taps whose
    sign is 0 (dst1's index -1, `zero`'s outside indices) -- the gathers load unconditionally, so those indices must stay
    inside the allocation.  Checked against the reference build when oracle/_ref travelled, else for finiteness only."""
    from monai_amd import _C
    from oracle import build_ref

    ref = build_ref.load()
    gen = torch.Generator().manual_seed(77)
    n_checked = 0
    for isp in ((1, 2, 3), (2, 1, 4), (3, 3, 1), (4, 2, 2)):
        inp = torch.randn(1, 2, *isp, generator=gen, dtype=torch.float64)
        grid = (torch.rand(1, 6, 5, 7, 3, generator=gen, dtype=torch.float64) * 8.0 - 4.0) * torch.tensor(isp, dtype=torch.float64)
        for b in (0, 1, 2, 3, 4, 5, 7):
            for o in (0, 1, 2, 3):
                got = _C.grid_pull(inp.to(device), grid.to(device), [_C.BoundType(b)], [_C.InterpolationType(o)], True).cpu()
                sg = _C.grid_grad(inp.to(device), grid.to(device), [_C.BoundType(b)], [_C.InterpolationType(o)], True).cpu()
                assert torch.isfinite(got).all() and torch.isfinite(sg).all()
                if ref is not None:
                    exp = ref.grid_pull(inp, grid, [ref.BoundType(b)], [ref.InterpolationType(o)], True)
                    assert torch.equal(got, exp), (isp, b, o, float((got - exp).abs().max()))
                n_checked += 1
    return n_checked


def case_grid_pull_reference_rows_all_orders(device):
    """The reference's own golden rows (tests/testing_data/1D_BP_fwd.txt and 1D_BP_bwd.txt, used by
    tests/networks/layers/test_grid_pull.py:35-100): input arange(10), grid arange(20) + 0.5, every interpolation order
    (0-7) x boundary condition, forward values and the gradients of result.sum() for the four combinations of
    (input.requires_grad, grid.requires_grad), through the differentiable `grid_pull` wrapper; rtol = atol = 1e-4 as there."""
    from monai_amd.networks.layers import grid_pull

    g = np.load(os.path.join(GOLDEN, "pushpull.npz"))
    rows = [str(v) for v in g["rows"]]
    assert len(rows) == 56
    for key in rows:
        interp, bound = key.split("_", 1)
        j = 0
        for input_g in (True, False):
            for grid_g in (True, False):
                inp = torch.arange(10, dtype=torch.float32, device=device).reshape(1, 1, 10).requires_grad_(input_g)
                grid = (torch.arange(20, dtype=torch.float32, device=device).reshape(1, 20, 1) + 0.5).requires_grad_(grid_g)
                y = grid_pull(inp, grid, interpolation=interp, bound=bound)
                np.testing.assert_allclose(y.detach().cpu().numpy().reshape(-1), g[f"fwd_{key}"], rtol=1e-4, atol=1e-4, err_msg=key)
                grads = []
                if input_g or grid_g:
                    y.sum().backward()
                if input_g:
                    grads.append(inp.grad.reshape(-1))
                if grid_g:
                    grads.append(grid.grad.reshape(-1))
                got = torch.cat(grads).cpu().numpy() if grads else np.zeros(1)
                np.testing.assert_allclose(got, g[f"bwd_{key}_{j}"].reshape(-1), rtol=1e-4, atol=1e-4, err_msg=f"{key} {input_g} {grid_g}")
                j += 1


def case_grid_functions_autograd(device):
    """The differentiable wrappers agree with finite differences of themselves (fp64, cubic, dct2): d/d input and d/d grid
    of grid_pull, grid_push and grid_grad; d/d grid of grid_count."""
    from monai_amd.networks.layers import grid_count, grid_grad, grid_pull, grid_push

    gen = torch.Generator().manual_seed(5)
    inp = torch.randn(1, 2, 5, 6, generator=gen, dtype=torch.float64).to(device)
    grid = ((torch.rand(1, 4, 3, 2, generator=gen, dtype=torch.float64) * 0.8 + 0.1) * torch.tensor([4.0, 5.0], dtype=torch.float64)).to(device)
    splat = torch.randn(1, 2, 4, 3, generator=gen, dtype=torch.float64).to(device)
    kw = {"interpolation": "cubic", "bound": "dct2"}

    def fd(fn, x, eps=1e-6):
        base = fn(x)
        w = torch.randn(base.shape, generator=torch.Generator().manual_seed(7), dtype=torch.float64).to(device)
        num = torch.zeros_like(x)
        flat = x.reshape(-1)
        for i in range(flat.numel()):
            xp, xm = flat.clone(), flat.clone()
            xp[i] += eps
            xm[i] -= eps
            num.reshape(-1)[i] = ((fn(xp.reshape(x.shape)) - fn(xm.reshape(x.shape))) * w).sum() / (2 * eps)
        xa = x.clone().requires_grad_(True)
        (fn(xa) * w).sum().backward()
        return float((xa.grad - num).abs().max())

    errs = {
        "pull/input": fd(lambda v: grid_pull(v, grid, **kw), inp),
        "pull/grid": fd(lambda v: grid_pull(inp, v, **kw), grid),
        "push/input": fd(lambda v: grid_push(v, grid, (5, 6), **kw), splat),
        "push/grid": fd(lambda v: grid_push(splat, v, (5, 6), **kw), grid),
        "count/grid": fd(lambda v: grid_count(v, (5, 6), **kw), grid),
        "grad/input": fd(lambda v: grid_grad(v, grid, **kw), inp),
        "grad/grid": fd(lambda v: grid_grad(inp, v, **kw), grid),
    }
    for k, e in errs.items():
        assert e < 1e-6, (k, e)
    return errs


# ---- Resample, USE_COMPILED branch ------------------------------------------------------------------------------------
def _rc_cases():
    cases = []
    for sd in (3, 2):
        for mode in ("bilinear", "nearest", "bicubic"):
            for pad in ("zeros", "border", "reflection"):
                for ac in (False, True):
                    cases.append({"sd": sd, "mode": mode, "padding_mode": pad, "align_corners": ac, "norm_coords": True,
                                  "dtype": np.float64, "seed": len(cases)})
    cases.append({"sd": 3, "mode": "bilinear", "padding_mode": "border", "align_corners": False, "norm_coords": False, "dtype": np.float64, "seed": 90})
    cases.append({"sd": 3, "mode": "bilinear", "padding_mode": "zeros", "align_corners": True, "norm_coords": False, "dtype": np.float32, "seed": 91})
    cases.append({"sd": 3, "mode": "bicubic", "padding_mode": "border", "align_corners": False, "norm_coords": True, "dtype": np.float32, "seed": 92})
    return cases


RC_CASES = _rc_cases()


def rc_inputs(case):
    sd = case["sd"]
    gen = torch.Generator().manual_seed(500 + case["seed"])
    isp, osp = (7, 8, 9)[:sd], (5, 6, 4)[:sd]
    img = torch.rand(2, *isp, generator=gen, dtype=torch.float32)
    half = torch.tensor([(n - 1) / 2.0 for n in isp], dtype=torch.float64).reshape((sd,) + (1,) * sd)
    u = torch.rand(sd, *osp, generator=gen, dtype=torch.float64) * 2.6 - 1.3          # beyond the field of view on both sides
    grid = u * half if case["norm_coords"] else u                                      # voxel units centred on the image | [-1, 1]
    return img, grid


def case_resample_compiled_vs_reference(device):
    """``Resample`` with ``USE_COMPILED`` (array.py:2076-2092) against the REAL reference transform run with its own
    native module (tests/golden/make_golden_resample_compiled.py).  The native sampler is bit-exact and the coordinate
    arithmetic is the same fp64 / grid-dtype expression; outputs are float32."""
    from monai_amd import config
    from monai_amd.transforms import Resample

    g = np.load(os.path.join(GOLDEN, "resample_compiled.npz"))
    assert int(g["rc_n"]) == len(RC_CASES)
    old = config.USE_COMPILED
    config.USE_COMPILED = True
    worst = 0.0
    try:
        for k, case in enumerate(RC_CASES):
            img, grid = rc_inputs(case)
            tr = Resample(mode=case["mode"], padding_mode=case["padding_mode"], norm_coords=case["norm_coords"],
                          align_corners=case["align_corners"], dtype=case["dtype"])
            y = tr(img.to(device), grid.to(device)).cpu().numpy()
            exp = g[f"rc_{k}"]
            assert y.shape == exp.shape and y.dtype == np.float32, (k, y.shape, exp.shape, y.dtype)
            err = float(np.abs(y - exp).max())
            worst = max(worst, err)
            assert err <= 1e-6, (k, case, err)
    finally:
        config.USE_COMPILED = old
    return worst


# ---- Warp / DVF2DDF ---------------------------------------------------------------------------------------------------
def _warp_cases():
    cases = []
    for compiled in (1, 0):
        for sd in (3, 2):
            for mode in (("bilinear", "nearest", "bicubic") if compiled else ("bilinear", "nearest")):
                for pad in ("zeros", "border", "reflection"):
                    cases.append({"compiled": compiled, "sd": sd, "mode": mode, "padding_mode": pad, "dvf": 0, "seed": len(cases)})
        cases.append({"compiled": compiled, "sd": 3, "mode": "bilinear", "padding_mode": "zeros", "dvf": 1, "seed": len(cases)})
    return cases


WARP_CASES = _warp_cases()


def warp_inputs(case):
    sd = case["sd"]
    gen = torch.Generator().manual_seed(700 + case["seed"])
    sp = (6, 7, 8)[:sd]
    image = torch.rand(2, 3, *sp, generator=gen, dtype=torch.float32)
    ddf = (torch.rand(2, sd, *sp, generator=gen, dtype=torch.float32) - 0.5) * (3.0 if case["dvf"] else 6.0)
    return image, ddf


def case_warp_vs_reference(device):
    """``Warp`` / ``DVF2DDF`` against the reference blocks (tests/golden/make_golden_resample_compiled.py): with USE_COMPILED
    (native grid_pull: bit-exact sampler) and without (the reference normalises the grid to [-1, 1] in fp32 and lets
    grid_sample unnormalise it; here the kernel samples at the voxel coordinates directly: <= 2e-5)."""
    import warnings

    from monai_amd import config
    from monai_amd.networks.blocks import DVF2DDF, Warp

    g = np.load(os.path.join(GOLDEN, "resample_compiled.npz"))
    assert int(g["warp_n"]) == len(WARP_CASES)
    old = config.USE_COMPILED
    worst = {0: 0.0, 1: 0.0}
    try:
        for k, case in enumerate(WARP_CASES):
            image, ddf = warp_inputs(case)
            config.USE_COMPILED = bool(case["compiled"])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                layer = DVF2DDF(num_steps=3, mode=case["mode"], padding_mode=case["padding_mode"]) if case["dvf"] else Warp(mode=case["mode"], padding_mode=case["padding_mode"])
            y = layer(ddf.to(device)) if case["dvf"] else layer(image.to(device), ddf.to(device))
            exp = g[f"warp_{k}"]
            y = y.cpu().numpy()
            assert y.shape == exp.shape, (k, y.shape, exp.shape)
            err = float(np.abs(y - exp).max())
            worst[case["compiled"]] = max(worst[case["compiled"]], err)
            if case["mode"] == "nearest" and not case["compiled"]:
                # exact .5 ties can round the other way after the reference's normalise / unnormalise round trip
                assert float((np.abs(y - exp) > 2e-5).mean()) < 0.02, (k, case)
            else:
                assert err <= (0.0 if case["compiled"] else 2e-5), (k, case, err)
    finally:
        config.USE_COMPILED = old
    return worst


def case_grid_pull_reference_golden_rows(device):
    """tests/testing_data/1D_BP_fwd.txt (read by tests/testing_data/cpp_resample_answers.py:19-42, used at
    tests/networks/layers/test_grid_pull.py:35-100): input arange(10), grid arange(20)+0.5; rows for orders 0 and 1."""
    from monai_amd import _C

    g = np.load(os.path.join(GOLDEN, "grid_pull.npz"))
    inp = torch.arange(10, dtype=torch.float32).reshape(1, 1, 10).to(device)
    grid = (torch.arange(20, dtype=torch.float32) + 0.5).reshape(1, 20, 1).to(device)
    n = 0
    for key in g["bp1d_keys"]:
        interp, bound = str(key).split("_")
        y = _C.grid_pull(inp, grid, [_C.BoundType.__members__[bound]], [_C.InterpolationType.__members__[interp]], True)
        np.testing.assert_allclose(y.cpu().numpy().reshape(-1), g[f"bp1d_{key}"], atol=1e-4, err_msg=str(key))
        n += 1
    assert n == 14
    assert _C.BoundType.__members__["reflect"] == _C.BoundType.dct2 and _C.BoundType.__members__["zeros"] == _C.BoundType.zero


def case_resample_dense_grid(device):
    from monai_amd.transforms import Resample

    g = np.load(os.path.join(GOLDEN, "resample_grid.npz"))
    img, grid = torch.from_numpy(g["rs_img"]), torch.from_numpy(g["rs_grid"])
    for k in range(int(g["rs_n"])):
        norm_coords, ac, pad, nearest = (int(v) for v in g[f"rs_{k}_cfg"])
        gg = grid if norm_coords else grid + torch.tensor([3.5, 4.0, 4.5])[:, None, None, None]
        y = Resample(mode="nearest" if nearest else "bilinear", padding_mode=PADS[pad], norm_coords=bool(norm_coords), align_corners=bool(ac),
                     dtype=np.float64)(img.to(device), grid=gg.to(device))
        exp = g[f"rs_{k}_out"]
        got = y.cpu().numpy()
        assert got.shape == exp.shape and got.dtype == np.float32
        if nearest:
            assert (got == exp).mean() > 0.99, (k, (got == exp).mean())
        else:
            assert np.abs(got - exp).max() < 5e-6, (k, np.abs(got - exp).max())


# ------------------------------------------------------------------------------------------ Gaussian smoothing
def case_gaussian_1d_tables():
    """Host taps vs the reference's gaussian_1d (convutils.py:78-131): erf / sampled bit-exact, scalespace to 1e-6."""
    from monai_amd.networks.layers import gaussian_1d

    g = np.load(os.path.join(GOLDEN, "gaussian.npz"))
    for i in range(int(g["g1d_n"])):
        sigma, trunc = (float(v) for v in g[f"g1d_{i}_cfg"])
        approx = str(g[f"g1d_{i}_approx"])
        k = gaussian_1d(torch.tensor(sigma), truncated=trunc, approx=approx).numpy()
        exp = g[f"g1d_{i}_k"]
        assert k.shape == exp.shape, (i, k.shape, exp.shape)
        if approx == "scalespace":
            np.testing.assert_allclose(k, exp, rtol=2e-6, atol=1e-9)
        else:
            assert np.array_equal(k, exp), (i, approx)
    # tests/networks/layers/test_gaussian.py: gaussian_1d(0.5, 8) is the 9-tap erf table, gaussian_1d(1, 1) three taps
    k = gaussian_1d(0.5, 8.0).numpy()
    assert k.shape == (9,) and abs(k[4] - 0.6826895) < 1e-6 and abs(k[3] - 0.1573054) < 1e-6


def case_gaussian_smooth(device):
    """GaussianSmooth known-answer tables of tests/transforms/test_gaussian_smooth.py:24-92 (2x3x3, sigma 1.5 / 0.5 /
    [1.5, 0.5]) and seeded 3-D volumes for erf / sampled / scalespace; expected = the reference's outputs."""
    from monai_amd.networks.layers import GaussianFilter
    from monai_amd.transforms import GaussianSmooth

    g = np.load(os.path.join(GOLDEN, "gaussian.npz"))
    for i in range(int(g["gs_n"])):
        sigma = g[f"gs_{i}_sigma"]
        sigma = float(sigma) if sigma.ndim == 0 else [float(s) for s in sigma]
        y = GaussianSmooth(sigma=sigma, approx=str(g[f"gs_{i}_approx"]))(torch.from_numpy(g[f"gs_{i}_in"]).to(device))
        np.testing.assert_allclose(y.cpu().numpy(), g[f"gs_{i}_out"], atol=1e-5, rtol=1e-5, err_msg=f"gaussian smooth case {i}")
    # first table literally (test_gaussian_smooth.py:24-46), atol of the reference test
    x = np.array([[[1, 1, 1], [2, 2, 2], [3, 3, 3]], [[4, 4, 4], [5, 5, 5], [6, 6, 6]]], dtype=np.float32)
    exp = np.array([[[0.59167546, 0.69312394, 0.59167546], [0.7956997, 0.93213004, 0.7956997], [0.7668002, 0.8982755, 0.7668002]],
                    [[1.6105323, 1.8866735, 1.6105323], [1.9892492, 2.3303251, 1.9892492], [1.7856569, 2.091825, 1.7856569]]])
    np.testing.assert_allclose(GaussianSmooth(sigma=1.5)(torch.from_numpy(x).to(device)).cpu().numpy(), exp, atol=1e-4)
    y = GaussianFilter(3, [1.0, 2.0, 0.7])(torch.from_numpy(g["gf_in"]).to(device))
    np.testing.assert_allclose(y.cpu().numpy(), g["gf_out"], atol=1e-5, rtol=1e-5)


def case_gaussian_z_chunks(device):
    """Long z axis: the streaming kernel cuts z into chunks with re-filtered halo planes; vs a zero-padded fp64 conv."""
    import torch.nn.functional as F

    from monai_amd import ops
    from monai_amd.networks.layers import gaussian_1d

    torch.manual_seed(21)
    for shape, sig in (((2, 90, 20, 70), (1.0, 1.0, 1.0)), ((1, 133, 17, 9), (0.5, 1.0, 2.0)), ((1, 50, 16, 64), (0.4, 0.4, 0.4))):
        x = torch.rand(shape)
        ks = [gaussian_1d(torch.tensor(s)) for s in sig]
        ref = x.double()[None]
        for ax, k in enumerate(ks):
            shp = [1, 1, 1, 1, 1]
            shp[2 + ax] = k.numel()
            pad = [0, 0, 0]
            pad[ax] = k.numel() // 2
            ref = F.conv3d(ref.transpose(0, 1), k.double().reshape(shp), padding=pad).transpose(0, 1)
        y = ops.separable_filter3d(x.to(device), [k.numpy() for k in ks])
        assert (y.cpu().double() - ref[0]).abs().max().item() < 2e-6, shape


def case_gaussian_rowvec_equals_tile(device):
    """The row-vector kernel (float4-aligned volumes, gaussian.h: 16-byte loads, register x-pass, one barrier per plane) performs the
    taps in the order of the round-1 tile kernel: BIT-IDENTICAL outputs -- ragged tiles (W = 260: a second x-tile with one live lane;
    H = 19), every tap-count class (3 / 5 / 9 / 17), anisotropic kernels, several z-chunks; plus a zero-padded fp64 convolution."""
    import os

    import torch.nn.functional as F

    from monai_amd import ops
    from monai_amd.networks.layers import gaussian_1d

    torch.manual_seed(23)
    for shape, sig in (((2, 40, 19, 260), (1.0, 1.0, 1.0)), ((1, 70, 33, 64), (0.5, 1.0, 2.0)), ((1, 21, 16, 512), (0.4, 0.4, 0.4)),
                       ((1, 37, 35, 20), (2.0, 2.0, 2.0)), ((1, 9, 8, 8), (0.7, 0.7, 0.7))):
        x = torch.rand(shape).to(device)
        ks = [gaussian_1d(torch.tensor(s)).numpy() for s in sig]
        os.environ["MONAI_AMD_GS_IMPL"] = "tile"
        try:
            a = ops.separable_filter3d(x, ks)
        finally:
            del os.environ["MONAI_AMD_GS_IMPL"]
        b = ops.separable_filter3d(x, ks)
        assert torch.equal(a, b), (shape, sig, float((a - b).abs().max()))
        ref = x.cpu().double()[None]
        for ax, k in enumerate(ks):
            shp = [1, 1, 1, 1, 1]
            shp[2 + ax] = len(k)
            pad = [0, 0, 0]
            pad[ax] = len(k) // 2
            ref = F.conv3d(ref.transpose(0, 1), torch.as_tensor(k).double().reshape(shp), padding=pad).transpose(0, 1)
        assert (b.cpu().double() - ref[0]).abs().max().item() < 2e-6, shape


def case_reference_argument_conventions(device):
    """Argument rules the reference's own tests pin (run over this package in tests/test_reference_suites_emu.py), restated so that the GPU box
    checks them without the reference: AffineTransform refuses an image whose dtype differs from theta's AFTER its shape checks
    (tests/networks/layers/test_affine_transform.py:280-333: ValueError first, RuntimeError for the dtype); scipy.ndimage padding names map to
    grid_sample's (monai/transforms/utils.py:2281-2297); Flip with a non-integer axis raises TypeError (test_flip.py:32)."""
    from monai_amd.data import MetaTensor
    from monai_amd.networks.layers import AffineTransform
    from monai_amd.transforms import Flip, Spacing
    from monai_amd.transforms.spatial.functional import _pad_name

    theta = torch.eye(4)[None].to(device)
    img = torch.arange(48, dtype=torch.float32).view(2, 1, 4, 2, 3).to(device)
    xf = AffineTransform((2, 3, 4), padding_mode="border", mode="bilinear", normalized=True)
    assert xf(img, theta.repeat(2, 1, 1)).shape == (2, 1, 2, 3, 4)
    for bad_img, bad_theta, err in ((img.to(torch.int32), theta.repeat(2, 1, 1), RuntimeError),        # dtype differs from theta's
                                    (img, theta.repeat(2, 1, 1).double(), RuntimeError),
                                    (img.to(torch.int32), theta.repeat(3, 1, 1), ValueError)):          # the batch check comes first
        try:
            xf(bad_img, bad_theta)
        except err:
            pass
        else:
            raise AssertionError(f"expected {err.__name__}")
    assert [_pad_name(p) for p in ("constant", "grid-constant", "nearest", "reflect", "wrap", "grid-mirror", "zeros", "border")] == \
        ["zeros", "zeros", "border", "reflection", "reflection", "reflection", "zeros", "border"]
    try:
        _pad_name("mirror!")
    except ValueError:
        pass
    else:
        raise AssertionError("unknown padding name must raise ValueError")
    vol = MetaTensor(torch.rand(1, 6, 6, 6).to(device), affine=np.diag([2.0, 2.0, 2.0, 1.0]))
    a = Spacing(pixdim=(1.0, 1.0, 1.0), padding_mode="constant")(vol)
    b = Spacing(pixdim=(1.0, 1.0, 1.0), padding_mode="zeros")(vol)
    assert torch.equal(a.as_tensor(), b.as_tensor())
    try:
        Flip(["s", 1])(vol)
    except TypeError:
        pass
    else:
        raise AssertionError("Flip with a non-integer axis must raise TypeError")


def case_general_rows_vs_linear(device):
    """Rotated / sheared matrices: the row-mapped kernel with compile-time mode and padding rule (the default) gives exactly what the
    linear-index kernel gives (MONAI_AMD_RS_GENERAL=linear), and both match a float64 torch restatement of trilinear sampling with border
    padding.  Ragged extents (Wo % 64, Ho % 4), several channels, coordinates far outside the volume, a NaN matrix entry."""
    import os

    from monai_amd import ops

    torch.manual_seed(12)
    vol = torch.rand(3, 9, 14, 37).to(device)
    th = 0.3
    rot = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.1], [0.05, -0.2, 0.8]])
    mats = []
    for off in ((1.0, -2.0, 3.0), (-30.0, 40.0, -50.0)):
        m = np.zeros((3, 4))
        m[:, :3], m[:, 3] = rot, off
        mats.append(m)
    mn = mats[0].copy()
    mn[1, 2] = np.nan
    saved = os.environ.pop("MONAI_AMD_RS_GENERAL", None)
    try:
        for m in mats + [mn]:
            for osz in ((7, 5, 70), (3, 9, 130), (11, 4, 64)):
                for mode in ("bilinear", "nearest"):
                    for pad in PADS:
                        for f64 in (True, False):
                            os.environ.pop("MONAI_AMD_RS_GENERAL", None)
                            a = ops.affine_resample(vol, m.reshape(-1), osz, mode, pad, False, f64)
                            os.environ["MONAI_AMD_RS_GENERAL"] = "linear"
                            b = ops.affine_resample(vol, m.reshape(-1), osz, mode, pad, False, f64)
                            assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), (osz, mode, pad, f64)
        os.environ.pop("MONAI_AMD_RS_GENERAL", None)
        # independent restatement (fp64): border padding, trilinear
        m, osz = mats[0], (7, 5, 70)
        got = ops.affine_resample(vol, m.reshape(-1), osz, "bilinear", "border", False, True).cpu().double()
        oz, oy, ox = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in osz], indexing="ij")
        v = vol.cpu().double().numpy()
        c = [np.clip(m[r, 0] * oz + m[r, 1] * oy + m[r, 2] * ox + m[r, 3], 0.0, v.shape[1 + r] - 1.0) for r in range(3)]
        f = [np.floor(x).astype(np.int64) for x in c]
        exp = np.zeros((v.shape[0],) + osz)
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    w = np.ones(osz)
                    idx = []
                    for r, d in enumerate((dz, dy, dx)):
                        t = c[r] - f[r]
                        w = w * (t if d else 1.0 - t)
                        i = f[r] + d
                        w = np.where(i < v.shape[1 + r], w, 0.0)
                        idx.append(np.minimum(i, v.shape[1 + r] - 1))
                    exp += v[:, idx[0], idx[1], idx[2]] * w
        assert np.abs(got.numpy() - exp).max() < 1e-6
    finally:
        os.environ.pop("MONAI_AMD_RS_GENERAL", None)
        if saved is not None:
            os.environ["MONAI_AMD_RS_GENERAL"] = saved


def case_separable_vs_general(device):
    """The axis-aligned fast path (per-axis tap tables + LDS-staged source box, with its global-gather fallback when the
    box does not fit) must give exactly what the general kernel gives for the same matrix: a 1e-300 off-diagonal makes
    the matrix formally non-separable without changing any coordinate."""
    from monai_amd import ops

    torch.manual_seed(11)
    vol = torch.rand(2, 24, 40, 300).to(device)
    for scale, osz in (((1.25, 1.25, 0.625), (19, 32, 480)), ((1.0, 3.0, 4.1), (24, 13, 73)), ((0.5, 0.5, 0.5), (47, 79, 599)),
                       ((-1.0, 0.8, -0.9), (24, 50, 333))):   # flips: decreasing tap tables
        for mode in ("bilinear", "nearest"):
            for pad in PADS:
                for f64 in (True, False):
                    m = np.zeros((3, 4))
                    m[0, 0], m[1, 1], m[2, 2] = scale
                    m[:, 3] = (-0.7, 0.4, -2.3) if scale[0] > 0 else (23.0, 0.4, 299.5)
                    a = ops.affine_resample(vol, m.reshape(-1), osz, mode, pad, False, f64)
                    m2 = m.copy()
                    m2[0, 1] = 1e-300
                    b = ops.affine_resample(vol, m2.reshape(-1), osz, mode, pad, False, f64)
                    assert torch.equal(a, b), (scale, mode, pad, f64, (a - b).abs().max().item())
    # the z-streaming kernels' other paths: the 16-loads-per-thread box, tiles (and planes) entirely outside the volume; round 5, the barrier-free wave form
    # (separable_resample_wave_kernel): in-plane down-sampling by 1.25 with an odd output width (4-byte stores), source planes skipped (z scale 2), the identity
    # scale at a fractional offset, wave tiles that end inside a workgroup tile (Ho = 35, 41), the tensor's last plane (its 16-byte pieces may pass the allocation's end)
    for scale, off, osz in (((1.25, 1.25, 1.0), (-0.7, 0.4, -2.3), (19, 32, 300)), ((1.25, 1.25, 0.625), (-9.0, -25.0, -100.0), (30, 40, 700)),
                            ((-1.25, 1.0, 0.625), (30.0, 30.0, 250.0), (33, 30, 300)),
                            ((0.625, 1.25, 1.25), (0.3, -0.6, 0.45), (38, 31, 239)), ((2.0, 1.0, 1.0), (0.5, 0.25, 0.75), (12, 41, 300)),
                            ((1.0, 1.0, 1.0), (0.5, 0.5, 0.5), (24, 35, 300)), ((0.9, 1.28, 1.9), (-1.2, -3.0, -7.5), (29, 33, 170))):
        for pad in PADS:
            for f64 in (True, False):
                m = np.zeros((3, 4))
                m[0, 0], m[1, 1], m[2, 2] = scale
                m[:, 3] = off
                a = ops.affine_resample(vol, m.reshape(-1), osz, "bilinear", pad, False, f64)
                m2 = m.copy()
                m2[0, 1] = 1e-300
                b = ops.affine_resample(vol, m2.reshape(-1), osz, "bilinear", pad, False, f64)
                assert torch.equal(a, b), (scale, off, pad, f64, (a - b).abs().max().item())
