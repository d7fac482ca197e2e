"""Kernel-level parity cases, shared by the CPU (SIMT-emulator) and the GPU (-m gpu) test modules.

Every case builds seeded inputs on the CPU, runs the C-ABI entry point through monai_amd.ops on `device`
("cpu" under the emulator fixture, "cuda" on the MI355X) and compares with a plain torch-CPU restatement
of the reference operator (the same ATen ops the oracle in oracle/ uses).  Tolerances are stated per case:
bit-exact for the blend / gather / pooling, 1e-5-class for fp32 convolutions whose summation order differs.
"""

from __future__ import annotations

import itertools

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from monai_amd import ops
from oracle import sliding_window as osw


def _act(x, nrm):
    """consumer-side view of a stored tensor: fma(x, alpha, beta) then leaky(slope) -- float4 per (n,c)"""
    if nrm is None:
        return x
    a = nrm[:, :, 0][:, :, None, None, None]
    b = nrm[:, :, 1][:, :, None, None, None]
    s = nrm[:, :, 2][:, :, None, None, None]
    y = torch.addcmul(b, x, a)
    return torch.where(y > 0, y, y * s)


def _rand_nrm(n, c, gen):
    nrm = torch.zeros(n, c, 4)
    nrm[:, :, 0] = torch.rand(n, c, generator=gen) * 1.5 + 0.25
    nrm[:, :, 0] *= torch.where(torch.rand(n, c, generator=gen) > 0.8, -1.0, 1.0)
    nrm[:, :, 1] = torch.randn(n, c, generator=gen) * 0.3
    nrm[:, :, 2] = 0.1
    return nrm


def _amax(x, nrm):
    """max |activated value| per (n, c): what a rigorous magnitude bound must not fall below"""
    return _act(x, nrm).abs().amax(dim=(2, 3, 4))


def _with_bounds(x, nrm, loosen=1.0):
    """records whose 4th component is a magnitude bound (include/monai_amd.h: mh_tensor5): max |act(x)| per (n, c), times `loosen`"""
    nrm = nrm.clone()
    nrm[:, :, 3] = torch.clamp(_amax(x, nrm) * loosen, min=1.2e-38)
    return nrm


# ------------------------------------------------------------------------------------------ sliding window
def case_window_extract(device, img=(2, 20, 24, 28), roi=(8, 12, 16), overlap=0.5):
    gen = torch.Generator().manual_seed(1)
    vol = torch.rand(img, generator=gen)
    itv = osw.get_scan_interval(img[1:], roi, (overlap,) * 3)
    starts, _ = osw.dense_patch_starts(img[1:], roi, itv)
    wins = list(itertools.product(*starts))
    w0, n = 3, len(wins) - 5
    out = torch.empty((n, img[0]) + tuple(roi), device=device)
    ops.window_extract(vol.to(device), starts, w0, n, roi, out)
    exp = torch.stack([vol[:, z:z + roi[0], y:y + roi[1], x:x + roi[2]] for (z, y, x) in wins[w0:w0 + n]])
    assert torch.equal(out.cpu(), exp)


def reference_blend(logits, imp, img, roi, starts):
    """The reference's scatter accumulation, monai/inferers/utils.py:264-298, on precomputed logits."""
    k = logits.shape[1]
    out = torch.zeros((1, k) + tuple(img))
    cnt = torch.zeros((1, 1) + tuple(img))
    wins = list(itertools.product(*starts))
    for (z, y, x) in wins:
        cnt[:, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]] += imp[None, None]
    for i, (z, y, x) in enumerate(wins):
        seg = logits[i:i + 1].clone()
        seg *= imp[None, None]
        out[:, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]] += seg
    out /= cnt
    return out[0]


def case_sw_blend(device, img=(24, 20, 32), roi=(16, 12, 16), overlap=0.5, k=5, mode="gaussian"):
    gen = torch.Generator().manual_seed(2)
    itv = osw.get_scan_interval(img, roi, (overlap,) * 3)
    starts, _ = osw.dense_patch_starts(img, roi, itv)
    nwin = int(np.prod([len(s) for s in starts]))
    logits = torch.randn((nwin, k) + tuple(roi), generator=gen)
    imp = osw.compute_importance_map(roi, mode=mode, sigma_scale=0.125)
    out = torch.empty((k,) + tuple(img), device=device)
    ops.sw_blend(logits.to(device), imp.to(device), out, starts, roi)
    exp = reference_blend(logits, imp, img, roi, starts)
    assert torch.equal(out.cpu(), exp), f"blend not bit-exact: max diff {(out.cpu() - exp).abs().max().item()}"
    # a padded window stride (the inferer's logits buffer, monai_amd/inferers/utils.py:_window_stride): same bits
    ws = logits[0].numel() + 12
    flat = torch.full((nwin * ws,), float("nan"))
    padded = flat.as_strided(logits.shape, (ws,) + tuple(logits.stride()[1:]))
    padded.copy_(logits)
    out2 = torch.empty((k,) + tuple(img), device=device)
    ops.sw_blend(padded.to(device) if device == "cpu" else flat.to(device).as_strided(logits.shape, (ws,) + tuple(logits.stride()[1:])),
                 imp.to(device), out2, starts, roi)
    assert torch.equal(out2.cpu(), exp), "blend with a padded window stride differs"
    # the argmax epilogue (AsDiscrete(argmax=True) fused into the blend): labels == torch.argmax of the blended logits, both dtypes
    for dt in (torch.float32, torch.uint8):
        lab = torch.full(tuple(img), 77, dtype=dt, device=device)
        ops.sw_blend_argmax(logits.to(device), imp.to(device), lab, starts, roi, k)
        assert torch.equal(lab.cpu().long(), exp.argmax(0)), f"fused argmax ({dt}) differs from argmax(blend)"


def case_sw_blend_special_values(device):
    """ties, NaN and infinities through the fused argmax: torch.argmax's rule (first maximal value, NaN maximal)"""
    img, roi, k = (8, 8, 16), (8, 8, 8), 6
    starts = [[0], [0], [0, 4, 8]]
    gen = torch.Generator().manual_seed(5)
    logits = torch.randint(-2, 3, (3, k) + roi, generator=gen).float()      # many exact ties
    logits[0, 2, 1, :, :4] = float("nan")
    logits[1, 4, 2, :, 4:] = float("inf")
    logits[2, 1, 3] = float("-inf")
    imp = torch.ones(roi)
    out = torch.empty((k,) + img, device=device)
    ops.sw_blend(logits.to(device), imp.to(device), out, starts, roi)
    exp = reference_blend(logits, imp, img, roi, starts)
    assert torch.equal(torch.nan_to_num(out.cpu(), nan=123.0), torch.nan_to_num(exp, nan=123.0))
    lab = torch.empty(img, dtype=torch.uint8, device=device)
    ops.sw_blend_argmax(logits.to(device), imp.to(device), lab, starts, roi, k)
    assert torch.equal(lab.cpu().long(), exp.argmax(0))


def case_sw_blend_irregular(device):
    """start lists that are not of dense_patch_slices' form take the table kernels (multi-resolution outputs whose scaled
    starts round unevenly): same bit-exact blend"""
    img, roi, k = (9, 12, 20), (4, 6, 8), 3
    starts = [[0, 1, 3, 5], [0, 6], [0, 4, 12]]
    gen = torch.Generator().manual_seed(7)
    nwin = 4 * 2 * 3
    logits = torch.randn((nwin, k) + roi, generator=gen)
    imp = osw.compute_importance_map(roi, mode="gaussian", sigma_scale=0.125)
    out = torch.empty((k,) + img, device=device)
    ops.sw_blend(logits.to(device), imp.to(device), out, starts, roi)
    assert torch.equal(out.cpu(), reference_blend(logits, imp, img, roi, starts))
    vol = torch.rand((2,) + img, generator=gen)
    got = torch.empty((nwin, 2) + roi, device=device)
    ops.window_extract(vol.to(device), starts, 0, nwin, roi, got)
    exp = torch.stack([vol[:, z:z + roi[0], y:y + roi[1], x:x + roi[2]] for (z, y, x) in itertools.product(*starts)])
    assert torch.equal(got.cpu(), exp)


def case_sw_blend_many_windows(device, slices=170):
    """more than 160 windows on one axis (SliceInferer: roi 1 along the slice axis, one window per slice)"""
    img, roi, k = (slices, 8, 8), (1, 8, 8), 2
    starts = [list(range(slices)), [0], [0]]
    gen = torch.Generator().manual_seed(9)
    logits = torch.randn((slices, k) + roi, generator=gen)
    imp = torch.ones(roi)
    out = torch.empty((k,) + img, device=device)
    ops.sw_blend(logits.to(device), imp.to(device), out, starts, roi)
    assert torch.equal(out.cpu(), reference_blend(logits, imp, img, roi, starts))
    vol = torch.rand((1,) + img, generator=gen)
    got = torch.empty((slices, 1) + roi, device=device)
    ops.window_extract(vol.to(device), starts, 0, slices, roi, got)
    assert torch.equal(got.cpu()[:, 0, 0], vol[0])


# ------------------------------------------------------------------------------------------ network blocks
def case_conv3d(device, cfg, n, cin, cout, dims, with_nrm=True, fused_stats=True, tol=2e-5):
    gen = torch.Generator().manual_seed(100 + cin + cout + dims[0])
    x = torch.randn((n, cin) + tuple(dims), generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) / np.sqrt(27.0 * cin)
    b = torch.randn(cout, generator=gen) * 0.1
    nrm = _rand_nrm(n, cin, gen) if with_nrm else None
    exp = F.conv3d(_act(x.double(), None if nrm is None else nrm.double()), w.double(), b.double(), padding=1)

    if cfg is None:
        cfg = ops.conv3d_k3_select(cin, cout, *dims)
    if cfg in (ops.conv3d_k3_h2_config(), ops.conv3d_k3_h2c_config(), ops.conv3d_k3_h2w_config()) and nrm is not None:
        # the split-precision kernel scales its input by the bounds its records carry: as loose as the finalize kernel's sqrt(count) bound
        nrm = _with_bounds(x, nrm, loosen=float(np.sqrt(np.prod(dims))))
    packed = ops.conv3d_k3_pack(cfg, w.to(device))
    out = torch.full((n, cout) + tuple(dims), float("nan"), device=device)
    tiles = ops.conv3d_k3_stat_tiles(cfg, *dims) if fused_stats else 0
    stats = torch.full((n, cout, tiles, 3), float("nan"), device=device) if tiles else None
    ops.conv3d_k3(cfg, x.to(device), None if nrm is None else nrm.to(device), packed, b.to(device), out, stats)
    got = out.cpu().double()
    err = (got - exp).abs().max().item()
    assert err < tol * max(1.0, exp.abs().max().item()), f"conv cfg {cfg} {cin}->{cout} {dims}: max err {err}"

    # statistics -> finalize -> {alpha, beta, slope}
    gamma = torch.rand(cout, generator=gen) + 0.5
    beta = torch.randn(cout, generator=gen) * 0.2
    if stats is None:
        tiles = ops.instnorm_stat_tiles(*dims)
        stats = torch.full((n, cout, tiles, 3), float("nan"), device=device)
        ops.instnorm_stats(out, stats)
    nrm_out = torch.full((n, cout, 4), float("nan"), device=device)
    ops.instnorm_finalize(stats, tiles, n, cout, gamma.to(device), beta.to(device), 1e-5, 0.1, nrm_out)
    mean = got.mean(dim=(2, 3, 4))
    var = got.var(dim=(2, 3, 4), unbiased=False)
    alpha = gamma.double()[None] / torch.sqrt(var + 1e-5)
    r = nrm_out.cpu().double()
    assert (r[:, :, 0] - alpha).abs().max().item() < 1e-5 * alpha.abs().max().item() + 1e-6
    assert (r[:, :, 1] - (beta.double()[None] - mean * alpha)).abs().max().item() < 2e-5
    assert torch.all(r[:, :, 2] == torch.tensor(0.1, dtype=torch.float32).double())
    # the magnitude bound: (|gamma| sqrt(count) + |beta|) max(1, |slope|), and never below what a consumer really sees
    cnt = float(np.prod(dims))
    assert torch.allclose(r[:, :, 3], (gamma.double().abs() * np.sqrt(cnt) + beta.double().abs())[None].expand(n, -1), rtol=1e-6)
    assert torch.all(r[:, :, 3] >= _amax(got, r[:, :, :3]))
    return cfg


def case_conv3d_pool(device, n, cin, cout, dims, cfg=None):
    """the split-precision convolution with the pooling epilogue (conv3d_h2.h, POOL): the convolution output and statistics bitwise those of the plain kernel, pool_max /
    pool_min bitwise MaxPool3d(2) of +out / -(-out); pool_select copies the minima for the channels whose alpha is negative; ragged regions, two z-chunks, two cout groups"""
    gen = torch.Generator().manual_seed(700 + cin + cout + dims[0])
    cfg = ops.conv3d_k3_h2_config() if cfg is None else cfg
    assert ops.conv3d_k3_pool_accepts(cfg, cin, cout, *dims)
    x = torch.randn((n, cin) + tuple(dims), generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) / np.sqrt(27.0 * cin)
    b = torch.randn(cout, generator=gen) * 0.1
    nrm = _with_bounds(x, _rand_nrm(n, cin, gen), loosen=float(np.sqrt(np.prod(dims))))
    packed = ops.conv3d_k3_pack(cfg, w.to(device))
    tiles = ops.conv3d_k3_stat_tiles(cfg, *dims)
    plain = torch.full((n, cout) + tuple(dims), float("nan"), device=device)
    st0 = torch.full((n, cout, tiles, 3), float("nan"), device=device)
    ops.conv3d_k3(cfg, x.to(device), nrm.to(device), packed, b.to(device), plain, st0)
    out = torch.full_like(plain, float("nan"))
    st1 = torch.full_like(st0, float("nan"))
    pdims = tuple(v // 2 for v in dims)
    pmx = torch.full((n, cout) + pdims, float("nan"), device=device)
    pmn = torch.full((n, cout) + pdims, float("nan"), device=device)
    ops.conv3d_k3_pool(cfg, x.to(device), nrm.to(device), packed, b.to(device), out, st1, pmx, pmn)
    assert torch.equal(out.cpu(), plain.cpu()) and torch.equal(st1.cpu(), st0.cpu())
    ref = plain.cpu()
    assert torch.equal(pmx.cpu(), F.max_pool3d(ref, 2)), (pmx.cpu() - F.max_pool3d(ref, 2)).abs().max()
    assert torch.equal(pmn.cpu(), -F.max_pool3d(-ref, 2))
    rec = torch.zeros((n, cout, 4))
    rec[:, :, 0] = torch.where(torch.arange(cout) % 3 == 1, -1.5, 0.75)[None]          # every third channel: negative alpha
    rec[0, 0, 0] = 0.0
    mx0 = pmx.clone()
    ops.pool_select(pmx, pmn, rec.to(device))
    want = torch.where((rec[:, :, 0] < 0)[:, :, None, None, None], pmn.cpu(), mx0.cpu())
    assert torch.equal(pmx.cpu(), want)
    return True


def case_conv3d_accumulate(device, n, cin, cout, dims, tol=2e-5, cfg=None):
    """the split-precision convolution in its accumulating form (conv3d_h2.h, ACC: out += conv + bias, statistics of the SUM): resident and streamed weight slabs, both
    region shapes, ragged regions, two z-chunks -- against old + conv in float64, and the finalize kernel on the statistics"""
    gen = torch.Generator().manual_seed(500 + cin + cout + dims[0])
    cfg = ops.conv3d_k3_h2_config() if cfg is None else cfg
    x = torch.randn((n, cin) + tuple(dims), generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) / np.sqrt(27.0 * cin)
    b = torch.randn(cout, generator=gen) * 0.1
    nrm = _with_bounds(x, _rand_nrm(n, cin, gen), loosen=float(np.sqrt(np.prod(dims))))
    old = torch.randn((n, cout) + tuple(dims), generator=gen) * 2.0
    exp = old.double() + F.conv3d(_act(x.double(), nrm.double()), w.double(), b.double(), padding=1)
    packed = ops.conv3d_k3_pack(cfg, w.to(device))
    out = old.clone().to(device)
    tiles = ops.conv3d_k3_stat_tiles(cfg, *dims)
    stats = torch.full((n, cout, tiles, 3), float("nan"), device=device)
    ops.conv3d_k3(cfg, x.to(device), nrm.to(device), packed, b.to(device), out, stats, accumulate=True)
    got = out.cpu().double()
    err = (got - exp).abs().max().item()
    assert err < tol * max(1.0, exp.abs().max().item()), f"accumulating conv {cin}->{cout} {dims}: max err {err}"
    gamma = torch.rand(cout, generator=gen) + 0.5
    beta = torch.randn(cout, generator=gen) * 0.2
    nrm_out = torch.full((n, cout, 4), float("nan"), device=device)
    ops.instnorm_finalize(stats, tiles, n, cout, gamma.to(device), beta.to(device), 1e-5, 0.1, nrm_out)
    mean, var = got.mean(dim=(2, 3, 4)), got.var(dim=(2, 3, 4), unbiased=False)
    alpha = gamma.double()[None] / torch.sqrt(var + 1e-5)
    r = nrm_out.cpu().double()
    assert (r[:, :, 0] - alpha).abs().max().item() < 1e-5 * alpha.abs().max().item() + 1e-6
    assert (r[:, :, 1] - (beta.double()[None] - mean * alpha)).abs().max().item() < 2e-5
    return err


def case_upconv_k4s2(device, n, cup, cout, ldims, with_bias=True, fused_stats=True, tol=2e-5):
    """UpCat's up half as ONE composite transposed convolution (csrc/kernels/upconv_h2.h): out += convT(k4, s2, p1)(act(low)) + bias table, against the two-layer
    evaluation conv3(deconv2(act(low)) + b_d) in float64 -- ragged tiles, both parities of every axis, the volume's borders (bias classes), several cout groups,
    two z-chunks -- and the statistics of the sum through the finalize kernel"""
    cin = 32
    gen = torch.Generator().manual_seed(300 + cup + cout + ldims[0] + ldims[2])
    dims = tuple(2 * v for v in ldims)
    low = torch.randn((n, cin) + tuple(ldims), generator=gen)
    nrm = _with_bounds(low, _rand_nrm(n, cin, gen), loosen=float(np.sqrt(np.prod(ldims))))
    wd = torch.randn((cin, cup, 2, 2, 2), generator=gen) / np.sqrt(float(cin))
    bd = torch.randn(cup, generator=gen) * 0.3 if with_bias else None
    wc = torch.randn((cout, cup, 3, 3, 3), generator=gen) / np.sqrt(27.0 * cup)
    ya = torch.randn((n, cout) + dims, generator=gen)                     # what the convolution's skip half left in `out`
    up = F.conv_transpose3d(_act(low.double(), nrm.double()), wd.double(), None if bd is None else bd.double(), stride=2)
    exp = ya.double() + F.conv3d(up, wc.double(), None, padding=1)

    assert ops.upconv_k4s2_accepts(cin, cout, *ldims)
    w4, table = ops.upconv_k4s2_weights(wd.to(device), None if bd is None else bd.to(device), wc.to(device))
    packed = ops.upconv_k4s2_pack(w4)
    out = ya.clone().to(device)
    tiles = ops.upconv_k4s2_stat_tiles(*ldims)
    stats = torch.full((n, cout, tiles, 3), float("nan"), device=device) if fused_stats else None
    ops.upconv_k4s2(low.to(device), nrm.to(device), packed, table, out, accumulate=True, stats=stats)
    got = out.cpu().double()
    # the writing form (the accumulating convolution adds the skip half afterwards): the same term without the old values, bit for bit the difference's source
    alone = torch.full((n, cout) + dims, float("nan"), device=device)
    ops.upconv_k4s2(low.to(device), nrm.to(device), packed, table, alone, accumulate=False)
    assert ((alone.cpu().double() + ya.double()) - exp).abs().max().item() < tol * max(1.0, exp.abs().max().item())
    err = (got - exp).abs().max().item()
    assert err < tol * max(1.0, exp.abs().max().item()), f"upconv {cin}->({cup})->{cout} {ldims}: max err {err}"
    if stats is not None:
        gamma = torch.rand(cout, generator=gen) + 0.5
        beta = torch.randn(cout, generator=gen) * 0.2
        nrm_out = torch.full((n, cout, 4), float("nan"), device=device)
        ops.instnorm_finalize(stats, tiles, n, cout, gamma.to(device), beta.to(device), 1e-5, 0.1, nrm_out)
        mean = got.mean(dim=(2, 3, 4))
        var = got.var(dim=(2, 3, 4), unbiased=False)
        alpha = gamma.double()[None] / torch.sqrt(var + 1e-5)
        r = nrm_out.cpu().double()
        assert (r[:, :, 0] - alpha).abs().max().item() < 1e-5 * alpha.abs().max().item() + 1e-6
        assert (r[:, :, 1] - (beta.double()[None] - mean * alpha)).abs().max().item() < 2e-5
    return err


def case_conv3d_k3_small_volume(device, n, cin, cout, dims, with_bias=True, fused_stats=True, tol=2e-5):
    """Conv3d k3 p1 of a SMALL volume (D H W <= 256) with the sample's whole volume as the workgroup's tile (csrc/kernels/conv3d_vol_h2.h, mh_conv3d_k3_h2v_config) against
    ATen in float64: volumes that do not fill the 256 rows, odd extents (scalar stores), one / two cout groups per workgroup, several channel chunks; statistics through
    the finalize kernel; what mh_conv3d_k3_select says about such shapes; a poisoned sample"""
    from monai_amd import config

    gen = torch.Generator().manual_seed(700 + cin + cout + dims[0] + 7 * dims[2])
    x = torch.randn((n, cin) + tuple(dims), generator=gen)
    nrm = _with_bounds(x, _rand_nrm(n, cin, gen), loosen=float(np.sqrt(np.prod(dims))))
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) / np.sqrt(27.0 * cin)
    b = torch.randn(cout, generator=gen) * 0.3 if with_bias else None
    exp = F.conv3d(_act(x.double(), nrm.double()), w.double(), None if b is None else b.double(), padding=1)
    cfg = ops.conv3d_k3_h2v_config()
    with config.conv_algo_scope("auto"):
        sel = ops.conv3d_k3_select(cin, cout, *dims, bounded=True)
        assert sel == (ops.conv3d_k3_h2_config() if (dims[1] >= 8 and dims[2] >= 8 and dims[2] % 4 == 0 and cin <= 256) else cfg), (sel, dims)
        assert ops.conv3d_k3_select(cin, cout, *dims, bounded=False) != cfg
    with config.conv_algo_scope("fp32"):
        assert ops.conv3d_k3_select(cin, cout, *dims, bounded=True) != cfg
    packed = ops.conv3d_k3_pack(cfg, w.to(device))
    out = torch.full(tuple(exp.shape), float("nan"), device=device)
    tiles = ops.conv3d_k3_stat_tiles(cfg, *dims)
    assert tiles == 1
    stats = torch.full((n, cout, tiles, 3), float("nan"), device=device) if fused_stats else None
    ops.conv3d_k3(cfg, x.to(device), nrm.to(device), packed, None if b is None else b.to(device), out, stats)
    got = out.cpu().double()
    err = (got - exp).abs().max().item()
    assert err < tol * max(1.0, exp.abs().max().item()), f"conv3d_k3 small volume {cin}->{cout} {dims}: max err {err}"
    if stats is not None:
        gamma = torch.rand(cout, generator=gen) + 0.5
        beta = torch.randn(cout, generator=gen) * 0.2
        nrm_out = torch.full((n, cout, 4), float("nan"), device=device)
        ops.instnorm_finalize(stats, tiles, n, cout, gamma.to(device), beta.to(device), 1e-5, 0.1, nrm_out)
        mean = got.mean(dim=(2, 3, 4))
        var = got.var(dim=(2, 3, 4), unbiased=False)
        alpha = gamma.double()[None] / torch.sqrt(var + 1e-5)
        r = nrm_out.cpu().double()
        assert (r[:, :, 0] - alpha).abs().max().item() < 1e-5 * alpha.abs().max().item() + 1e-6
        assert (r[:, :, 1] - (beta.double()[None] - mean * alpha)).abs().max().item() < 2e-5
    if n > 1:
        bad = nrm.clone()
        bad[1, 1, 3] = float("inf")
        out2 = torch.zeros(tuple(exp.shape), device=device)
        ops.conv3d_k3(cfg, x.to(device), bad.to(device), packed, None, out2, None)
        assert torch.isnan(out2[1]).all() and torch.isfinite(out2[0]).all()
    return err


def case_deconv_k2s2_h2(device, n, cin, cout, dims, with_bias=True, tol=2e-5):
    """ConvTranspose3d k2 s2 as one split-precision GEMM with (cout, parity) rows, stored pixel-shuffled (csrc/kernels/deconv_h2.h), into a channel slice of a wider
    buffer, against ATen in float64: volumes that end inside a wave, 16- and 32-channel output groups, more input channels than one weight chunk; the magnitude bound
    left in the output records; a poisoned sample"""
    from monai_amd import config

    gen = torch.Generator().manual_seed(600 + cin + cout + dims[0] + 5 * dims[2])
    x = torch.randn((n, cin) + tuple(dims), generator=gen)
    nrm = _with_bounds(x, _rand_nrm(n, cin, gen), loosen=3.0)
    w = torch.randn((cin, cout, 2, 2, 2), generator=gen) / np.sqrt(float(cin))
    b = torch.randn(cout, generator=gen) * 0.3 if with_bias else None
    exp = F.conv_transpose3d(_act(x.double(), nrm.double()), w.double(), None if b is None else b.double(), stride=2)
    odims = tuple(2 * v for v in dims)
    buf = torch.full((n, cout + 5) + odims, float("nan"), device=device)
    rec = torch.full((n, cout + 5, 4), float("nan"), device=device)
    with config.conv_algo_scope("auto"):
        assert config.deconv_h2()
        wd = w.to(device)
        ops.deconv_k2s2(x.to(device), nrm.to(device), wd, None if b is None else b.to(device), buf[:, 5:], ops.nrm_identity(rec[:, 5:]), bounded=True)
        got = buf[:, 5:].cpu().double()
        err = (got - exp).abs().max().item()
        assert err < tol * max(1.0, exp.abs().max().item()), f"deconv_k2s2_h2 {cin}->{cout} {dims}: max err {err}"
        assert torch.isnan(buf[:, :5]).all()
        r = rec.cpu()
        amax = got.abs().amax(dim=(2, 3, 4)).float()
        grp = 32 if cout % 32 == 0 else 16
        gmax = amax.view(n, cout // grp, grp).amax(dim=2, keepdim=True).expand(-1, -1, grp).reshape(n, cout)
        assert torch.all(r[:, 5:, 0] == 1.0) and torch.all(r[:, 5:, 1] == 0.0) and torch.all(r[:, 5:, 2] == 1.0)
        assert torch.equal(r[:, 5:, 3], gmax), "deconv_k2s2_h2: bound != max |value written| of the channel group"
        # against the direct fp32 kernel (same call without bounds), and a poisoned sample
        ref = torch.empty((n, cout) + odims, device=device)
        ops.deconv_k2s2(x.to(device), nrm.to(device), wd, None if b is None else b.to(device), ref)
        assert (ref.cpu().double() - got).abs().max().item() < tol * max(1.0, exp.abs().max().item())
        if n > 1:
            bad = nrm.clone()
            bad[1, 2, 3] = float("nan")
            out = torch.zeros((n, cout) + odims, device=device)
            ops.deconv_k2s2(x.to(device), bad.to(device), wd, None, out, bounded=True)
            assert torch.isnan(out[1]).all() and torch.isfinite(out[0]).all()
    return err


def case_conv3d_k3s2(device, n, cin, cout, dims, with_bias=True, fused_stats=True, tol=2e-5, fused=None):
    """Conv3d k3 s2 p1 on the fp16 matrix cores in split precision (csrc/kernels/conv3d_s2_h2.h: phase-split pass + GEMM over the 8 parity phases) against ATen in
    float64: ragged tiles, the zero padding at index -1 of every axis, one / two cout groups per workgroup, several z-chunks (run-in plane), and the InstanceNorm
    statistics of the result through the finalize kernel"""
    gen = torch.Generator().manual_seed(500 + cin + cout + dims[0] + 3 * dims[2])
    x = torch.randn((n, cin) + tuple(dims), generator=gen)
    nrm = _with_bounds(x, _rand_nrm(n, cin, gen), loosen=float(np.sqrt(np.prod(dims))))
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) / np.sqrt(27.0 * cin)
    b = torch.randn(cout, generator=gen) * 0.3 if with_bias else None
    exp = F.conv3d(_act(x.double(), nrm.double()), w.double(), None if b is None else b.double(), stride=2, padding=1)
    assert ops.conv3d_k3s2_accepts(cin, cout, *dims)
    packed = ops.conv3d_k3s2_pack(w.to(device))
    out = torch.full(tuple(exp.shape), float("nan"), device=device)
    tiles = ops.conv3d_k3s2_stat_tiles(*dims)
    stats = torch.full((n, cout, tiles, 3), float("nan"), device=device) if fused_stats else None
    if fused is None:          # both forms (phase-split pass + GEMM / conversion inside the GEMM's staging): the same products in the same order -> the same bits
        other = torch.full(tuple(exp.shape), float("nan"), device=device)
        ops.conv3d_k3s2(x.to(device), nrm.to(device), packed, None if b is None else b.to(device), other, None, fused=True)
        ops.conv3d_k3s2(x.to(device), nrm.to(device), packed, None if b is None else b.to(device), out, stats, fused=False)
        assert torch.equal(other, out), f"conv3d_k3s2 {cin}->{cout} {dims}: the fused and the split form differ"
        if stats is not None:
            st2 = torch.full_like(stats, float("nan"))
            ops.conv3d_k3s2(x.to(device), nrm.to(device), packed, None if b is None else b.to(device), other, st2, fused=True)
            assert torch.equal(st2, stats)
    else:
        ops.conv3d_k3s2(x.to(device), nrm.to(device), packed, None if b is None else b.to(device), out, stats, fused=fused)
    got = out.cpu().double()
    err = (got - exp).abs().max().item()
    assert err < tol * max(1.0, exp.abs().max().item()), f"conv3d_k3s2 {cin}->{cout} {dims}: max err {err}"
    if stats is not None:
        gamma = torch.rand(cout, generator=gen) + 0.5
        beta = torch.randn(cout, generator=gen) * 0.2
        nrm_out = torch.full((n, cout, 4), float("nan"), device=device)
        ops.instnorm_finalize(stats, tiles, n, cout, gamma.to(device), beta.to(device), 1e-5, 0.1, nrm_out)
        mean = got.mean(dim=(2, 3, 4))
        var = got.var(dim=(2, 3, 4), unbiased=False)
        alpha = gamma.double()[None] / torch.sqrt(var + 1e-5)
        r = nrm_out.cpu().double()
        assert (r[:, :, 0] - alpha).abs().max().item() < 1e-5 * alpha.abs().max().item() + 1e-6
        assert (r[:, :, 1] - (beta.double()[None] - mean * alpha)).abs().max().item() < 2e-5
    return err


def case_conv3d_k3s2_poison_and_scale(device):
    """the bound contract of the split-precision family on the stride-2 kernel: huge / tiny magnitudes keep fp32-equivalent relative precision (power-of-two input
    scale), a sample with a non-finite bound comes out NaN as a whole while its neighbour is untouched"""
    gen = torch.Generator().manual_seed(77)
    n, cin, cout, dims = 2, 16, 32, (4, 8, 8)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) / np.sqrt(27.0 * cin)
    packed = ops.conv3d_k3s2_pack(w.to(device))
    for mag in (3.0e4, 1.0e-6, 1.0e12):
        x = torch.randn((n, cin) + dims, generator=gen) * mag
        nrm = _with_bounds(x, _rand_nrm(n, cin, gen))
        exp = F.conv3d(_act(x.double(), nrm.double()), w.double(), None, stride=2, padding=1)
        out = torch.full(tuple(exp.shape), float("nan"), device=device)
        for fused in (False, True):
            out.fill_(float("nan"))
            ops.conv3d_k3s2(x.to(device), nrm.to(device), packed, None, out, fused=fused)
            assert (out.cpu().double() - exp).abs().max().item() < 2e-5 * exp.abs().max().item(), (mag, fused)
    x = torch.randn((n, cin) + dims, generator=gen)
    nrm = _with_bounds(x, _rand_nrm(n, cin, gen))
    nrm[1, 3, 3] = float("inf")
    exp = F.conv3d(_act(x.double(), nrm.double()), w.double(), None, stride=2, padding=1)
    for fused in (False, True):
        out = torch.zeros(tuple(exp.shape), device=device)
        ops.conv3d_k3s2(x.to(device), nrm.to(device), packed, None, out, fused=fused)
        assert torch.isnan(out[1]).all() and (out[0].cpu().double() - exp[0]).abs().max().item() < 2e-5, fused


def _conv_err(device, cfg, x, nrm, w, b, exp):
    packed = ops.conv3d_k3_pack(cfg, w.to(device))
    out = torch.full(tuple(exp.shape), float("nan"), device=device)
    tiles = ops.conv3d_k3_stat_tiles(cfg, *x.shape[2:])
    stats = torch.full((x.shape[0], w.shape[0], tiles, 3), float("nan"), device=device) if tiles else None
    ops.conv3d_k3(cfg, x.to(device), None if nrm is None else nrm.to(device), packed, b.to(device), out, stats)
    return out.cpu(), None if stats is None else stats.cpu()


def case_h2_input_scaling(device, cin=32, cout=32, dims=(4, 16, 16)):
    """The split-precision convolution at any input magnitude (VERDICT r2 weak #1 / ADVICE medium): activations of 1e-20 ... 1e20 -- far
    outside fp16's range either way, incl. > 65504 -- come out with the relative accuracy of the exact-fp32 tiles, because the kernel scales
    its input by a power of two taken from the records' bounds; tight and loose bounds (the finalize kernel's sqrt(count) factor) both hold."""
    h2 = ops.conv3d_k3_h2_config()
    fp32 = ops.conv3d_k3_select(cin, cout, *dims, algo=1)          # MH_ALGO_DIRECT: an exact-fp32 matrix-core tile
    assert 1 <= fp32 <= ops.conv3d_k3_num_configs()
    gen = torch.Generator().manual_seed(77)
    n = 2
    x0 = torch.randn((n, cin) + tuple(dims), generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) / np.sqrt(27.0 * cin)
    b0 = torch.randn(cout, generator=gen) * 0.1
    worst = 0.0
    for mag, loosen in ((1e-20, 1.0), (1e-6, 30.0), (1e-3, 900.0), (1.0, 1.0), (1.0, 3.0e4), (1e3, 30.0), (1e5, 1.0), (7e4, 900.0), (1e20, 30.0)):
        nrm = _rand_nrm(n, cin, gen)
        nrm[:, :, 0] *= mag                       # gamma-like spread over 40 orders of magnitude
        nrm[:, :, 1] *= mag
        nrm[0, 3, 1] = 0.8 * mag                  # one beta-dominated channel
        nrm = _with_bounds(x0, nrm, loosen)
        b = b0 * mag
        exp = F.conv3d(_act(x0.double(), nrm.double()), w.double(), b.double(), padding=1)
        got, _ = _conv_err(device, h2, x0, nrm, w, b, exp)
        ref, _ = _conv_err(device, fp32, x0, nrm, w, b, exp)
        scale = exp.abs().max().item()
        e_h2 = (got.double() - exp).abs().max().item() / scale
        e_32 = (ref.double() - exp).abs().max().item() / scale
        assert torch.isfinite(got).all(), f"magnitude {mag}: non-finite output"
        assert e_h2 < 2e-5 and e_h2 < 4.0 * e_32 + 1e-6, f"magnitude {mag} (bound x{loosen}): h2 {e_h2:.2e} vs fp32 tile {e_32:.2e}"
        worst = max(worst, e_h2)
    return worst


def case_h2_nonfinite_and_missing_bounds(device, cin=16, cout=32, dims=(3, 8, 8)):
    """A non-finite bound (the input plane or its statistics held inf / NaN) or a missing one (0: the caller selected the kernel for an input
    without bounds) turns THAT sample's output and statistics into NaN -- behind an InstanceNorm exactly the reference's result
    (blocks/convolutions.py:98-171: conv -> norm of a tensor with a non-finite element is NaN everywhere) -- and leaves the other samples alone."""
    h2 = ops.conv3d_k3_h2_config()
    gen = torch.Generator().manual_seed(78)
    n = 3
    x = torch.randn((n, cin) + tuple(dims), generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) / np.sqrt(27.0 * cin)
    b = torch.randn(cout, generator=gen) * 0.1
    base = _with_bounds(x, _rand_nrm(n, cin, gen), 10.0)
    exp = F.conv3d(_act(x.double(), base.double()), w.double(), b.double(), padding=1)
    for bad in (float("nan"), float("inf"), 0.0):
        nrm = base.clone()
        nrm[1, 5, 3] = bad
        got, stats = _conv_err(device, h2, x, nrm, w, b, exp)
        assert torch.isnan(got[1]).all() and torch.isnan(stats[1][..., 1:]).all(), f"bound {bad}: sample 1 must be NaN"
        for k in (0, 2):
            assert (got[k].double() - exp[k]).abs().max().item() < 2e-5 * max(1.0, exp.abs().max().item()), f"bound {bad}: sample {k} disturbed"
    # the reference's own behaviour for such an input, for the record: conv -> InstanceNorm of a tensor holding one inf is NaN everywhere
    xi = x.clone()
    xi[1, 0, 1, 2, 3] = float("inf")
    y = F.instance_norm(F.conv3d(xi, w, b, padding=1))
    assert torch.isnan(y[1]).all() and torch.isfinite(y[0]).all()


def case_bound_producers(device):
    """Raw producers leave max |value written| in the identity records of their output (nrm_identity + atomic fold); pooling and the replicate pad hand
    their input's bounds on; a non-finite element poisons the bound instead of vanishing in a max."""
    gen = torch.Generator().manual_seed(79)
    n, cin, cout, dims = 2, 16, 6, (4, 6, 10)
    x = torch.randn((n, cin) + dims, generator=gen)
    x[1] *= 1e4
    nrm = _rand_nrm(n, cin, gen)
    w = torch.randn((cin, cout, 2, 2, 2), generator=gen) / np.sqrt(cin)
    b = torch.randn(cout, generator=gen) * 0.1
    odims = tuple(2 * d for d in dims)
    # transposed convolution into a channel slice of a wider buffer (BasicUNet's concat buffer): only its records are touched
    buf = torch.zeros((n, 3 + cout) + odims, device=device)
    rec = torch.full((n, 3 + cout, 4), 7.0, device=device)
    ops.deconv_k2s2(x.to(device), nrm.to(device), w.to(device), b.to(device), buf[:, 3:], ops.nrm_identity(rec[:, 3:]))
    r = rec.cpu()
    assert torch.all(r[:, :3] == 7.0) and torch.all(r[:, 3:, 0] == 1) and torch.all(r[:, 3:, 1] == 0) and torch.all(r[:, 3:, 2] == 1)
    amax = buf.cpu()[:, 3:].abs().amax(dim=(2, 3, 4))
    # one reduction per workgroup over the channel group a thread owns (4 channels): every channel gets its group's maximum
    grp = torch.stack([amax[:, g:g + 4].amax(dim=1) for g in range(0, cout, 4)], dim=1).repeat_interleave(4, dim=1)[:, :cout]
    assert torch.equal(r[:, 3:, 3], grp) and torch.all(r[:, 3:, 3] >= amax), "deconv_k2s2: bound != max |value written| of the channel group"
    out = torch.empty((n, cout) + (dims[0], 2 * dims[1], 2 * dims[2]), device=device)
    rec2 = torch.empty((n, cout, 4), device=device)
    ops.deconv_ks(x.to(device), nrm.to(device), w[:, :, :1].contiguous().to(device), b.to(device), out, (1, 2, 2), ops.nrm_identity(rec2))
    assert torch.equal(rec2.cpu()[:, :, 3], out.cpu().abs().amax(dim=(2, 3, 4)))
    # residual join
    a, c = torch.randn(2, 5, 6, 8, 12, generator=gen), torch.randn(2, 5, 6, 8, 12, generator=gen) * 50.0
    na = _rand_nrm(2, 5, gen)
    o = torch.empty_like(a).to(device)
    rec3 = torch.empty((2, 5, 4), device=device)
    ops.add_act(a.to(device), na.to(device), c.to(device), None, 0.01, o, ops.nrm_identity(rec3))
    assert torch.equal(rec3.cpu()[:, :, 3], o.cpu().abs().amax(dim=(2, 3, 4)))
    c[1, 2, 0, 0, 0] = float("nan")
    c[0, 4, 1, 1, 1] = float("-inf")
    ops.add_act(a.to(device), na.to(device), c.to(device), None, 0.01, o, ops.nrm_identity(rec3))
    r3 = rec3.cpu()[:, :, 3]
    assert torch.isnan(r3[1, 2]) and torch.isinf(r3[0, 4]) and torch.isfinite(r3[0, :4]).all()
    # all-zero planes keep the floor (FLT_MIN: "known"), not 0 ("none given")
    ops.add_act(torch.zeros_like(a).to(device), None, None, None, 1.0, o, ops.nrm_identity(rec3))
    assert torch.all(rec3.cpu()[:, :, 3] == torch.finfo(torch.float32).tiny)
    # pooling / replicate padding: the input's bounds
    nb = _with_bounds(a, na, 3.0)
    po = torch.empty((2, 5, 3, 4, 6), device=device)
    prec = torch.empty((2, 5, 4), device=device)
    ops.maxpool2(a.to(device), nb.to(device), po, prec)
    pr = prec.cpu()
    assert torch.equal(pr[:, :, 3], nb[:, :, 3]) and torch.all(pr[:, :, 0] == 1) and torch.all(pr[:, :, 1] == 0) and torch.all(pr[:, :, 2] == 1)
    assert torch.all(pr[:, :, 3] >= po.cpu().abs().amax(dim=(2, 3, 4)))
    pad = torch.empty((2, 5, 7, 9, 13), device=device)
    prec2 = torch.empty((2, 5, 4), device=device)
    ops.pad_replicate(a.to(device), pad, nb.to(device), prec2)
    assert torch.equal(prec2.cpu()[:, :, 3], nb[:, :, 3])


def case_conv3d_into_channel_slice(device):
    """The skip connection writes straight into the first channels of the concat buffer (batch stride of the
    wider buffer), and reads use a nrm slice with the wider stride."""
    gen = torch.Generator().manual_seed(7)
    n, cin, cout, dims = 2, 8, 32, (8, 8, 16)
    x = torch.randn((n, cin) + dims, generator=gen)
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen) / np.sqrt(27.0 * cin)
    cfg = ops.conv3d_k3_select(cin, cout, *dims)
    assert cfg >= 1
    packed = ops.conv3d_k3_pack(cfg, w.to(device))
    buf = torch.full((n, 2 * cout) + dims, 7.0, device=device)
    ops.conv3d_k3(cfg, x.to(device), None, packed, None, buf[:, cout:], None)
    exp = F.conv3d(x.double(), w.double(), None, padding=1)
    got = buf.cpu().double()
    assert (got[:, cout:] - exp).abs().max().item() < 2e-5
    assert torch.all(got[:, :cout] == 7.0)


def case_maxpool(device, n=2, c=5, dims=(8, 12, 20)):
    gen = torch.Generator().manual_seed(3)
    x = torch.randn((n, c) + dims, generator=gen)
    nrm = _rand_nrm(n, c, gen)
    out = torch.empty((n, c) + tuple(d // 2 for d in dims), device=device)
    ops.maxpool2(x.to(device), nrm.to(device), out)
    a, b, s = (nrm[:, :, i][:, :, None, None, None] for i in range(3))
    y = torch.addcmul(b, x, a)  # fp32; the kernel's fmaf differs from mul+add by <= 1 ulp
    exp = F.max_pool3d(torch.where(y > 0, y, y * s), 2)
    assert (out.cpu() - exp).abs().max().item() < 1e-6
    out2 = torch.empty_like(out)
    ops.maxpool2(x.to(device), None, out2)
    assert torch.equal(out2.cpu(), F.max_pool3d(x, 2))


def case_deconv(device, n=2, cin=16, cout=6, dims=(4, 6, 10)):
    gen = torch.Generator().manual_seed(4)
    x = torch.randn((n, cin) + dims, generator=gen)
    w = torch.randn((cin, cout, 2, 2, 2), generator=gen) / np.sqrt(cin)
    b = torch.randn(cout, generator=gen) * 0.1
    nrm = _rand_nrm(n, cin, gen)
    buf = torch.zeros((n, 3 + cout) + tuple(2 * d for d in dims), device=device)
    ops.deconv_k2s2(x.to(device), nrm.to(device), w.to(device), b.to(device), buf[:, 3:])
    exp = F.conv_transpose3d(_act(x.double(), nrm.double()), w.double(), b.double(), stride=2)
    got = buf.cpu().double()
    assert (got[:, 3:] - exp).abs().max().item() < 1e-5
    assert torch.all(got[:, :3] == 0)


def case_conv1x1(device, n=2, cin=32, cout=5, dims=(6, 8, 12)):
    gen = torch.Generator().manual_seed(5)
    x = torch.randn((n, cin) + dims, generator=gen)
    w = torch.randn((cout, cin), generator=gen) / np.sqrt(cin)
    b = torch.randn(cout, generator=gen) * 0.1
    nrm = _rand_nrm(n, cin, gen)
    out = torch.empty((n, cout) + dims, device=device)
    ops.conv1x1(x.to(device), nrm.to(device), w.to(device), b.to(device), out)
    exp = F.conv3d(_act(x.double(), nrm.double()), w.double()[:, :, None, None, None], b.double())
    assert (out.cpu().double() - exp).abs().max().item() < 1e-5


def case_conv1x1_stats(device):
    """The 1x1x1 convolution that leaves the InstanceNorm statistics of its output (UnetResBlock's conv3 -> norm3 shortcut): same output as the plain launch, bit for
    bit, and {alpha, beta} after the finalize equal to those of a stand-alone statistics pass over the output -- 16 + 8 + tail output channels, a plane that is not a
    whole number of workgroup tiles, a plane whose size is not a multiple of 4 (one voxel per thread), a large mean (E[x^2] - E[x]^2 would lose digits)."""
    gen = torch.Generator().manual_seed(15)
    for n, cin, cout, dims in ((2, 32, 16, (6, 8, 44)), (1, 8, 27, (5, 7, 9)), (2, 4, 5, (16, 16, 20)), (1, 16, 48, (4, 4, 4))):
        x = torch.randn((n, cin) + dims, generator=gen)
        w = torch.randn((cout, cin), generator=gen) / np.sqrt(cin)
        b = torch.randn(cout, generator=gen) * 0.1 + 30.0
        nrm = _rand_nrm(n, cin, gen)
        plain = torch.empty((n, cout) + dims, device=device)
        ops.conv1x1(x.to(device), nrm.to(device), w.to(device), b.to(device), plain)
        out = torch.full((n, cout) + dims, float("nan"), device=device)
        tiles = ops.conv1x1_stat_tiles(*dims)
        stats = torch.full((n, cout, tiles, 3), float("nan"), device=device)
        ops.conv1x1(x.to(device), nrm.to(device), w.to(device), b.to(device), out, stats)
        assert torch.equal(out, plain), (cin, cout, dims)
        st = stats.cpu().double()
        assert not torch.isnan(st).any()
        vox = dims[0] * dims[1] * dims[2]
        assert torch.all(st[..., 0].sum(-1) == vox), "every voxel counted once"
        got = torch.empty((n, cout, 4), device=device)
        ops.instnorm_finalize(stats, tiles, n, cout, None, None, 1e-5, 1.0, got)
        od = out.cpu().double()
        alpha = 1.0 / torch.sqrt(od.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
        g = got.cpu().double()
        assert ((g[:, :, 0] - alpha).abs() / alpha).max().item() < 2e-6, (cin, cout, dims)
        assert (g[:, :, 1] + od.mean(dim=(2, 3, 4)) * alpha).abs().max().item() < 5e-5 * max(1.0, float(alpha.max())), (cin, cout, dims)


def case_conv1x1_h2(device, n=2, cin=96, cout=48, dims=(6, 8, 44), scale=1.0, loosen=3.0):
    """mh_conv1x1_h2_f32: the 1x1x1 convolution with all output channels from one read of the input (split precision on the matrix cores) against an fp64
    convolution at the exact-fp32 kernel's tolerance, its statistics against a stand-alone pass over its own output, and against conv1x1 (the VALU kernel) on the
    same input -- ragged last workgroup tile (D*H*W not a multiple of 1024), Cin not a multiple of 16, Cout = 48 / 64 / 80 / 5 (half-filled second tile, two passes)"""
    gen = torch.Generator().manual_seed(23 + cin + cout)
    x = torch.randn((n, cin) + dims, generator=gen) * scale
    w = torch.randn((cout, cin), generator=gen) / np.sqrt(cin)
    b = torch.randn(cout, generator=gen) * 0.1 + 3.0
    nrm = _with_bounds(x, _rand_nrm(n, cin, gen), loosen)
    assert ops.conv1x1_h2_accepts(cin, cout, *dims)
    packed = ops.conv1x1_h2_pack(w.to(device))
    tiles = ops.conv1x1_stat_tiles(*dims)
    out = torch.full((n, cout) + dims, float("nan"), device=device)
    stats = torch.full((n, cout, tiles, 3), float("nan"), device=device)
    ops.conv1x1_h2(x.to(device), nrm.to(device), packed, b.to(device), out, stats)
    plain = torch.full((n, cout) + dims, float("nan"), device=device)
    ops.conv1x1_h2(x.to(device), nrm.to(device), packed, b.to(device), plain)
    assert torch.equal(out, plain)
    exp = F.conv3d(_act(x.double(), nrm.double()), w.double()[:, :, None, None, None], b.double())
    ref_scale = max(1.0, exp.abs().max().item())
    err = (out.cpu().double() - exp).abs().max().item()
    valu = torch.empty((n, cout) + dims, device=device)
    ops.conv1x1(x.to(device), nrm.to(device), w.to(device), b.to(device), valu)
    err_valu = (valu.cpu().double() - exp).abs().max().item()
    assert err < 4e-6 * ref_scale and err < 4 * err_valu + 1e-7 * ref_scale, (cin, cout, err, err_valu)
    st = stats.cpu().double()
    assert not torch.isnan(st).any()
    vox = dims[0] * dims[1] * dims[2]
    assert torch.all(st[..., 0].sum(-1) == vox), "every voxel counted once"
    got = torch.empty((n, cout, 4), device=device)
    ops.instnorm_finalize(stats, tiles, n, cout, None, None, 1e-5, 1.0, got)
    od = out.cpu().double()
    alpha = 1.0 / torch.sqrt(od.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    g = got.cpu().double()
    assert ((g[:, :, 0] - alpha).abs() / alpha).max().item() < 2e-6, (cin, cout, dims)
    assert (g[:, :, 1] + od.mean(dim=(2, 3, 4)) * alpha).abs().max().item() < 5e-5 * max(1.0, float(alpha.max())), (cin, cout, dims)
    return err


def case_conv1x1_sum2(device, n=2, cin=16, cout=5, dims=(6, 8, 44)):
    """mh_conv1x1_sum2_f32 (the residual join inside the output convolution) == add_act followed by conv1x1, bit for bit; and both against fp64"""
    gen = torch.Generator().manual_seed(31 + cin + cout)
    a, b = torch.randn((n, cin) + dims, generator=gen), torch.randn((n, cin) + dims, generator=gen)
    na, nb = _rand_nrm(n, cin, gen), _rand_nrm(n, cin, gen)
    na[:, :, 2] = 1.0
    w = torch.randn((cout, cin), generator=gen) / np.sqrt(cin)
    bias = torch.randn(cout, generator=gen) * 0.1
    got = torch.full((n, cout) + dims, float("nan"), device=device)
    assert ops.conv1x1_sum2_accepts(cout, *dims)
    ops.conv1x1_sum2(a.to(device), na.to(device), b.to(device), nb.to(device), 0.01, w.to(device), bias.to(device), got)
    joined = torch.empty((n, cin) + dims, device=device)
    ops.add_act(a.to(device), na.to(device), b.to(device), nb.to(device), 0.01, joined)
    two = torch.empty((n, cout) + dims, device=device)
    ops.conv1x1(joined, None, w.to(device), bias.to(device), two)
    assert torch.equal(got, two)
    exp = F.conv3d(F.leaky_relu(_act(a.double(), na.double()) + _act(b.double(), nb.double()), 0.01), w.double()[:, :, None, None, None], bias.double())
    assert (got.cpu().double() - exp).abs().max().item() < 1e-5
    # identity shortcut (no records on b)
    ops.conv1x1_sum2(a.to(device), na.to(device), b.to(device), None, 0.01, w.to(device), bias.to(device), got)
    ops.add_act(a.to(device), na.to(device), b.to(device), None, 0.01, joined)
    ops.conv1x1(joined, None, w.to(device), bias.to(device), two)
    assert torch.equal(got, two)
    return True


def case_instnorm_stats(device, n=2, c=3, dims=(10, 17, 31)):
    gen = torch.Generator().manual_seed(6)
    x = torch.randn((n, c) + dims, generator=gen) * 3 + 50.0  # large mean: E[x^2]-E[x]^2 would lose digits
    tiles = ops.instnorm_stat_tiles(*dims)
    stats = torch.empty((n, c, tiles, 3), device=device)
    ops.instnorm_stats(x.to(device), stats)
    nrm = torch.empty((n, c, 4), device=device)
    ops.instnorm_finalize(stats, tiles, n, c, None, None, 1e-5, 0.25, nrm)
    xd = x.double()
    alpha = 1.0 / torch.sqrt(xd.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
    r = nrm.cpu().double()
    assert (r[:, :, 0] - alpha).abs().max().item() < 1e-6
    assert (r[:, :, 1] + xd.mean(dim=(2, 3, 4)) * alpha).abs().max().item() < 2e-5
    assert torch.all(r[:, :, 2] == 0.25)


def case_attention(device, b=2, s=216, heads=3, hd=64, tol=2e-6):
    """softmax(Q K^T * scale) V against the reference's einsum formulation (selfattention.py:189-212) in fp64."""
    gen = torch.Generator().manual_seed(8 + s + hd)
    qkv = torch.randn(b, s, 3 * heads * hd, generator=gen) * 0.7
    qkv[0, :, : heads * hd] *= 3.0          # sharper softmax rows in one batch element
    out = ops.attention(qkv.to(device), heads, hd ** -0.5, hd)
    t = qkv.double().reshape(b, s, 3, heads, hd).permute(2, 0, 3, 1, 4)       # "b h (qkv l d) -> qkv b l h d"
    q, k, v = t[0], t[1], t[2]
    att = (torch.einsum("blxd,blyd->blxy", q, k) * (hd ** -0.5)).softmax(dim=-1)
    exp = torch.einsum("bhxy,bhyd->bhxd", att, v).permute(0, 2, 1, 3).reshape(b, s, heads * hd)
    err = (out.cpu().double() - exp).abs().max().item()
    assert err < tol, (s, hd, err)
    return err


def case_window_attention(device, bw=4, s=343, heads=3, hd=16, nw=2, tol=2e-6):
    """WindowAttention core (monai/networks/nets/swin_unetr.py:519-541): softmax((q scale) k^T + bias[head] + mask[window % nW]) v per (window, head) against
    fp64 torch -- head dim 8 = the VALU kernel, 16 / 32 = the split-precision matrix-core kernel (round 4); ragged last key tile (343 = 10 x 32 + 23), several query blocks"""
    gen = torch.Generator().manual_seed(7 + s + hd)
    qkv = torch.randn(bw, s, 3 * heads * hd, generator=gen)
    bias = torch.randn(heads, s, s, generator=gen) * 0.5              # [head][query][key]
    mask = torch.where(torch.rand(nw, s, s, generator=gen) > 0.8, -100.0, 0.0)
    mask = torch.minimum(mask, mask.transpose(1, 2))                  # symmetric, like the shift mask
    scale = hd ** -0.5
    got = ops.window_attention(qkv.to(device), heads, scale, bias.transpose(1, 2).contiguous().to(device), mask.to(device)).cpu().double()
    q, k, v = qkv.double().reshape(bw, s, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = (q * scale) @ k.transpose(-2, -1) + bias.double()[None] + mask.double()[torch.arange(bw) % nw][:, None]
    exp = (att.softmax(-1) @ v).transpose(1, 2).reshape(bw, s, heads * hd)
    err = (got - exp).abs().max().item()
    assert err < tol * max(1.0, exp.abs().max().item()), f"window attention S={s} hd={hd}: max err {err}"
    nomask = ops.window_attention(qkv.to(device), heads, scale, None, None).cpu().double()
    exp2 = (((q * scale) @ k.transpose(-2, -1)).softmax(-1) @ v).transpose(1, 2).reshape(bw, s, heads * hd)
    assert (nomask - exp2).abs().max().item() < tol * max(1.0, exp2.abs().max().item())
    return err


def _swin_relative_index(ws):
    """relative_position_index of a 3-D window as WindowAttention.__init__ builds it (monai/networks/nets/swin_unetr.py:492-519) -- restated from its description:
    per-axis coordinate differences, shifted to start at 0, mixed-radix with digits (2 ws - 1)"""
    zz, yy, xx = torch.meshgrid(torch.arange(ws[0]), torch.arange(ws[1]), torch.arange(ws[2]), indexing="ij")
    co = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1)])               # [3, S]
    rel = co[:, :, None] - co[:, None, :] + torch.tensor([ws[0] - 1, ws[1] - 1, ws[2] - 1])[:, None, None]
    return rel[0] * (2 * ws[1] - 1) * (2 * ws[2] - 1) + rel[1] * (2 * ws[2] - 1) + rel[2]


def case_window_attention_rel(device, bw=4, ws=(7, 7, 7), n=None, heads=3, hd=16, nw=2, masked=True, tol=2e-6):
    """mh_window_attention_rel_f32: bias gathered from the relative-position table and mask from region ids inside the kernel == the table form
    (mh_window_attention_f32 on the materialised [heads, S, S] bias and [nW, S, S] mask) BIT FOR BIT, and both against fp64 torch.  n < prod(ws): the
    reference's clamped-window case (relative_position_index[:n, :n] of the full window's index, swin_unetr.py:524-528)"""
    s_full = ws[0] * ws[1] * ws[2]
    n = n or s_full
    gen = torch.Generator().manual_seed(11 + n + hd)
    rows = (2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1)
    table = torch.randn(rows, heads, generator=gen) * 0.5
    index = _swin_relative_index(ws)[:n, :n]
    qkv = torch.randn(bw, n, 3 * heads * hd, generator=gen)
    region = torch.randint(0, 5, (nw, n), generator=gen).to(torch.int32) if masked else None
    scale = hd ** -0.5
    bias = table[index.reshape(-1)].reshape(n, n, heads).permute(2, 0, 1).contiguous()       # [head][query][key]
    mask = None
    if masked:
        diff = region[:, None, :].float() - region[:, :, None].float()
        mask = torch.where(diff != 0, -100.0, 0.0).contiguous()
    coord, off = index[:, 0].to(torch.int32).contiguous(), int(index[0, 0])
    assert ops.window_attention_rel_accepts(n, hd, rows)
    got = ops.window_attention_rel(qkv.to(device), heads, scale, table.to(device), coord.to(device), off, None if region is None else region.to(device)).cpu()
    tab = ops.window_attention(qkv.to(device), heads, scale, bias.transpose(1, 2).contiguous().to(device), None if mask is None else mask.to(device), exact=False).cpu()
    assert torch.equal(got, tab), f"window attention rel vs table form: {(got - tab).abs().max().item()}"
    q, k, v = qkv.double().reshape(bw, n, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = (q * scale) @ k.transpose(-2, -1) + bias.double()[None]
    if masked:
        att = att + mask.double()[torch.arange(bw) % nw][:, None]
    exp = (att.softmax(-1) @ v).transpose(1, 2).reshape(bw, n, heads * hd)
    err = (got.double() - exp).abs().max().item()
    assert err < tol * max(1.0, exp.abs().max().item()), f"window attention rel n={n} hd={hd}: max err {err}"
    return err


def case_add_act(device):
    gen = torch.Generator().manual_seed(9)
    a, b = torch.randn(2, 5, 6, 8, 12, generator=gen), torch.randn(2, 5, 6, 8, 12, generator=gen)
    na, nb = _rand_nrm(2, 5, gen), _rand_nrm(2, 5, gen)
    na[:, :, 2] = 1.0
    nb[:, :, 2] = 1.0
    out = torch.empty_like(a).to(device)
    ops.add_act(a.to(device), na.to(device), b.to(device), nb.to(device), 0.01, out)
    y = _act(a, na) + _act(b, nb)
    assert (out.cpu() - torch.where(y > 0, y, y * 0.01)).abs().max().item() < 2e-6
    ops.add_act(a.to(device), na.to(device), b.to(device), None, 0.01, out)
    y = _act(a, na) + b
    assert (out.cpu() - torch.where(y > 0, y, y * 0.01)).abs().max().item() < 2e-6


def case_strided_conv_and_deconv_k3(device):
    """UNet's strided conv (k3 s2 p1) and transposed conv (k3 s2 p1 op1), plus stride 1, vs ATen in fp64."""
    gen = torch.Generator().manual_seed(12)
    for stride, dims, cout in ((2, (8, 10, 12), 20), (1, (5, 6, 7), 20), (2, (7, 9, 6), 20), (2, (4, 6, 10), 5), (2, (5, 4, 6), 16), (2, (3, 4, 5), 32),
                              (1, (4, 6, 10), 5), (1, (3, 5, 4), 3), (1, (4, 4, 6), 8)):      # exact-width passes (UNet's tiny top level)
        n, cin = 2, 6
        x = torch.randn((n, cin) + dims, generator=gen)
        nrm = _rand_nrm(n, cin, gen)
        w = torch.randn((cout, cin, 3, 3, 3), generator=gen) * 0.1
        b = torch.randn(cout, generator=gen) * 0.1
        exp = F.conv3d(_act(x.double(), nrm.double()), w.double(), b.double(), stride=stride, padding=1)
        out = torch.full(tuple(exp.shape), float("nan"), device=device)
        ops.conv3d_k3_strided(x.to(device), nrm.to(device), ops.conv3d_k3_pack(0, w.to(device)), b.to(device), out, stride)
        assert (out.cpu().double() - exp).abs().max().item() < 2e-5, stride
        wt = torch.randn((cin, cout, 3, 3, 3), generator=gen) * 0.1
        exp = F.conv_transpose3d(_act(x.double(), nrm.double()), wt.double(), b.double(), stride=stride, padding=1, output_padding=stride - 1)
        out = torch.full(tuple(exp.shape), float("nan"), device=device)
        ops.deconv_k3(x.to(device), nrm.to(device), wt.to(device), b.to(device), out, stride)
        assert (out.cpu().double() - exp).abs().max().item() < 2e-5, stride
    # materialise a deferred tensor (add_act without a second operand)
    a = torch.randn(2, 3, 4, 5, 8, generator=gen)
    na = _rand_nrm(2, 3, gen)
    out = torch.empty_like(a).to(device)
    ops.add_act(a.to(device), na.to(device), None, None, 1.0, out)
    assert (out.cpu() - _act(a, na)).abs().max().item() < 2e-6


def case_linear(device, m, n, k, gelu=False, residual=False, bias=True, tol=2e-5):
    """nn.Linear (+ GELU, + residual) on the fp16 matrix cores in two-piece split precision: held to the tolerance of an fp32 GEMM."""
    gen = torch.Generator().manual_seed(1000 + m + n + k)
    x = torch.randn((m, k), generator=gen)
    w = torch.randn((n, k), generator=gen) / np.sqrt(k)
    b = torch.randn(n, generator=gen) * 0.1 if bias else None
    r = torch.randn((m, n), generator=gen) if residual else None
    exp = F.linear(x.double(), w.double(), None if b is None else b.double())
    if gelu:
        exp = F.gelu(exp)
    if r is not None:
        exp = exp + r.double()
    packed = ops.linear_pack(w.to(device))
    got = ops.linear(x.to(device), packed, n, None if b is None else b.to(device), None if r is None else r.to(device), gelu=gelu).cpu().double()
    err = (got - exp).abs().max().item()
    assert err < tol * max(1.0, exp.abs().max().item()), f"linear {m}x{k} -> {n} gelu={gelu} res={residual}: max err {err}"
    # the workgroup tiles (128 x 64 | 128 x 128: the large one is chosen for the token counts of the ViT blocks) give the same bits
    for tile in (64, 128):
        gt = ops.linear(x.to(device), packed, n, None if b is None else b.to(device), None if r is None else r.to(device), gelu=gelu, tile=tile).cpu().double()
        assert torch.equal(gt, got), f"linear tile {tile}: {float((gt - got).abs().max())}"
    with pytest.raises(RuntimeError):       # an unknown tile is an argument error, not a silent default
        ops.linear(x.to(device), packed, n, tile=96)
    # leading dimensions are flattened
    got3 = ops.linear(x.reshape(2, m // 2, k).to(device), packed, n, None if b is None else b.to(device), gelu=gelu)
    assert got3.shape == (2, m // 2, n)
    return err


def case_layernorm(device, m, k):
    gen = torch.Generator().manual_seed(2000 + m + k)
    x = torch.randn((m, k), generator=gen) * 3.0 + 1.5
    g = torch.rand(k, generator=gen) + 0.5
    b = torch.randn(k, generator=gen) * 0.2
    exp = F.layer_norm(x.double(), (k,), g.double(), b.double(), 1e-5)
    got = ops.layernorm(x.to(device), g.to(device), b.to(device), 1e-5).cpu().double()
    assert (got - exp).abs().max().item() < 5e-6, (got - exp).abs().max().item()
    got = ops.layernorm(x.to(device), None, None, 1e-6).cpu().double()
    assert (got - F.layer_norm(x.double(), (k,), None, None, 1e-6)).abs().max().item() < 5e-6


def case_layernorm_gather_linear_scatter(device, m_src=150, m_out=210, k=48, n=96):
    """mh_layernorm_gather_f32 / mh_linear_scatter_f32 (SwinTransformerBlock's copies folded into the kernels around the attention): rows gathered through an int32
    map with -1 = zero rows, and scattered back through it with the residual taken at the destination -- each against the separate operations, bit for bit"""
    gen = torch.Generator().manual_seed(77 + k)
    x = torch.randn(m_src, k, generator=gen) * 2.0 + 0.3
    g, b = torch.rand(k, generator=gen) + 0.5, torch.randn(k, generator=gen) * 0.2
    perm = torch.randperm(m_out, generator=gen)
    rows = torch.full((m_out,), -1, dtype=torch.int32)
    rows[perm[:m_src]] = torch.arange(m_src, dtype=torch.int32)            # every source row exactly once, the other m_out - m_src rows are padding
    xd, gd, bd, rd = x.to(device), g.to(device), b.to(device), rows.to(device)
    assert ops.layernorm_gather_accepts(k)
    got = ops.layernorm_gather(xd, gd, bd, 1e-5, rd).cpu()
    plain = ops.layernorm(xd, gd, bd, 1e-5).cpu()
    exp = torch.zeros(m_out, k)
    exp[rows >= 0] = plain[rows[rows >= 0].long()]
    assert torch.equal(got, exp)
    ref = F.layer_norm(x.double(), (k,), g.double(), b.double(), 1e-5)
    assert (plain.double() - ref).abs().max().item() < 5e-6
    # the way back: a linear map of the gathered rows, written to the source order with the residual added there
    w = torch.randn(n, k, generator=gen) / k ** 0.5
    bias = torch.randn(n, generator=gen) * 0.1
    res = torch.randn(m_src, n, generator=gen)
    packed = ops.linear_pack(w.to(device))
    y_rows = ops.linear(got.to(device), packed, n, bias.to(device)).cpu()                   # [m_out, n]
    scat = ops.linear_scatter(got.to(device), packed, n, bias.to(device), res.to(device), rd).cpu()
    exp2 = torch.empty(m_src, n)
    exp2[rows[rows >= 0].long()] = y_rows[rows >= 0] + res[rows[rows >= 0].long()]
    assert torch.equal(scat, exp2), (scat - exp2).abs().max().item()
    return True


def case_sw_blend_mosaic(device):
    """The mosaic logits layout (residue classes of non-overlapping windows stored as dense arrays, kernels/sliding.h): the blend over it is BIT-IDENTICAL to
    the window-major blend (and so to the reference's scatter order) -- overlap 0.5 with a clipped last window (two classes + the last), overlap 0.25
    (two classes with gaps), overlap 0.75 (four classes), axes with one / two windows, 1 ... 8 output channels; conv1x1_windows writes the layout."""
    gen = torch.Generator().manual_seed(41)
    for img, roi, overlap, k in (((40, 24, 56), (16, 12, 16), 0.5, 5), ((44, 20, 36), (16, 12, 16), 0.25, 3), ((28, 12, 40), (16, 12, 16), 0.75, 2),
                                 ((16, 12, 72), (16, 12, 16), 0.5, 8), ((48, 30, 20), (16, 12, 16), 0.5, 1)):
        itv = osw.get_scan_interval(img, roi, (overlap,) * 3)
        starts, _ = osw.dense_patch_starts(img, roi, itv)
        assert ops.LogitsMosaic.supported(starts, roi, k), (img, overlap)
        nwin = int(np.prod([len(s) for s in starts]))
        logits = torch.randn((nwin, k) + tuple(roi), generator=gen)
        imp = osw.compute_importance_map(roi, mode="gaussian", sigma_scale=0.125)
        exp = reference_blend(logits, imp, img, roi, starts)
        mos = ops.LogitsMosaic(starts, roi, k, device)
        mos.flat.fill_(float("nan"))
        for w in range(nwin):
            mos.window_view(w).copy_(logits[w].to(device))
        assert int(torch.isnan(mos.flat).sum()) == mos.flat.numel() - logits.numel(), "windows must tile the class arrays exactly once"
        out = torch.full((k,) + tuple(img), float("nan"), device=device)
        ops.sw_blend_mosaic(mos, imp.to(device), out)
        assert torch.equal(out.cpu(), exp), f"mosaic blend {img} overlap {overlap}: max diff {(out.cpu() - exp).abs().max().item()}"
        # the importance map given by its factors [gz | gy | gx | floor] and re-formed in registers: the same bits
        from monai_amd.data.utils import importance_map_factors

        fac = importance_map_factors(roi, "gaussian", 0.125)
        assert fac is not None
        out.fill_(float("nan"))
        ops.sw_blend_mosaic(mos, torch.cat([fac[0], fac[1], fac[2], torch.tensor([fac[3]])]).to(device), out)
        assert torch.equal(out.cpu(), exp), f"mosaic blend with a factored importance map {img}: max diff {(out.cpu() - exp).abs().max().item()}"
        # the writer: conv1x1 of a window batch straight into the layout == conv1x1 into a dense batch, window by window
        cin = 8
        x = torch.randn((nwin, cin) + tuple(roi), generator=gen)
        nrm = _rand_nrm(nwin, cin, gen)
        wgt = torch.randn((k, cin), generator=gen) / np.sqrt(cin)
        b = torch.randn(k, generator=gen) * 0.1
        dense = torch.empty((nwin, k) + tuple(roi), device=device)
        ops.conv1x1(x.to(device), nrm.to(device), wgt.to(device), b.to(device), dense)
        mos.flat.fill_(float("nan"))
        half = nwin // 2
        ops.conv1x1_windows(x[:half].to(device), nrm[:half].to(device), wgt.to(device), b.to(device), mos, 0)
        if nwin - half:
            ops.conv1x1_windows(x[half:].to(device), nrm[half:].to(device), wgt.to(device), b.to(device), mos, half)
        for w in range(nwin):
            assert torch.equal(mos.window_view(w).cpu(), dense[w].cpu()), (img, w)
    # grids the layout does not take
    assert not ops.LogitsMosaic.supported([[0, 1, 3, 5], [0, 6], [0, 4, 12]], (4, 6, 8), 3)            # irregular starts
    assert not ops.LogitsMosaic.supported([[0, 2, 4, 6, 7], [0], [0]], (16, 8, 8), 3)                   # step 2, roi 16: more than 4 classes
    assert not ops.LogitsMosaic.supported([[0], [0], [0, 6, 10]], (8, 8, 10), 3)                        # x extents not divisible by 4
