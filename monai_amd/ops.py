"""Thin typed wrappers over the C ABI (include/monai_amd.h): torch tensors in, kernel launches out.

Each function checks devices/dtypes, builds the ``mh_tensor5`` view descriptors and enqueues on torch's
current HIP stream.  No arithmetic happens in Python.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib

Grid = Sequence[Sequence[int]]  # per-axis ascending window starts (z, y, x)


def _s(t):
    return _lib.stream_ptr(t)


def window_extract(vol: torch.Tensor, grid: Grid, w0: int, nwin: int, roi: Sequence[int], out: torch.Tensor) -> torch.Tensor:
    """vol [C,D,H,W] -> out [nwin,C,rd,rh,rw]: windows w0..w0+nwin-1 of the dense window grid
    (reference: ``torch.cat([inputs[win_slice] ...])``, monai/inferers/utils.py:217-224)."""
    _lib.require_device(vol, out)
    if vol.dim() != 4 or not vol.is_contiguous() or not out.is_contiguous():
        raise RuntimeError("monai_amd.window_extract: vol must be a contiguous [C,D,H,W] tensor")
    c, d, h, w = vol.shape
    sz, sy, sx = grid
    _lib.lib().call(
        "mh_window_extract_f32", _lib.ptr(vol), c, d, h, w, _lib.int_array(sz), len(sz), _lib.int_array(sy), len(sy),
        _lib.int_array(sx), len(sx), int(w0), int(nwin), int(roi[0]), int(roi[1]), int(roi[2]), _lib.ptr(out), _s(vol),
    )
    return out


def _window_rows(logits: torch.Tensor, k: int, roi, nwin: int, who: str) -> int:
    """logits [nwin, K, rd, rh, rw]: every window's [K, rd, rh, rw] block dense, windows `stride(0)` floats apart (a padded
    window stride spreads the blend's concurrent read streams over the HBM channels) -> that stride"""
    if logits.dim() != 5 or logits.shape[0] != nwin or logits.shape[1] != k or tuple(logits.shape[2:]) != tuple(roi):
        raise RuntimeError(f"monai_amd.{who}: logits shape {tuple(logits.shape)} does not match the window grid")
    if not logits[0].is_contiguous() or (nwin > 1 and logits.stride(0) < logits[0].numel()):
        raise RuntimeError(f"monai_amd.{who}: each window's logits must be dense (strides {logits.stride()})")
    return int(logits.stride(0)) if nwin > 1 else 0


def sw_blend(logits: torch.Tensor, imp: torch.Tensor, out: torch.Tensor, grid: Grid, roi: Sequence[int], premultiplied: bool = False) -> torch.Tensor:
    """logits [nwin,K,rd,rh,rw] (all windows of the grid, in order; the window stride may be padded), imp [rd,rh,rw] ->
    out [K,D,H,W] = sum_w logits*imp / sum_w imp in the reference's summation order (monai/inferers/utils.py:264-298).
    premultiplied: `logits` already hold logit * weight (process_fn path); only the count uses `imp`."""
    _lib.require_device(logits, imp, out)
    if not (imp.is_contiguous() and out.is_contiguous()):
        raise RuntimeError("monai_amd.sw_blend: contiguous tensors required")
    k, d, h, w = out.shape
    sz, sy, sx = grid
    ws = _window_rows(logits, k, roi, len(sz) * len(sy) * len(sx), "sw_blend")
    _lib.lib().call(
        "mh_sw_blend_f32", _lib.ptr(logits), ws, _lib.ptr(imp), _lib.ptr(out), k, d, h, w, int(roi[0]), int(roi[1]), int(roi[2]),
        _lib.int_array(sz), len(sz), _lib.int_array(sy), len(sy), _lib.int_array(sx), len(sx), int(bool(premultiplied)), _s(out),
    )
    return out


def sw_blend_buffered(logits: torch.Tensor, imp: torch.Tensor, out: torch.Tensor, grid: Grid, roi: Sequence[int], buffer_axis: int, buffer_steps: int,
                      premultiplied: bool = False) -> torch.Tensor:
    """`sw_blend` in the summation order of the reference's buffered schedule (monai/inferers/utils.py:239-253, 276-284, 324-348): windows sorted by their
    start along `buffer_axis` (0 / 1 / 2 of the 3-D view), groups of `buffer_steps` distinct starts summed from zero and added to the output.
    premultiplied: `logits` already hold logit * weight (process_fn); only the count uses `imp`."""
    _lib.require_device(logits, imp, out)
    if not (imp.is_contiguous() and out.is_contiguous()):
        raise RuntimeError("monai_amd.sw_blend_buffered: contiguous tensors required")
    k, d, h, w = out.shape
    sz, sy, sx = grid
    ws = _window_rows(logits, k, roi, len(sz) * len(sy) * len(sx), "sw_blend_buffered")
    _lib.lib().call(
        "mh_sw_blend_buffered_f32", _lib.ptr(logits), ws, _lib.ptr(imp), _lib.ptr(out), k, d, h, w, int(roi[0]), int(roi[1]), int(roi[2]),
        _lib.int_array(sz), len(sz), _lib.int_array(sy), len(sy), _lib.int_array(sx), len(sx), int(buffer_axis), int(buffer_steps), int(bool(premultiplied)), _s(out),
    )
    return out


def sw_blend_argmax(logits: torch.Tensor, imp: torch.Tensor, labels: torch.Tensor, grid: Grid, roi: Sequence[int], num_classes: int,
                    premultiplied: bool = False) -> torch.Tensor:
    """The blend of ``sw_blend`` with ``torch.argmax(out, dim=0)`` fused into its epilogue: labels [D,H,W] (float32 or uint8) =
    index of the first maximal blended value per voxel (AsDiscrete(argmax=True), monai/transforms/post/array.py:132-237)."""
    _lib.require_device(logits, imp)
    _lib.require_device(labels, dtypes=(torch.float32, torch.uint8))
    if not (imp.is_contiguous() and labels.is_contiguous()):
        raise RuntimeError("monai_amd.sw_blend_argmax: contiguous tensors required")
    d, h, w = labels.shape
    k = int(num_classes)
    sz, sy, sx = grid
    ws = _window_rows(logits, k, roi, len(sz) * len(sy) * len(sx), "sw_blend_argmax")
    _lib.lib().call(
        "mh_sw_blend_argmax_f32", _lib.ptr(logits), ws, _lib.ptr(imp), _lib.ptr(labels), int(labels.dtype == torch.uint8), k, d, h, w,
        int(roi[0]), int(roi[1]), int(roi[2]), _lib.int_array(sz), len(sz), _lib.int_array(sy), len(sy), _lib.int_array(sx), len(sx),
        int(bool(premultiplied)), _s(labels),
    )
    return labels


def pointwise(op: str, src: torch.Tensor, param: float = 0.0) -> torch.Tensor:
    """sigmoid | threshold (x >= param -> 0/1) | round (half to even) on a contiguous fp32 tensor."""
    _lib.require_device(src)
    if not src.is_contiguous():
        raise RuntimeError("monai_amd.pointwise: contiguous tensor required")
    out = torch.empty_like(src)
    if src.numel():
        _lib.lib().call("mh_pointwise_f32", {"sigmoid": 0, "threshold": 1, "round": 2}[op], _lib.ptr(src), _lib.ptr(out), int(src.numel()), float(param), _s(src))
    return out


def channel_reduce(op: str, src: torch.Tensor) -> torch.Tensor:
    """argmax (-> [1, spatial] float) | softmax (-> like src) over axis 0 of a contiguous channel-first fp32 tensor."""
    _lib.require_device(src)
    if not src.is_contiguous() or src.dim() < 1:
        raise RuntimeError("monai_amd.channel_reduce: contiguous channel-first tensor required")
    c = int(src.shape[0])
    n = int(src.numel() // max(c, 1))
    out = torch.empty((1,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device) if op == "argmax" else torch.empty_like(src)
    if n:
        _lib.lib().call("mh_channel_reduce_f32", {"argmax": 0, "softmax": 1}[op], _lib.ptr(src), _lib.ptr(out), c, n, _s(src))
    return out


def onehot(labels: torch.Tensor, num_classes: int) -> torch.Tensor:
    """labels [1, spatial] fp32 -> [num_classes, spatial] fp32 of 0 / 1."""
    _lib.require_device(labels)
    if not labels.is_contiguous() or labels.shape[0] != 1:
        raise RuntimeError("monai_amd.onehot: contiguous [1, spatial] tensor required")
    out = torch.empty((int(num_classes),) + tuple(labels.shape[1:]), dtype=torch.float32, device=labels.device)
    if labels.numel():
        _lib.lib().call("mh_onehot_f32", _lib.ptr(labels), _lib.ptr(out), int(num_classes), int(labels.numel()), _s(labels))
    return out


def patch_accumulate(values: torch.Tensor, counts: torch.Tensor, patch: torch.Tensor, location: Sequence[int]) -> None:
    """AvgMerger.aggregate: values[..., loc:loc+size] += patch; counts[...] += 1 (values / patch fp32 [B, C, spatial 1-3],
    counts uint8 of the same shape as values)."""
    _lib.require_device(values, patch)
    _lib.require_device(counts, dtypes=(torch.uint8,))
    if not (values.is_contiguous() and counts.is_contiguous() and patch.is_contiguous()):
        raise RuntimeError("monai_amd.patch_accumulate: contiguous tensors required")
    sd = values.dim() - 2
    if sd < 1 or sd > 3 or patch.dim() != values.dim() or tuple(patch.shape[:2]) != tuple(values.shape[:2]) or counts.shape != values.shape:
        raise RuntimeError(f"monai_amd.patch_accumulate: shapes {tuple(values.shape)} / {tuple(counts.shape)} / {tuple(patch.shape)} do not match")
    pad = 3 - sd
    v3 = [1] * pad + [int(v) for v in values.shape[2:]]
    p3 = [1] * pad + [int(v) for v in patch.shape[2:]]
    l3 = [0] * pad + [int(v) for v in location]
    _lib.lib().call("mh_patch_accumulate_f32", _lib.ptr(values), _lib.ptr(counts), _lib.ptr(patch), int(values.shape[0] * values.shape[1]),
                    *v3, *p3, *l3, _s(values))


def patch_accumulate_batch(values: torch.Tensor, counts: torch.Tensor, patches: torch.Tensor, locations: Sequence[Sequence[int]]) -> None:
    """AvgMerger.aggregate for a whole batch in one launch: `patches` [npatch * B, C, spatial] (the network's output for npatch patches of a
    [B, C, ...] image, i.e. torch.cat order), `locations` npatch start tuples; same bits as npatch `patch_accumulate` calls in that order."""
    _lib.require_device(values, patches)
    _lib.require_device(counts, dtypes=(torch.uint8,))
    if not (values.is_contiguous() and counts.is_contiguous() and patches.is_contiguous()):
        raise RuntimeError("monai_amd.patch_accumulate_batch: contiguous tensors required")
    sd, npatch = values.dim() - 2, len(locations)
    if (sd < 1 or sd > 3 or patches.dim() != values.dim() or npatch < 1 or patches.shape[0] != npatch * values.shape[0] or patches.shape[1] != values.shape[1]
            or counts.shape != values.shape):
        raise RuntimeError(f"monai_amd.patch_accumulate_batch: shapes {tuple(values.shape)} / {tuple(counts.shape)} / {tuple(patches.shape)} x {npatch} do not match")
    pad = 3 - sd
    v3 = [1] * pad + [int(v) for v in values.shape[2:]]
    p3 = [1] * pad + [int(v) for v in patches.shape[2:]]
    flat = [int(v) for loc in locations for v in ([0] * pad + list(loc))]
    _lib.lib().call("mh_patch_accumulate_batch_f32", _lib.ptr(values), _lib.ptr(counts), _lib.ptr(patches), npatch, _lib.int_array(flat),
                    int(values.shape[0] * values.shape[1]), *v3, *p3, _s(values))


def avg_finalize(values: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """AvgMerger.finalize: values /= counts, in place."""
    _lib.require_device(values)
    _lib.require_device(counts, dtypes=(torch.uint8,))
    if not (values.is_contiguous() and counts.is_contiguous()) or counts.shape != values.shape:
        raise RuntimeError("monai_amd.avg_finalize: contiguous tensors of one shape required")
    _lib.lib().call("mh_avg_finalize_f32", _lib.ptr(values), _lib.ptr(counts), int(values.numel()), _s(values))
    return values


class LogitsMosaic:
    """The mosaic logits layout of the fused single-GPU sliding-window path (include/monai_amd.h: mh_sw_blend_mosaic_f32): per axis the windows
    fall into 2^log2m residue classes (+ one class for the last, clipped window); the logits of class (cz, cy, cx) are one dense array
    [K][cnt_z*rd][cnt_y*rh][cnt_x*rw] inside ONE flat allocation.  `supported(...)` says whether a window grid can use it."""

    CLASSES = 5

    @staticmethod
    def residue_log2(n: int, step: int, roi: int):
        """smallest power-of-two class count m with m * step >= roi (windows i, i + m do not overlap); None when more than 4 would be needed"""
        if n <= 2:
            return 0
        for lg in (0, 1, 2):
            if (step << lg) >= roi:
                return lg
        return None

    @classmethod
    def supported(cls, grid, roi, k: int) -> bool:
        if k < 1 or k > 8 or roi[2] % 4 or any(v % 4 for v in grid[2]):
            return False
        for starts, r in zip(grid, roi):
            n = len(starts)
            if n > 1 and any(starts[i] != i * starts[1] for i in range(n - 1)):
                return False
            if n > 2 and cls.residue_log2(n, starts[1], r) is None:
                return False
        return True

    def __init__(self, grid, roi, k: int, device, dtype=torch.float32, allocate: bool = True):
        self.grid, self.roi, self.k = [list(g) for g in grid], tuple(int(v) for v in roi), int(k)
        self.n = [len(g) for g in self.grid]
        self.log2m = [self.residue_log2(n, g[1] if n > 1 else 1, r) for n, g, r in zip(self.n, self.grid, self.roi)]
        self.cnt = []
        for n, lg in zip(self.n, self.log2m):
            c = (C.c_int32 * self.CLASSES)()
            _lib.lib().call("mh_sw_mosaic_class_counts", int(n), int(lg), c)
            self.cnt.append([int(v) for v in c])
        # class arrays one after the other, each start 256-byte aligned and shifted by a further 17 x 256 bytes (HBM channel spread, as the
        # padded window stride of the window-major buffer)
        self.base = [0] * self.CLASSES ** 3
        total = 0
        for cz in range(self.CLASSES):
            for cy in range(self.CLASSES):
                for cx in range(self.CLASSES):
                    size = self.k * self.cnt[0][cz] * self.roi[0] * self.cnt[1][cy] * self.roi[1] * self.cnt[2][cx] * self.roi[2]
                    self.base[(cz * self.CLASSES + cy) * self.CLASSES + cx] = total
                    if size:
                        total = (total + size + 3) // 4 * 4
                        if size * 4 >= (1 << 20):
                            total = (total + 63) // 64 * 64 + 17 * 64
        self.total, self._device, self._dtype = max(total, 4), device, dtype
        self.flat = None
        if allocate:
            self.allocate()

    def allocate(self):
        self.flat = torch.empty(self.total, dtype=self._dtype, device=self._device)
        return self

    def place(self, w: int):
        """(float offset, channel stride, z stride, y stride) of window w (row-major window index)"""
        nz, ny, nx = self.n
        idx = (w // (ny * nx), (w // nx) % ny, w % nx)
        cls, j = [], []
        for a in range(3):
            last = idx[a] == self.n[a] - 1
            m = 1 << self.log2m[a]
            cls.append(m if last else idx[a] & (m - 1))
            j.append(0 if last else idx[a] >> self.log2m[a])
        dc, hc, wc = (self.cnt[a][cls[a]] * self.roi[a] for a in range(3))
        off = self.base[(cls[0] * self.CLASSES + cls[1]) * self.CLASSES + cls[2]] + ((j[0] * self.roi[0]) * hc + j[1] * self.roi[1]) * wc + j[2] * self.roi[2]
        return off, dc * hc * wc, hc * wc, wc

    def window_view(self, w: int) -> torch.Tensor:
        """[K, rd, rh, rw] strided view of window w's logits"""
        off, sc, sd, sh = self.place(w)
        return self.flat.as_strided((self.k,) + self.roi, (sc, sd, sh, 1), off)

    def places(self, w0: int, n: int):
        flat = [int(v) for w in range(w0, w0 + n) for v in self.place(w)]
        return (C.c_int64 * len(flat))(*flat)


def sw_blend_mosaic(mosaic: LogitsMosaic, imp: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """the blend of `sw_blend` over the mosaic logits layout: same arithmetic in the same order, identical bits.  `imp`: the [rd, rh, rw] importance
    map, or its factors as ONE 1-D tensor [gz | gy | gx | floor] (rd + rh + rw + 1 floats; see include/monai_amd.h)"""
    _lib.require_device(mosaic.flat, imp, out)
    if not (imp.is_contiguous() and out.is_contiguous()):
        raise RuntimeError("monai_amd.sw_blend_mosaic: contiguous tensors required")
    factored = imp.dim() == 1
    if factored and imp.numel() != sum(mosaic.roi) + 1:
        raise RuntimeError("monai_amd.sw_blend_mosaic: factored importance map must hold rd + rh + rw + 1 floats")
    k, d, h, w = out.shape
    sz, sy, sx = mosaic.grid
    _lib.lib().call("mh_sw_blend_mosaic_f32", _lib.ptr(mosaic.flat), (C.c_int64 * len(mosaic.base))(*mosaic.base), *[int(v) for v in mosaic.log2m], _lib.ptr(imp),
                    int(factored), _lib.ptr(out), k, d, h, w, *mosaic.roi, _lib.int_array(sz), len(sz), _lib.int_array(sy), len(sy), _lib.int_array(sx), len(sx), _s(out))
    return out


def conv1x1_windows(x, x_nrm, weight, bias, mosaic: LogitsMosaic, w0: int):
    """conv1x1 of a batch of windows written straight into the mosaic logits layout: batch element i is window w0 + i"""
    _lib.require_device(x, x_nrm, weight, bias, mosaic.flat)
    xi = _lib.tensor5(x, x_nrm)
    _lib.lib().call("mh_conv1x1_windows_f32", C.byref(xi), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(mosaic.flat), int(weight.shape[0]),
                    mosaic.places(int(w0), int(x.shape[0])), _s(x))


def conv3d_k3_select(cin: int, cout: int, d: int, h: int, w: int, bounded: bool = False, algo: Optional[int] = None) -> int:
    """Kernel configuration for a 3x3x3 convolution.  `bounded`: every record of the input view carries a magnitude bound
    (written by instnorm / groupnorm_finalize, or by a raw producer into `nrm_identity` records) -- only then may the fp16
    split-precision configuration be returned.  `algo`: MH_ALGO_* family, default `monai_amd.config.conv_algo()`."""
    from . import config

    cfg = _lib.lib().query("mh_conv3d_k3_select", config.conv_algo() if algo is None else int(algo), int(bool(bounded)), cin, cout, d, h, w)
    if not config.small_volume_h2() and cfg == _lib.lib().query("mh_conv3d_k3_h2v_config"):      # host-side switch (measurements): what the selector returned before the small-volume kernel existed
        cfg = _lib.lib().query("mh_conv3d_k3_select", config.CONV_ALGOS["fp32"], int(bool(bounded)), cin, cout, d, h, w)
    return cfg


def conv3d_k3_num_configs() -> int:
    """Highest configuration id (= the Winograd configuration; 1 .. n-1 are the direct implicit-GEMM tiles, 0 the plain kernel)."""
    return _lib.lib().query("mh_conv3d_k3_num_configs")


def conv3d_k3_h2_config() -> int:
    """Id of the fp16 two-piece split-precision configuration (fp16 matrix cores, hi + lo pieces per operand, three piece
    products, fp32 accumulation: fp32-equivalent results); outside 1 .. conv3d_k3_num_configs()."""
    return _lib.lib().query("mh_conv3d_k3_h2_config")


def conv3d_k3_h2c_config() -> int:
    """Id of the split-precision configuration with output channel groups of 16 (two z-taps per 32-column matrix instruction): what `conv3d_k3_select` returns for
    bounded inputs of layers with 16 output channels; same tolerance class as `conv3d_k3_h2_config`."""
    return _lib.lib().query("mh_conv3d_k3_h2c_config")


def conv3d_k3_c1_config() -> int:
    """Id of the one-input-channel configuration (first layer of the networks: packed fp32 vector arithmetic, write-bound, exact fp32);
    outside 1 .. conv3d_k3_num_configs()."""
    return _lib.lib().query("mh_conv3d_k3_c1_config")


def conv3d_k3_h2v_config() -> int:
    """Id of the split-precision configuration for small volumes (one sample's whole D x H x W <= 256 voxels as the workgroup's tile: the 6^3 level of a 96^3 window);
    same tolerance class as `conv3d_k3_h2_config`."""
    return _lib.lib().query("mh_conv3d_k3_h2v_config")


def conv3d_k3_h2w_config() -> int:
    """Id of the split-precision configuration behind an in-plane Winograd F(2x2, 3x3) transform (32 input channels, planes of whole 4 x 16 regions;
    `conv3d_k3_h2w_fits`); same tolerance class as `conv3d_k3_h2_config`."""
    return _lib.lib().query("mh_conv3d_k3_h2w_config")


def conv3d_k3_h2w_fits(d: int, h: int, w: int) -> bool:
    return bool(_lib.lib().query("mh_conv3d_k3_h2w_fits", int(d), int(h), int(w)))


def conv3d_k3_accepts(cfg: int, cin: int, cout: int) -> bool:
    return bool(_lib.lib().query("mh_conv3d_k3_accepts", int(cfg), int(cin), int(cout)))


def conv3d_k3_pack(cfg: int, weight: torch.Tensor) -> torch.Tensor:
    """torch conv weight [Cout,Cin,3,3,3] -> the packed layout of configuration `cfg`."""
    _lib.require_device(weight)
    cout, cin = weight.shape[:2]
    wc = weight.detach().contiguous()
    packed = torch.empty(_lib.lib().query("mh_conv3d_k3_packed_floats", cfg, cin, cout), dtype=torch.float32, device=weight.device)
    _lib.lib().call("mh_conv3d_k3_pack_f32", cfg, _lib.ptr(wc), cin, cout, _lib.ptr(packed), _s(weight))
    return packed


def conv3d_k3_stat_tiles(cfg: int, d: int, h: int, w: int) -> int:
    return _lib.lib().query("mh_conv3d_k3_stat_tiles", cfg, d, h, w)


def conv3d_k3(cfg, x, x_nrm, packed_w, bias, out, stats: Optional[torch.Tensor] = None, accumulate: bool = False):
    """out = conv3x3x3(act(x)) + bias; optionally emits the fused InstanceNorm statistics records.  accumulate=True: out += ... (the split-precision configuration
    with records and statistics only; the statistics are those of the sum)"""
    _lib.require_device(x, x_nrm, packed_w, bias, out, stats)
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out)
    _lib.lib().call("mh_conv3d_k3_accumulate_f32" if accumulate else "mh_conv3d_k3_f32", int(cfg), C.byref(xi), _lib.ptr(packed_w), _lib.ptr(bias), C.byref(xo),
                    _lib.ptr(stats), _s(x))
    return out


def conv3d_k3_pool_accepts(cfg: int, cin: int, cout: int, d: int, h: int, w: int) -> bool:
    return bool(_lib.lib().query("mh_conv3d_k3_pool_accepts", int(cfg), int(cin), int(cout), int(d), int(h), int(w)))


def conv3d_k3_pool(cfg, x, x_nrm, packed_w, bias, out, stats, pool_max: torch.Tensor, pool_min: torch.Tensor):
    """`conv3d_k3` that also writes the 2 x 2 x 2 maxima / minima of its raw output (pool_max / pool_min [N, C, D/2, H/2, W/2]): MaxPool3d(2) of the next block without a pass
    over the full-resolution tensor.  After the output's records exist: `pool_select(pool_max, pool_min, out_nrm)`, then consumers read pool_max under out_nrm."""
    _lib.require_device(x, x_nrm, packed_w, bias, out, stats, pool_max, pool_min)
    if pool_max.shape != pool_min.shape or pool_max.stride() != pool_min.stride() or not pool_max[0].is_contiguous():
        raise RuntimeError("monai_amd.conv3d_k3_pool: pool_max / pool_min must be two equally laid out [N, C, D/2, H/2, W/2] tensors, dense per sample")
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out)
    _lib.lib().call("mh_conv3d_k3_pool_f32", int(cfg), C.byref(xi), _lib.ptr(packed_w), _lib.ptr(bias), C.byref(xo), _lib.ptr(stats), _lib.ptr(pool_max), _lib.ptr(pool_min),
                    int(pool_max.stride(0)), _s(x))
    return out


def pool_select(pool_max: torch.Tensor, pool_min: torch.Tensor, nrm: torch.Tensor) -> torch.Tensor:
    """channels whose record has alpha < 0 take their minima (see conv3d_k3_pool); nrm: the [N, C, 4] records the consumer will apply"""
    _lib.require_device(pool_max, pool_min, nrm)
    n, c = int(pool_max.shape[0]), int(pool_max.shape[1])
    vol = int(pool_max.shape[2] * pool_max.shape[3] * pool_max.shape[4])
    if nrm.dim() != 3 or nrm.shape[2] != 4 or nrm.stride(2) != 1 or nrm.stride(1) != 4:
        raise RuntimeError("monai_amd.pool_select: nrm must be an [N, C, 4] record view")
    _lib.lib().call("mh_pool_select_f32", _lib.ptr(pool_max), _lib.ptr(pool_min), _lib.ptr(nrm), int(nrm.stride(0)), n, c, int(pool_max.stride(0)), vol, _s(pool_max))
    return pool_max


def instnorm_stat_tiles(d: int, h: int, w: int) -> int:
    return _lib.lib().query("mh_instnorm_stat_tiles", d, h, w)


def instnorm_stats(x: torch.Tensor, stats: torch.Tensor):
    _lib.require_device(x, stats)
    xi = _lib.tensor5(x)
    _lib.lib().call("mh_instnorm_stats_f32", C.byref(xi), _lib.ptr(stats), _s(x))
    return stats


def instnorm_finalize(stats, tiles: int, n: int, c: int, gamma, beta, eps: float, slope: float, nrm: torch.Tensor):
    """Merge `tiles` records per (n, c); write {alpha, beta, slope, bound} into nrm ([N, C, 4] slice)."""
    _lib.require_device(stats, gamma, beta, nrm)
    if nrm.dim() != 3 or nrm.shape[2] != 4 or nrm.stride(2) != 1 or nrm.stride(1) != 4:
        raise RuntimeError("monai_amd.instnorm_finalize: nrm must be a [N,C,4] slice")
    ns = nrm.stride(0) if n > 1 else max(nrm.stride(0), 4 * c)
    _lib.lib().call(
        "mh_instnorm_finalize_f32", _lib.ptr(stats), int(tiles), int(n), int(c), _lib.ptr(gamma), _lib.ptr(beta), float(eps),
        float(slope), _lib.ptr(nrm), int(ns), _s(stats),
    )
    return nrm


def nrm_identity(nrm: torch.Tensor) -> torch.Tensor:
    """nrm [N, C, 4] slice <- identity records {1, 0, 1, FLT_MIN}: hand it as `out_nrm` to a raw producer (deconv_k2s2, deconv_ks,
    add_act) and the kernel leaves max |value written| per (n, c) in the 4th component -- the magnitude bound the split-precision
    convolution scales its input by."""
    _lib.require_device(nrm)
    if nrm.dim() != 3 or nrm.shape[2] != 4 or nrm.stride(2) != 1 or nrm.stride(1) != 4:
        raise RuntimeError("monai_amd.nrm_identity: nrm must be a [N,C,4] slice")
    n, c = nrm.shape[:2]
    _lib.lib().call("mh_nrm_identity_f32", _lib.ptr(nrm), int(n), int(c), int(nrm.stride(0) if n > 1 else max(nrm.stride(0), 4 * c)), _s(nrm))
    return nrm


def maxpool2(x, x_nrm, out, out_nrm=None):
    """out_nrm: identity records for `out` carrying the input's magnitude bounds (written by the kernel)"""
    _lib.require_device(x, x_nrm, out, out_nrm)
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out, out_nrm)
    _lib.lib().call("mh_maxpool2_f32", C.byref(xi), C.byref(xo), _s(x))
    return out


_DECONV_H2_PACKED: dict = {}          # id(weight) -> (weakref to the parameter, its version, device, storage address, packed tap matrices): re-packed when the parameter changed, moved or got new storage (`param.data = other` keeps `_version`)


def deconv_k2s2_h2_packed(weight: torch.Tensor) -> torch.Tensor:
    """[Cin, Cout, 2, 2, 2] -> the matrix-core kernel's split-precision (cout, parity) row matrices (kernels/deconv_h2.h), cached per parameter version"""
    import weakref

    hit = _DECONV_H2_PACKED.get(id(weight))
    if hit is None or hit[0]() is not weight or hit[1] != weight._version or hit[2] != str(weight.device) or hit[3] != weight.data_ptr():
        if len(_DECONV_H2_PACKED) > 256:          # parameters that no longer exist
            for k in [k for k, v in _DECONV_H2_PACKED.items() if v[0]() is None]:
                del _DECONV_H2_PACKED[k]
        cin, cout = int(weight.shape[0]), int(weight.shape[1])
        packed = torch.zeros(_lib.lib().query("mh_deconv_k2s2_h2_packed_floats", cin, cout), dtype=torch.float32, device=weight.device)
        _lib.lib().call("mh_deconv_k2s2_h2_pack_f32", _lib.ptr(weight.detach().contiguous()), cin, cout, _lib.ptr(packed), _s(weight))
        hit = (weakref.ref(weight), weight._version, str(weight.device), weight.data_ptr(), packed)
        _DECONV_H2_PACKED[id(weight)] = hit
    return hit[4]


def deconv_k2s2(x, x_nrm, weight, bias, out, out_nrm=None, bounded: bool = False):
    """out = conv_transpose3d(act(x), k 2, s 2) + bias.  out_nrm: `nrm_identity` records of `out`; the kernel folds max |value written| into their bound.
    bounded: every record of `x_nrm` carries a magnitude bound -- the transposed convolution then runs on the fp16 matrix cores in split precision
    (kernels/deconv_h2.h; shapes it takes, families "auto" / "h2", `monai_amd.config.deconv_h2()`), otherwise on the direct fp32 kernel"""
    _lib.require_device(x, x_nrm, weight, bias, out, out_nrm)
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out, out_nrm)
    if bounded and x_nrm is not None and tuple(weight.shape[2:]) == (2, 2, 2):
        from . import config

        cin, cout = int(weight.shape[0]), int(weight.shape[1])
        if config.deconv_h2() and _lib.lib().query("mh_deconv_k2s2_h2_accepts", cin, cout, int(x.shape[2]), int(x.shape[3]), int(x.shape[4])):
            _lib.lib().call("mh_deconv_k2s2_h2_f32", C.byref(xi), _lib.ptr(deconv_k2s2_h2_packed(weight)), _lib.ptr(bias), C.byref(xo), _s(x))
            return out
    _lib.lib().call("mh_deconv_k2s2_f32", C.byref(xi), _lib.ptr(weight), _lib.ptr(bias), C.byref(xo), _s(x))
    return out


def upconv_k4s2_accepts(cin: int, cout: int, dl: int, hl: int, wl: int) -> bool:
    """can `upconv_k4s2_accum` serve a deconvolution input of `cin` channels and extents (dl, hl, wl) feeding a convolution with `cout` outputs?"""
    return bool(_lib.lib().query("mh_upconv_k4s2_accepts", int(cin), int(cout), int(dl), int(hl), int(wl)))


def upconv_k4s2_weights(deconv_w: torch.Tensor, deconv_b: Optional[torch.Tensor], conv_w_up: torch.Tensor):
    """UpCat's `conv3(cat([x_e, deconv2(x)]))`: the composite of the two layers acting on x (kernels/upconv_h2.h).
    deconv_w [Cin][Cup][2][2][2], deconv_b [Cup] or None, conv_w_up [Cout][Cup][3][3][3] (the convolution's weights for the up channels)
    -> (w4 [Cin][Cout][4][4][4], bias_table [27][Cout]), fp32 on the parameters' device; per axis W4[u + 1] = sum over (d, k) with d - k + 1 == u of Wd[d] x Wc[k],
    the table entry of position class (cz, cy, cx) in {first, interior, last}^3 = sum_cup b[cup] x sum of the taps of Wc that stay inside the volume there."""
    wd, wc = deconv_w.detach().float(), conv_w_up.detach().float()
    cin, cout = wd.shape[0], wc.shape[0]
    pairs = {-1: ((0, 2),), 0: ((0, 1), (1, 2)), 1: ((0, 0), (1, 1)), 2: ((1, 0),)}
    w4 = torch.zeros((cin, cout, 4, 4, 4), dtype=torch.float32, device=wd.device)
    for uz, pz_ in pairs.items():
        for uy, py_ in pairs.items():
            for ux, px_ in pairs.items():
                acc = None
                for dz, kz in pz_:
                    for dy, ky in py_:
                        for dx, kx in px_:
                            t = wd[:, :, dz, dy, dx] @ wc[:, :, kz, ky, kx].t()          # [Cin, Cup] x [Cup, Cout]
                            acc = t if acc is None else acc + t
                w4[:, :, uz + 1, uy + 1, ux + 1] = acc
    inside = ((1, 2), (0, 1, 2), (0, 1))                  # the convolution's taps that stay inside at the first / an interior / the last position of an axis
    table = torch.zeros((27, cout), dtype=torch.float32, device=wd.device)
    if deconv_b is not None:
        bd = deconv_b.detach().float()
        for cz in range(3):
            for cy in range(3):
                for cx in range(3):
                    wsum = wc[:, :, list(inside[cz])][:, :, :, list(inside[cy])][:, :, :, :, list(inside[cx])].sum(dim=(2, 3, 4))      # [Cout, Cup]
                    table[(cz * 3 + cy) * 3 + cx] = wsum @ bd
    return w4.contiguous(), table.contiguous()


def upconv_k4s2_pack(w4: torch.Tensor) -> torch.Tensor:
    """composite weights [Cin][Cout][4][4][4] -> the kernel's split-precision tap matrices (once per parameter version)"""
    _lib.require_device(w4)
    cin, cout = int(w4.shape[0]), int(w4.shape[1])
    packed = torch.zeros(_lib.lib().query("mh_upconv_k4s2_packed_floats", cin, cout), dtype=torch.float32, device=w4.device)
    _lib.lib().call("mh_upconv_k4s2_pack_f32", _lib.ptr(w4.contiguous()), cin, cout, _lib.ptr(packed), _s(w4))
    return packed


def upconv_k4s2_stat_tiles(dl: int, hl: int, wl: int) -> int:
    return _lib.lib().query("mh_upconv_k4s2_stat_tiles", int(dl), int(hl), int(wl))


def upconv_k4s2(low, low_nrm, packed, bias_table, out, accumulate: bool = False, stats=None):
    """accumulate=False: out = convT(k4, s2, p1)(act(low)) + bias_table[position class] (the accumulating convolution adds the skip half: `conv3d_k3(..., accumulate=True)`);
    accumulate=True: added to `out` in place (`out`: the raw skip half of the convolution); then with `stats` ([N * Cout * upconv_k4s2_stat_tiles(*low.shape[2:]) * 3]
    floats) the InstanceNorm statistics of the sum"""
    _lib.require_device(low, low_nrm, packed, bias_table, out, stats)
    xi, xo = _lib.tensor5(low, low_nrm), _lib.tensor5(out)
    _lib.lib().call("mh_upconv_k4s2_f32", C.byref(xi), _lib.ptr(packed), _lib.ptr(bias_table), C.byref(xo), int(bool(accumulate)), _lib.ptr(stats), _s(out))
    return out


def conv1x1_stat_tiles(d: int, h: int, w: int) -> int:
    return _lib.lib().query("mh_conv1x1_stat_tiles", d, h, w)


def conv1x1(x, x_nrm, weight, bias, out, stats=None):
    """1x1x1 convolution of act(x); with `stats` ([N * Cout * conv1x1_stat_tiles(D, H, W) * 3] floats) the kernel also leaves the InstanceNorm statistics of `out`"""
    _lib.require_device(x, x_nrm, weight, bias, out, stats)
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out)
    if stats is None:
        _lib.lib().call("mh_conv1x1_f32", C.byref(xi), _lib.ptr(weight), _lib.ptr(bias), C.byref(xo), _s(x))
    else:
        _lib.lib().call("mh_conv1x1_stats_f32", C.byref(xi), _lib.ptr(weight), _lib.ptr(bias), C.byref(xo), _lib.ptr(stats), _s(x))
    return out


def conv1x1_h2_accepts(cin: int, cout: int, d: int, h: int, w: int) -> bool:
    return bool(_lib.lib().query("mh_conv1x1_h2_accepts", int(cin), int(cout), int(d), int(h), int(w)))


def conv1x1_h2_wanted(cin: int, cout: int, d: int, h: int, w: int) -> bool:
    """the split-precision 1x1x1 kernel takes the shape and the session has not pinned an exact-fp32 convolution family (config.CONV_ALGO / MONAI_AMD_CONV_ALGO)"""
    from . import config

    return config.conv_algo() in (config.CONV_ALGOS["auto"], config.CONV_ALGOS["h2"]) and conv1x1_h2_accepts(cin, cout, d, h, w)


def conv1x1_h2_pack(weight: torch.Tensor) -> torch.Tensor:
    """weight [Cout, Cin] -> the packed split-precision slabs of `conv1x1_h2` (once per layer)"""
    _lib.require_device(weight)
    if weight.dim() != 2:
        raise RuntimeError(f"monai_amd.conv1x1_h2_pack: weight must be [Cout, Cin], got {tuple(weight.shape)}")
    cout, cin = (int(v) for v in weight.shape)
    packed = torch.empty(_lib.lib().query("mh_conv1x1_h2_packed_floats", cin, cout), dtype=torch.float32, device=weight.device)
    _lib.lib().call("mh_conv1x1_h2_pack_f32", _lib.ptr(weight.contiguous()), cout, cin, _lib.ptr(packed), _s(weight))
    return packed


def conv1x1_h2(x, x_nrm, packed, bias, out, stats=None):
    """1x1x1 convolution of act(x) with all output channels from one read of x (fp16 matrix cores, split precision: fp32-equivalent); x_nrm must carry magnitude
    bounds.  stats as `conv1x1` (same tile count)."""
    _lib.require_device(x, x_nrm, packed, bias, out, stats)
    if x_nrm is None:
        raise RuntimeError("monai_amd.conv1x1_h2: the input needs records with magnitude bounds")
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out)
    _lib.lib().call("mh_conv1x1_h2_f32", C.byref(xi), _lib.ptr(packed), _lib.ptr(bias), C.byref(xo), _lib.ptr(stats), _s(x))
    return out


_MODES = {"nearest": 0, "bilinear": 1, "linear": 1, "trilinear": 1}
_PADS = {"zeros": 0, "border": 1, "reflection": 2}


def affine_resample(src: torch.Tensor, m, out_size: Sequence[int], mode: str, padding_mode: str, align_corners: bool, compute_f64: bool):
    """src [NC, Di, Hi, Wi] fp32 -> [NC, Do, Ho, Wo] fp32, sampling at  index = m @ (oz, oy, ox, 1)  (m: 3x4, fp64, host)."""
    _lib.require_device(src)
    if src.dim() != 4 or not src.is_contiguous():
        raise RuntimeError("monai_amd.affine_resample: src must be a contiguous [NC, D, H, W] tensor")
    nc, di, hi, wi = src.shape
    do, ho, wo = (int(v) for v in out_size)
    out = torch.empty((nc, do, ho, wo), dtype=torch.float32, device=src.device)
    mm = (C.c_double * 12)(*[float(v) for v in m])
    ws = torch.empty(_lib.lib().query("mh_affine_resample_workspace_bytes", do, ho, wo), dtype=torch.uint8, device=src.device)
    _lib.lib().call("mh_affine_resample_f32", _lib.ptr(src), nc, di, hi, wi, _lib.ptr(out), do, ho, wo, mm, _MODES[mode], _PADS[padding_mode],
                    int(bool(align_corners)), int(bool(compute_f64)), _lib.ptr(ws), _s(src))
    return out


def grid_resample(src: torch.Tensor, coords: torch.Tensor, mode: str, padding_mode: str, align_corners: bool, compute_f64: bool,
                  scale=(1.0, 1.0, 1.0), offset=(0.0, 0.0, 0.0)):
    """src [NC, Di, Hi, Wi] fp32, coords [3, Do, Ho, Wo] (z, y, x planes; fp32/fp64) -> [NC, Do, Ho, Wo] fp32; the source
    index along axis a is scale[a] * coords[a] + offset[a]."""
    _lib.require_device(src)
    _lib.require_device(coords, dtypes=(torch.float32, torch.float64))
    if src.dim() != 4 or coords.dim() != 4 or coords.shape[0] != 3 or not src.is_contiguous() or not coords.is_contiguous():
        raise RuntimeError("monai_amd.grid_resample: src [NC,D,H,W] and coords [3,Do,Ho,Wo] must be contiguous")
    nc, di, hi, wi = src.shape
    _, do, ho, wo = coords.shape
    out = torch.empty((nc, do, ho, wo), dtype=torch.float32, device=src.device)
    sc = (C.c_double * 3)(*[float(v) for v in scale])
    of = (C.c_double * 3)(*[float(v) for v in offset])
    _lib.lib().call("mh_grid_resample_f32", _lib.ptr(src), nc, di, hi, wi, _lib.ptr(coords), int(coords.dtype == torch.float64), sc, of,
                    _lib.ptr(out), do, ho, wo, _MODES[mode], _PADS[padding_mode], int(bool(align_corners)), int(bool(compute_f64)), _s(src))
    return out


def separable_filter3d(src: torch.Tensor, kernels) -> torch.Tensor:
    """src [NC, D, H, W] fp32; kernels = (kz, ky, kx) 1-D float sequences with odd lengths; zero padding."""
    _lib.require_device(src)
    if src.dim() != 4 or not src.is_contiguous():
        raise RuntimeError("monai_amd.separable_filter3d: src must be a contiguous [NC, D, H, W] tensor")
    out = torch.empty_like(src)
    arrs = [(C.c_float * len(k))(*[float(v) for v in k]) for k in kernels]
    nc, d, h, w = src.shape
    _lib.lib().call("mh_separable_filter3d_f32", _lib.ptr(src), _lib.ptr(out), nc, d, h, w, arrs[0], len(kernels[0]), arrs[1], len(kernels[1]),
                    arrs[2], len(kernels[2]), _s(src))
    return out


def add_act(a, a_nrm, b, b_nrm, slope: float, out, out_nrm=None):
    """out = leaky_relu(act(a) + act(b), slope) -- residual join of UnetResBlock.  out_nrm: `nrm_identity` records of `out` (magnitude bound)."""
    _lib.require_device(a, a_nrm, b, b_nrm, out, out_nrm)
    ta, to = _lib.tensor5(a, a_nrm), _lib.tensor5(out, out_nrm)
    tb = None if b is None else _lib.tensor5(b, b_nrm)
    _lib.lib().call("mh_add_act_f32", C.byref(ta), None if tb is None else C.byref(tb), float(slope), C.byref(to), _s(a))
    return out


def conv1x1_sum2_accepts(cout: int, d: int, h: int, w: int) -> bool:
    return bool(_lib.lib().query("mh_conv1x1_sum2_accepts", int(cout), int(d), int(h), int(w)))


def conv1x1_sum2(a, a_nrm, b, b_nrm, slope: float, weight, bias, out):
    """out = conv1x1(leaky_relu(act(a) + act(b), slope)) + bias: the residual join of UnetResBlock inside the output convolution that reads it (no joined tensor);
    the same bits as add_act followed by conv1x1.  weight [Cout <= 8, Cin]."""
    _lib.require_device(a, a_nrm, b, b_nrm, weight, bias, out)
    ta, tb, to = _lib.tensor5(a, a_nrm), _lib.tensor5(b, b_nrm), _lib.tensor5(out)
    _lib.lib().call("mh_conv1x1_sum2_f32", C.byref(ta), C.byref(tb), float(slope), _lib.ptr(weight), _lib.ptr(bias), C.byref(to), _s(a))
    return out


def pad_replicate(x, out, x_nrm=None, out_nrm=None):
    """out[z, y, x] = x[min(z, D-1), min(y, H-1), min(x, W-1)]: replicate padding at the far end (UpCat's odd-edge case).  A raw copy:
    x_nrm is read for its magnitude bounds only, which go into the identity records out_nrm."""
    _lib.require_device(x, out, x_nrm, out_nrm)
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out, out_nrm)
    _lib.lib().call("mh_pad_replicate_f32", C.byref(xi), C.byref(xo), _s(x))
    return out


def pixelshuffle(x, out, fz: int = 2, pad_pool: bool = True, out_nrm=None):
    """SubpixelUpsample behind its convolution (scale factor 2): x [N, C * fz * 4, D, H, W] raw -> out [N, C, fz * D, 2 H, 2 W], then (pad_pool) zero padding in front of every
    spatial axis + average pooling 2, stride 1.  out_nrm: `nrm_identity` records of `out`; the kernel folds max |value written| into their bound."""
    _lib.require_device(x, out, out_nrm)
    xi, xo = _lib.tensor5(x), _lib.tensor5(out, out_nrm)
    _lib.lib().call("mh_pixelshuffle_f32", C.byref(xi), C.byref(xo), int(fz), int(bool(pad_pool)), _s(x))
    return out


def attention(qkv: torch.Tensor, heads: int, scale: float, head_dim: int = 64) -> torch.Tensor:
    """qkv [B, S, 3*heads*head_dim] -> [B, S, heads*head_dim] = softmax(Q K^T * scale) V per head (fp16 matrix cores, split precision: fp32-equivalent;
    any sequence length; head_dim 32 / 64 / 96 / 128)."""
    _lib.require_device(qkv)
    if qkv.dim() != 3 or not qkv.is_contiguous() or qkv.shape[2] != 3 * heads * head_dim:
        raise RuntimeError(f"monai_amd.attention: qkv must be contiguous [B, S, {3 * heads * head_dim}], got {tuple(qkv.shape)}")
    b, s, _ = qkv.shape
    out = torch.empty((b, s, heads * head_dim), dtype=torch.float32, device=qkv.device)
    _lib.lib().call("mh_attention_f32", _lib.ptr(qkv), _lib.ptr(out), b, s, int(heads), int(head_dim), float(scale), _s(qkv))
    return out


def window_attention(qkv: torch.Tensor, heads: int, scale: float, bias_t: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                     exact: Optional[bool] = None) -> torch.Tensor:
    """qkv [BW, S, 3*heads*hd] -> [BW, S, heads*hd] = softmax((q*scale) k^T + bias[head] + mask[window % nW]) v per (window, head);
    bias_t [heads, S, S] holds the relative position bias transposed (key-major), mask [nW, S, S] the shifted-window mask.
    exact: keep the exact-fp32 kernel (default: when config.CONV_ALGO pins the exact-fp32 family); otherwise head dims 16 / 32 run in fp16 split precision."""
    _lib.require_device(qkv, bias_t, mask)
    if qkv.dim() != 3 or not qkv.is_contiguous() or qkv.shape[2] % (3 * heads):
        raise RuntimeError(f"monai_amd.window_attention: qkv must be contiguous [BW, S, 3*heads*hd], got {tuple(qkv.shape)}")
    bw, s, c3 = qkv.shape
    hd = c3 // (3 * heads)
    if hd not in (8, 16, 32):
        raise NotImplementedError(f"monai_amd.window_attention: head_dim {hd} is not on the HIP path (8, 16, 32 are)")
    for t, nm in ((bias_t, "bias_t"), (mask, "mask")):
        if t is not None and (not t.is_contiguous() or tuple(t.shape[1:]) != (s, s)):
            raise RuntimeError(f"monai_amd.window_attention: {nm} must be contiguous [*, {s}, {s}], got {tuple(t.shape)}")
    if bias_t is not None and bias_t.shape[0] != heads:
        raise RuntimeError("monai_amd.window_attention: bias_t must have one [S, S] table per head")
    nw = int(mask.shape[0]) if mask is not None else 1
    out = torch.empty((bw, s, heads * hd), dtype=torch.float32, device=qkv.device)
    if exact is None:
        from . import config

        exact = config.conv_algo() in (config.CONV_ALGOS["fp32"], config.CONV_ALGOS["direct"], config.CONV_ALGOS["wino2d"])
    _lib.lib().call("mh_window_attention_f32", _lib.ptr(qkv), _lib.ptr(bias_t), _lib.ptr(mask), _lib.ptr(out), bw, nw, s, int(heads), hd, float(scale), int(bool(exact)), _s(qkv))
    return out


def window_attention_rel_accepts(s: int, hd: int, table_rows: int) -> bool:
    return bool(_lib.lib().query("mh_window_attention_rel_accepts", int(s), int(hd), int(table_rows)))


def window_attention_rel(qkv: torch.Tensor, heads: int, scale: float, rel_table: torch.Tensor, coord: torch.Tensor, coord_off: int,
                         region: Optional[torch.Tensor] = None) -> torch.Tensor:
    """window_attention with bias[h][q][k] = rel_table[coord[q] - coord[k] + coord_off][h] and mask[w][q][k] = 0 where region[w][q] == region[w][k], else -100,
    evaluated inside the kernel (no S x S tables; split precision, head dims 16 / 32).  rel_table [rows, heads] fp32, coord [S] int32, region [nW, S] int32 or None."""
    _lib.require_device(qkv, rel_table)
    _lib.require_device(coord, region, dtypes=(torch.int32,))
    if qkv.dim() != 3 or not qkv.is_contiguous() or qkv.shape[2] % (3 * heads):
        raise RuntimeError(f"monai_amd.window_attention_rel: qkv must be contiguous [BW, S, 3*heads*hd], got {tuple(qkv.shape)}")
    bw, s, c3 = qkv.shape
    hd = c3 // (3 * heads)
    if rel_table.dim() != 2 or rel_table.shape[1] != heads or rel_table.dtype != torch.float32 or not rel_table.is_contiguous():
        raise RuntimeError(f"monai_amd.window_attention_rel: rel_table must be contiguous fp32 [rows, {heads}], got {tuple(rel_table.shape)}")
    if coord.dtype != torch.int32 or tuple(coord.shape) != (s,) or not coord.is_contiguous():
        raise RuntimeError(f"monai_amd.window_attention_rel: coord must be contiguous int32 [{s}]")
    if region is not None and (region.dtype != torch.int32 or region.dim() != 2 or region.shape[1] != s or not region.is_contiguous()):
        raise RuntimeError(f"monai_amd.window_attention_rel: region must be contiguous int32 [nW, {s}]")
    nw = int(region.shape[0]) if region is not None else 1
    out = torch.empty((bw, s, heads * hd), dtype=torch.float32, device=qkv.device)
    _lib.lib().call("mh_window_attention_rel_f32", _lib.ptr(qkv), _lib.ptr(rel_table), int(rel_table.shape[0]), _lib.ptr(coord), int(coord_off), _lib.ptr(region),
                    _lib.ptr(out), bw, nw, s, int(heads), hd, float(scale), _s(qkv))
    return out


def linear_pack(weight: torch.Tensor) -> torch.Tensor:
    """Pack an nn.Linear weight [N, K] for `linear` (two fp16 pieces per value, power-of-two layer scale, tile layout); once per layer."""
    _lib.require_device(weight)
    if weight.dim() != 2:
        raise RuntimeError(f"monai_amd.linear_pack: weight must be [N, K], got {tuple(weight.shape)}")
    n, k = (int(v) for v in weight.shape)
    packed = torch.empty(_lib.lib().query("mh_linear_packed_floats", n, k), dtype=torch.float32, device=weight.device)
    _lib.lib().call("mh_linear_pack_f32", _lib.ptr(weight.contiguous()), n, k, _lib.ptr(packed), _s(weight))
    return packed


def linear(x: torch.Tensor, packed_w: torch.Tensor, n_out: int, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           gelu: bool = False, out: Optional[torch.Tensor] = None, tile: int = 0) -> torch.Tensor:
    """y[..., N] = act(x[..., K] . W^T + bias) (+ residual) -- nn.Linear (+ nn.GELU()) (+ the residual sum of a transformer block) in one
    launch on the fp16 matrix cores in two-piece split precision (fp32-equivalent).  `packed_w` = linear_pack(W).  `tile`: workgroup tile
    (0 = chosen by the problem size; 64 / 128 force 128 x 64 / 128 x 128 -- the same bits, the tests run both)."""
    _lib.require_device(x, packed_w, bias, residual, out)
    if not x.is_contiguous():
        x = x.contiguous()
    k = int(x.shape[-1])
    m = x.numel() // k
    shape = tuple(x.shape[:-1]) + (int(n_out),)
    if k % 4:
        raise NotImplementedError(f"monai_amd.linear: {k} input features are not on the HIP path (a multiple of 4 is)")
    if residual is not None and (tuple(residual.shape) != shape or not residual.is_contiguous()):
        raise RuntimeError(f"monai_amd.linear: residual must be contiguous {shape}, got {tuple(residual.shape)}")
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != shape or not out.is_contiguous():
        raise RuntimeError(f"monai_amd.linear: out must be contiguous {shape}")
    _lib.lib().call("mh_linear_tile_f32", _lib.ptr(x), _lib.ptr(packed_w), _lib.ptr(bias), _lib.ptr(residual), _lib.ptr(out), m, int(n_out), k, 1 if gelu else 0,
                    int(tile), _s(x))
    return out


def layernorm(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], eps: float = 1e-5) -> torch.Tensor:
    """nn.LayerNorm over the last dimension (<= 4096 features)."""
    _lib.require_device(x, weight, bias)
    if not x.is_contiguous():
        x = x.contiguous()
    k = int(x.shape[-1])
    if k > 4096:
        raise NotImplementedError(f"monai_amd.layernorm: {k} features are not on the HIP path (<= 4096 are)")
    out = torch.empty_like(x)
    _lib.lib().call("mh_layernorm_f32", _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), float(eps), _lib.ptr(out), x.numel() // k, k, _s(x))
    return out


def layernorm_gather_accepts(k: int) -> bool:
    return bool(_lib.lib().query("mh_layernorm_gather_accepts", int(k)))


def layernorm_gather(x: torch.Tensor, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor], eps: float, src_row: torch.Tensor) -> torch.Tensor:
    """y[r] = LayerNorm(x.reshape(-1, K)[src_row[r]]), zeros where src_row[r] < 0: [len(src_row), K] -- SwinUNETR's norm1 -> pad -> roll -> window_partition in one pass
    (src_row int32: the voxel row every (window, token) row holds)."""
    _lib.require_device(x, weight, bias)
    _lib.require_device(src_row, dtypes=(torch.int32,))
    if not x.is_contiguous() or not src_row.is_contiguous() or src_row.dim() != 1:
        raise RuntimeError("monai_amd.layernorm_gather: contiguous x and a contiguous 1-D int32 row map are required")
    k = int(x.shape[-1])
    out = torch.empty((src_row.numel(), k), dtype=torch.float32, device=x.device)
    _lib.lib().call("mh_layernorm_gather_f32", _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), float(eps), _lib.ptr(out), src_row.numel(), k, _lib.ptr(src_row), _s(x))
    return out


def linear_scatter(x: torch.Tensor, packed_w: torch.Tensor, n_out: int, bias: Optional[torch.Tensor], residual: torch.Tensor, dst_row: torch.Tensor,
                   gelu: bool = False) -> torch.Tensor:
    """out = empty_like(residual); out[dst_row[m]] = act(x[m] . W^T + bias) + residual[dst_row[m]] for every row m of x with dst_row[m] >= 0 -- SwinUNETR's proj ->
    window_reverse -> roll back -> crop -> shortcut + x in the projection's epilogue.  Every row of `residual` must be the image of exactly one row of x (the caller's
    map is a bijection between the non-padding window rows and the voxel rows); residual [..., N] contiguous."""
    _lib.require_device(x, packed_w, bias, residual)
    _lib.require_device(dst_row, dtypes=(torch.int32,))
    if not x.is_contiguous():
        x = x.contiguous()
    k = int(x.shape[-1])
    m = x.numel() // k
    if k % 4:
        raise NotImplementedError(f"monai_amd.linear_scatter: {k} input features are not on the HIP path (a multiple of 4 is)")
    if not residual.is_contiguous() or residual.shape[-1] != n_out or not dst_row.is_contiguous() or dst_row.numel() != m:
        raise RuntimeError("monai_amd.linear_scatter: residual must be contiguous [..., N] and dst_row hold one entry per row of x")
    out = torch.empty_like(residual)
    _lib.lib().call("mh_linear_scatter_f32", _lib.ptr(x), _lib.ptr(packed_w), _lib.ptr(bias), _lib.ptr(residual), _lib.ptr(out), m, int(n_out), k, 1 if gelu else 0,
                    _lib.ptr(dst_row), _s(x))
    return out


def conv3d_k3_strided(x, x_nrm, packed_w0, bias, out, stride: int):
    """out = conv3x3x3(act(x), stride, padding 1) + bias; packed_w0 = conv3d_k3_pack(0, weight)."""
    _lib.require_device(x, x_nrm, packed_w0, bias, out)
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out)
    _lib.lib().call("mh_conv3d_k3_strided_f32", C.byref(xi), _lib.ptr(packed_w0), _lib.ptr(bias), C.byref(xo), int(stride), _s(x))
    return out


def deconv_k3(x, x_nrm, weight, bias, out, stride: int):
    """out = conv_transpose3d(act(x), k=3, stride, padding 1, output_padding stride-1) + bias."""
    _lib.require_device(x, x_nrm, weight, bias, out)
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out)
    _lib.lib().call("mh_deconv_k3_f32", C.byref(xi), _lib.ptr(weight), _lib.ptr(bias), C.byref(xo), int(stride), _s(x))
    return out


# ---- pre-processing in front of the path (ScaleIntensityRange / CropForeground) -------------------------------------
def scale_intensity_range(src: torch.Tensor, a_min: float, a_div: float, b_scale: Optional[float], b_min: float,
                          lo: Optional[float], hi: Optional[float]) -> torch.Tensor:
    """y = (x - a_min) / a_div [* b_scale + b_min when b_scale is not None] [clamped to lo / hi when given]; contiguous fp32."""
    _lib.require_device(src)
    if not src.is_contiguous():
        raise RuntimeError("monai_amd.scale_intensity_range: contiguous tensor required")
    out = torch.empty_like(src)
    if src.numel():
        _lib.lib().call("mh_scale_intensity_range_f32", _lib.ptr(src), _lib.ptr(out), int(src.numel()), float(a_min), float(a_div),
                        int(b_scale is not None), float(b_scale or 0.0), float(b_min), int(lo is not None), float(lo or 0.0),
                        int(hi is not None), float(hi or 0.0), _s(src))
    return out


def foreground_bbox(src: torch.Tensor):
    """src [C, D, H, W] fp32 -> (zmin, ymin, xmin, zmax, ymax, xmax), inclusive, of the voxels with any channel > 0, or None
    when there is no foreground.  One device -> host read of 24 bytes (the caller needs the box to size its output)."""
    _lib.require_device(src)
    if src.dim() != 4 or not src.is_contiguous() or not src.numel():
        raise RuntimeError("monai_amd.foreground_bbox: contiguous non-empty [C, D, H, W] tensor required")
    c, d, h, w = (int(v) for v in src.shape)
    L = _lib.lib()
    ws = torch.empty(L.query("mh_foreground_bbox_workspace_ints", d, h) + 6, dtype=torch.int32, device=src.device)
    box = ws[-6:]
    L.call("mh_foreground_bbox_f32", _lib.ptr(src), c, d, h, w, _lib.ptr(ws), _lib.ptr(box), _s(src))
    b = [int(v) for v in box.cpu().tolist()]
    return None if b[3] < 0 else tuple(b)


def crop_pad(src: torch.Tensor, start: Sequence[int], size: Sequence[int], value: float = 0.0) -> torch.Tensor:
    """src [C, D, H, W] -> [C, *size]: out[c, z, y, x] = src[c, z + start_z, ...] inside the source, `value` outside."""
    _lib.require_device(src)
    if src.dim() != 4 or not src.is_contiguous() or not src.numel():
        raise RuntimeError("monai_amd.crop_pad: contiguous non-empty [C, D, H, W] tensor required")
    c, d, h, w = (int(v) for v in src.shape)
    do, ho, wo = (int(v) for v in size)
    out = torch.empty((c, do, ho, wo), dtype=torch.float32, device=src.device)
    if out.numel():
        _lib.lib().call("mh_crop_pad_f32", _lib.ptr(src), _lib.ptr(out), c, d, h, w, do, ho, wo, int(start[0]), int(start[1]), int(start[2]),
                        float(value), _s(src))
    return out


def groupnorm_finalize(stats, tiles: int, n: int, c: int, groups: int, gamma, beta, eps: float, slope: float, nrm: torch.Tensor):
    """GroupNorm flavour of `instnorm_finalize`: the records of the C / groups channels of a group are merged; every channel
    gets {alpha = gamma_c / sqrt(var_g + eps), beta = beta_c - mean_g * alpha, slope, 0}."""
    _lib.require_device(stats, gamma, beta, nrm)
    if nrm.dim() != 3 or nrm.shape[2] != 4 or nrm.stride(2) != 1 or nrm.stride(1) != 4:
        raise RuntimeError("monai_amd.groupnorm_finalize: nrm must be a [N,C,4] slice")
    ns = nrm.stride(0) if n > 1 else max(nrm.stride(0), 4 * c)
    _lib.lib().call("mh_groupnorm_finalize_f32", _lib.ptr(stats), int(tiles), int(n), int(c), int(groups), _lib.ptr(gamma), _lib.ptr(beta),
                    float(eps), float(slope), _lib.ptr(nrm), int(ns), _s(stats))
    return nrm


def flip_permute(src: torch.Tensor, perm: Sequence[int], flip: Sequence[bool]) -> torch.Tensor:
    """src [C, D, H, W] -> [C, size[perm[0]], size[perm[1]], size[perm[2]]]: output axis k is input axis perm[k]; input axis a is
    reversed when flip[a] (== torch.flip over those axes followed by permute)."""
    _lib.require_device(src)
    if src.dim() != 4 or not src.is_contiguous() or not src.numel():
        raise RuntimeError("monai_amd.flip_permute: contiguous non-empty [C, D, H, W] tensor required")
    size = [int(v) for v in src.shape[1:]]
    out = torch.empty((int(src.shape[0]),) + tuple(size[int(p)] for p in perm), dtype=torch.float32, device=src.device)
    _lib.lib().call("mh_flip_permute_f32", _lib.ptr(src), _lib.ptr(out), int(src.shape[0]), _lib.int_array(size), _lib.int_array(perm),
                    _lib.int_array([1 if f else 0 for f in flip]), _s(src))
    return out


def normalize_stats(src: torch.Tensor, channels: int, n: int, nonzero: bool) -> torch.Tensor:
    """src = `channels` runs of n contiguous fp32 values -> DEVICE table [channels, 2] of {mean, std (population; 0 -> 1)} over all
    (or only the non-zero) values of each run.  No host synchronisation."""
    _lib.require_device(src)
    if not src.is_contiguous() or src.numel() != channels * n or n < 1:
        raise RuntimeError("monai_amd.normalize_stats: contiguous tensor of channels * n elements required")
    L = _lib.lib()
    ws = torch.empty(L.query("mh_normalize_stats_workspace_doubles", int(channels), int(n)), dtype=torch.float64, device=src.device)
    table = torch.empty((int(channels), 2), dtype=torch.float32, device=src.device)
    L.call("mh_normalize_stats_f32", _lib.ptr(src), int(channels), int(n), int(bool(nonzero)), _lib.ptr(ws), _lib.ptr(table), _s(src))
    return table


def normalize_apply(src: torch.Tensor, channels: int, n: int, nonzero: bool, table: torch.Tensor) -> torch.Tensor:
    """y = (x - table[c, 0]) / table[c, 1] per run c (non-zero values only when `nonzero`; zeros pass through)."""
    _lib.require_device(src, table)
    if not src.is_contiguous() or src.numel() != channels * n or tuple(table.shape) != (channels, 2) or not table.is_contiguous():
        raise RuntimeError("monai_amd.normalize_apply: contiguous tensor of channels * n elements and a [channels, 2] table required")
    out = torch.empty_like(src)
    _lib.lib().call("mh_normalize_apply_f32", _lib.ptr(src), _lib.ptr(out), int(channels), int(n), int(bool(nonzero)), _lib.ptr(table), _s(src))
    return out


def conv3d_k3_strided3(x, x_nrm, packed_w0, bias, out, strides: Sequence[int]):
    """out = conv3x3x3(act(x), strides (sz, sy, sx), padding 1) + bias; packed_w0 = conv3d_k3_pack(0, weight)."""
    _lib.require_device(x, x_nrm, packed_w0, bias, out)
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out)
    _lib.lib().call("mh_conv3d_k3_strided3_f32", C.byref(xi), _lib.ptr(packed_w0), _lib.ptr(bias), C.byref(xo), int(strides[0]), int(strides[1]),
                    int(strides[2]), _s(x))
    return out


def conv3d_k3s2_accepts(cin: int, cout: int, d: int, h: int, w: int) -> bool:
    """can the split-precision stride-2 kernel (kernels/conv3d_s2_h2.h) serve `cin` -> `cout` channels on a d x h x w input?"""
    return bool(_lib.lib().query("mh_conv3d_k3s2_accepts", int(cin), int(cout), int(d), int(h), int(w)))


def conv3d_k3s2_selected(cin: int, cout: int, d: int, h: int, w: int, stride, bounded: bool) -> bool:
    """should this stride-`stride` 3x3x3 convolution run on the split-precision stride-2 kernel? (stride (2, 2, 2), a bounded input, a shape the kernel takes, and the
    host-side family switch `monai_amd.config.strided_h2()`)"""
    from . import config

    st = (int(stride),) * 3 if isinstance(stride, int) else tuple(int(v) for v in stride)
    # output planes below 8 x 8 fill a quarter of a workgroup's 256-voxel tile at best: measured slower than the direct kernel there (256 -> 320 @ 12^3 -> 6^3 x 64
    # windows: 2.18 vs 1.61 ms, profiles/r05_s2_layers.txt), 3.0-3.9 x faster from 12 x 12 outputs on
    return st == (2, 2, 2) and bool(bounded) and config.strided_h2() and h >= 16 and w >= 16 and conv3d_k3s2_accepts(cin, cout, d, h, w)


def conv3d_k3s2_workspace_floats(n: int, cin: int, d: int, h: int, w: int) -> int:
    return _lib.lib().query("mh_conv3d_k3s2_workspace_floats", int(n), int(cin), int(d), int(h), int(w))


def conv3d_k3s2_pack(weight: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3, 3] -> the stride-2 kernel's phase-ordered split-precision tap matrices (once per parameter version)"""
    _lib.require_device(weight)
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    packed = torch.zeros(_lib.lib().query("mh_conv3d_k3s2_packed_floats", cin, cout), dtype=torch.float32, device=weight.device)
    _lib.lib().call("mh_conv3d_k3s2_pack_f32", _lib.ptr(weight.contiguous()), cin, cout, _lib.ptr(packed), _s(weight))
    return packed


def conv3d_k3s2_stat_tiles(d: int, h: int, w: int) -> int:
    return _lib.lib().query("mh_conv3d_k3s2_stat_tiles", int(d), int(h), int(w))


def conv3d_k3s2_fused(cin: int, cout: int, voxels: int = 0) -> bool:
    """should the stride-2 kernel convert its input inside the GEMM's staging (no split pass, no workspace)?  `monai_amd.config.strided_h2_fused()`: "auto" = layers
    with ONE group of 64 output channels on large volumes (`voxels` = D * H * W of the input >= 80^3) -- measured 5.3 vs 6.1 ms (32 -> 64 @ 96^3 x 64 windows) and 2.25 vs
    2.61 ms (16 -> 32), but 3.4 vs 2.8 ms where the conversion is repeated for a second group and slower on every smaller plane (profiles/r05_s2_layers_fused.txt);
    at most 512 input channels (their records sit in LDS)"""
    from . import config

    mode = config.strided_h2_fused()
    if cin > 512 or mode == "0":
        return False
    return mode == "1" or (cout <= 64 and voxels >= 80 ** 3)


def conv3d_k3s2(x, x_nrm, packed, bias, out, stats=None, workspace=None, fused: Optional[bool] = None):
    """out = conv3x3x3(act(x), stride 2, padding 1) + bias on the fp16 matrix cores (split precision); x_nrm: records WITH magnitude bounds.
    stats ([N * Cout * conv3d_k3s2_stat_tiles(*x.shape[2:]) * 3] floats): the InstanceNorm statistics of `out`.  fused (default `conv3d_k3s2_fused`): convert inside the
    GEMM's staging; otherwise a phase-split pass into `workspace` (scratch of `conv3d_k3s2_workspace_floats` floats, allocated here when not given) comes first."""
    _lib.require_device(x, x_nrm, packed, bias, out, stats, workspace)
    if fused is None:
        fused = conv3d_k3s2_fused(int(x.shape[1]), int(out.shape[1]), int(x.shape[2] * x.shape[3] * x.shape[4]))
    if not fused:
        need = conv3d_k3s2_workspace_floats(*x.shape)
        if workspace is None or workspace.numel() < need:
            workspace = torch.empty(need, dtype=torch.float32, device=x.device)
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out)
    _lib.lib().call("mh_conv3d_k3s2_f32", C.byref(xi), _lib.ptr(packed), _lib.ptr(bias), C.byref(xo), None if fused else _lib.ptr(workspace), _lib.ptr(stats), int(bool(fused)), _s(x))
    return out


def deconv_ks(x, x_nrm, weight, bias, out, factors: Sequence[int], out_nrm=None):
    """out = conv_transpose3d(act(x), kernel == stride == factors (each 1 or 2)) + bias; weight [Cin, Cout, fz, fy, fx] contiguous.
    out_nrm: `nrm_identity` records of `out` (magnitude bound)."""
    _lib.require_device(x, x_nrm, weight, bias, out, out_nrm)
    if not weight.is_contiguous() or tuple(weight.shape[2:]) != tuple(int(f) for f in factors):
        raise RuntimeError("monai_amd.deconv_ks: contiguous weight [Cin, Cout, fz, fy, fx] required")
    xi, xo = _lib.tensor5(x, x_nrm), _lib.tensor5(out, out_nrm)
    _lib.lib().call("mh_deconv_ks_f32", C.byref(xi), _lib.ptr(weight), _lib.ptr(bias), C.byref(xo), int(factors[0]), int(factors[1]), int(factors[2]), _s(x))
    return out


def minmax_scale(src: torch.Tensor, channels: int, n: int, b_scale: Optional[float], b_min: float, flat_mul: Optional[float]) -> torch.Tensor:
    """rescale_array per run of n values: (x - min) / (max - min) [* b_scale + b_min when b_scale is not None]; a constant run becomes
    x * flat_mul (x when flat_mul is None).  Min / max stay on the device between the two passes."""
    _lib.require_device(src)
    if not src.is_contiguous() or src.numel() != channels * n or n < 1:
        raise RuntimeError("monai_amd.minmax_scale: contiguous tensor of channels * n elements required")
    L = _lib.lib()
    ws = torch.empty(L.query("mh_minmax_workspace_floats", int(channels), int(n)) + 2 * int(channels), dtype=torch.float32, device=src.device)
    table = ws[-2 * int(channels):]
    out = torch.empty_like(src)
    L.call("mh_minmax_f32", _lib.ptr(src), int(channels), int(n), _lib.ptr(ws), _lib.ptr(table), _s(src))
    L.call("mh_minmax_scale_f32", _lib.ptr(src), _lib.ptr(out), int(channels), int(n), _lib.ptr(table), int(b_scale is not None), float(b_scale or 0.0),
           float(b_min), int(flat_mul is not None), float(flat_mul or 0.0), _s(src))
    return out
