"""``sliding_window_inference`` on the MI355X kernels -- drop-in for monai/inferers/utils.py:42-321.

Same signature and results as the reference (bit-identical blend for the same window logits), different
schedule.  The reference runs a Python loop that, per window batch, multiplies the predictor output by the
importance map and read-modify-writes it into the output volume and a count map (5x the algorithmic HBM
traffic).  Here:

  1. windows are gathered by ``mh_window_extract_f32`` straight from the volume into a batch buffer;
  2. the predictor writes its logits into ONE buffer holding every window of the image
     (``[num_win, K, roi]`` -- 17.7 GB for 512^3 / 96^3 / overlap 0.5 / 5 classes; HBM is 288 GB);
  3. ``mh_sw_blend_f32`` produces every output voxel in a single gather pass that walks the covering windows
     in ascending window index: ``acc += logit*w; cnt += w; out = acc/cnt`` -- exactly the floating-point
     operations, in exactly the order, of the reference's ``*=``, ``+=``, ``/=``; the count map is never stored.

With window sharding enabled (monai_amd.parallel) each rank runs the predictor on a contiguous range of
windows and one RCCL all-gather of the logits precedes the (replicated, deterministic) blend.
"""

from __future__ import annotations

import os
import threading
import warnings
from collections.abc import Callable, Mapping, Sequence
from typing import Any

import torch
import torch.nn.functional as F

from .. import _lib, _prof, ops, parallel
from .._fallback import function_fallback
from ..data.utils import compute_importance_map, get_valid_patch_size, importance_map_factors, window_starts
from ..utils.misc import ensure_tuple, ensure_tuple_rep, fall_back_tuple, look_up_option

__all__ = ["sliding_window_inference", "sliding_window_argmax"]

_PAD_MODES = ("constant", "reflect", "replicate", "circular")
_NEAREST = "nearest-exact"


def _get_scan_interval(image_size, roi_size, num_spatial_dims: int, overlap) -> tuple:
    """monai/inferers/utils.py:363-384: ``int(roi * (1 - overlap))``, at least 1; the roi itself when it spans the image."""
    if len(image_size) != num_spatial_dims:
        raise ValueError(f"len(image_size) {len(image_size)} different from spatial dims {num_spatial_dims}.")
    if len(roi_size) != num_spatial_dims:
        raise ValueError(f"len(roi_size) {len(roi_size)} different from spatial dims {num_spatial_dims}.")
    steps = []
    for size, roi, o in zip(image_size, roi_size, overlap):
        if roi == size:
            steps.append(int(roi))
        else:
            step = int(roi * (1 - o))
            steps.append(step if step > 0 else 1)
    return tuple(steps)


def _flatten_struct(seg_out):
    """tensor | tuple | dict -> (sorted keys or None, tuple of tensors)  (utils.py:387-398)"""
    if isinstance(seg_out, torch.Tensor):
        return None, (seg_out,)
    if isinstance(seg_out, Mapping):
        keys = sorted(seg_out.keys())
        return keys, tuple(seg_out[k] for k in keys)
    return None, ensure_tuple(seg_out)


def _pack_struct(seg_out, dict_keys=None):
    if dict_keys is not None:
        return dict(zip(dict_keys, seg_out))
    if isinstance(seg_out, (list, tuple)) and len(seg_out) == 1:
        return seg_out[0]
    return ensure_tuple(seg_out)


def _to3(values, fill):
    """left-pad a 1/2/3-element spatial tuple to 3 entries"""
    values = tuple(values)
    return (fill,) * (3 - len(values)) + values


def _auto_batch(predictor, roi3, num_win: int, sw_batch_size: int, device, world: int = 1, sharded: bool = False) -> int:
    """Windows per predictor call.  A generic predictor gets exactly the user's ``sw_batch_size``.  The fused engines
    size the batch for 288 GB of HBM instead (results do not depend on the batch: InstanceNorm is per sample): more
    windows per launch fill the 256 CUs at the deep U-Net levels, and a power-of-two count keeps the workgroup grids of
    the large layers whole multiples of the CU count -- measured on the BASELINE workload (profiles/): 25 windows per
    launch 1.85 s, 32: 1.75 s, 64: 1.73 s, 128: 1.72 s.  Default 64 on one GPU.  When the windows are sharded over `world`
    GPUs the rounds follow parallel.WindowShard.schedule (main rounds of world x nb windows, shorter tail rounds): nb is the value in [32, 64] that
    gives the busiest rank the fewest windows, ties to the larger.
    `num_win` is the TOTAL number of windows.  Override with MONAI_AMD_SW_BATCH; MONAI_AMD_STRICT_SW_BATCH=1 keeps the user's value."""
    if not hasattr(predictor, "forward_into") or os.environ.get("MONAI_AMD_STRICT_SW_BATCH") == "1":
        return max(1, int(sw_batch_size))
    per_rank = max(-(-num_win // world), 1)
    env = os.environ.get("MONAI_AMD_SW_BATCH")
    if env:
        return max(1, min(int(env), per_rank))
    cap = 64
    if device.type == "cuda":
        free, _ = torch.cuda.mem_get_info(device)
        per_win = 6.0 * 4 * max(getattr(predictor, "features", (32,))[0], 1) * roi3[0] * roi3[1] * roi3[2]
        fit = int(max(1, (0.35 * free) // max(per_win, 1)))
        while cap > 1 and cap > fit:
            cap //= 2
    cap = max(cap, int(sw_batch_size)) if cap >= sw_batch_size else cap
    cap = max(1, min(cap, per_rank))
    if (world > 1 or sharded) and cap > 1:
        # the busiest rank (rank 0: a full slot in every round of parallel.WindowShard.schedule) sets the step time: the nb in [cap / 2, cap] that gives it the fewest
        # windows, a launch counted as at least 7 windows (below that the large layers no longer fill 256 CUs), ties to the larger nb
        probe = parallel.partition(num_win, world, 0, force=sharded)
        best, best_cost = cap, None
        for nb in range(cap, max(cap // 2, 1) - 1, -1):
            cost = sum(max(n, min(7, nb)) for _, n in probe.schedule(nb))
            if best_cost is None or cost < best_cost:
                best, best_cost = nb, cost
        cap = best
    return cap


@function_fallback("monai.inferers.utils", "sliding_window_inference")
def sliding_window_inference(
    inputs: torch.Tensor,
    roi_size: Sequence[int] | int,
    sw_batch_size: int,
    predictor: Callable[..., Any],
    overlap: Sequence[float] | float = 0.25,
    mode: str = "constant",
    sigma_scale: Sequence[float] | float = 0.125,
    padding_mode: str = "constant",
    cval: float = 0.0,
    sw_device=None,
    device=None,
    progress: bool = False,
    roi_weight_map: torch.Tensor | None = None,
    process_fn: Callable | None = None,
    buffer_steps: int | None = None,
    buffer_dim: int = -1,
    with_coord: bool = False,
    *args: Any,
    **kwargs: Any,
):
    """See the reference docstring (monai/inferers/utils.py:63-141) for the argument semantics; they are kept.

    ``buffer_steps`` / ``buffer_dim`` (utils.py:239-253, 276-284, 324-348): the reference's buffered schedule sums in ANOTHER ORDER than its plain path
    (windows sorted by their start along ``buffer_dim``, slabs of ``buffer_steps`` distinct starts accumulated from zero and then added to the output), so
    its result differs from the plain one by roundings (tests/golden/buffered.npz: up to 26 000 voxels of a small volume, <= 6e-7).  The schedule itself --
    a memory-saving device -- is not reproduced (the logits of all windows sit in HBM), its ARITHMETIC is: with ``buffer_steps`` the blend runs in the
    buffered order (``mh_sw_blend_buffered_f32``) and returns the bits of the reference's buffered run.  With ``process_fn`` / ``with_coord`` / tuple or dict
    outputs the predictor is also CALLED as the reference calls it there (windows in the sorted order, batches that end at the slab boundaries, the sorted
    slices as coordinates; only the FIRST output is blended, as utils.py:244 does; the count map is the weight map of the batch at the first flush):
    ``_buffered_batches``.  Multi-resolution outputs with ``buffer_steps`` fail in the reference too and are refused.
    Other differences, all result-neutral: when the logits of all windows do not fit in HBM the volume is processed slab by slab along its first spatial
    axis with bit-identical results, see ``_slabwise``; ``sw_device`` must be a ROCm device (a CPU volume is moved to it once; half / bfloat16 volumes are widened to fp32 and the result
    returned in the caller's dtype).  A predictor with ``forward_into`` (the conv engines) is given up to 64 windows per launch
    instead of ``sw_batch_size`` -- result-neutral, but it changes peak memory; MONAI_AMD_STRICT_SW_BATCH=1 keeps the caller's value
    and MONAI_AMD_SW_BATCH=n sets it.  ``process_fn`` (utils.py:232-234) is honoured with
    the reference's semantics: each batch is multiplied by the weight map it returns, the count map uses the first batch's.
    """
    num_spatial_dims = inputs.dim() - 2
    buffered = buffer_steps is not None and buffer_steps > 0
    if buffered:
        if buffer_dim < -num_spatial_dims or buffer_dim > num_spatial_dims:
            raise ValueError(f"buffer_dim must be in [{-num_spatial_dims}, {num_spatial_dims}], got {buffer_dim}.")
        if buffer_dim < 0:
            buffer_dim += num_spatial_dims
        if buffer_dim >= num_spatial_dims:
            raise NotImplementedError("monai_amd: buffer_steps with buffer_dim == the number of spatial dims is not on the HIP path")
        if kwargs.get("_monai_amd_argmax") is not None:       # rejected BEFORE any window is predicted (a fall-through would otherwise predict them all twice)
            raise NotImplementedError("monai_amd: the fused argmax epilogue with buffer_steps is not on the HIP path (blend in the buffered order, then AsDiscrete)")
        if not kwargs.pop("_monai_amd_buffered_inner", False):
            # `buffer_steps` is the reference's MEMORY-SAVING option: the callers who pass it are the ones with large volumes.  The buffered summation order
            # needs the logits of all windows resident; when they do not fit, the volume goes through the plain order slab by slab (bit-identical to the
            # reference's plain run, which differs from its buffered run by roundings <= 1e-6 -- tests/golden/buffered.npz) instead of failing.
            common = (roi_size, sw_batch_size, predictor, overlap, mode, sigma_scale, padding_mode, cval, sw_device, device, progress, roi_weight_map, process_fn)
            try:
                return sliding_window_inference(inputs, *common, buffer_steps, buffer_dim, with_coord, *args, _monai_amd_buffered_inner=True, **kwargs)
            except _LogitsDoNotFit as e:
                if process_fn is not None or with_coord:      # no slab-wise form exists for these (their semantics are tied to the whole volume): the plain retry could only
                    raise                                     # repeat the predictor calls already made and fail the same way
                reason = str(e).split(" (")[0]
            warnings.warn(f"{reason}: buffer_steps={buffer_steps} is served in the plain summation order, slab by slab "
                          "(equal to the reference's unbuffered result; its buffered result differs from that by roundings)")
            return sliding_window_inference(inputs, *common, None, -1, with_coord, *args, **kwargs)
    overlap = ensure_tuple_rep(overlap, num_spatial_dims)
    for o in overlap:
        if o < 0 or o >= 1:
            raise ValueError(f"overlap must be >= 0 and < 1, got {overlap}.")
    if num_spatial_dims < 1 or num_spatial_dims > 3:
        raise NotImplementedError(f"monai_amd: sliding windows over {num_spatial_dims} spatial dims are not supported (1-3 are)")
    meta_src = inputs if (type(inputs) is not torch.Tensor and hasattr(inputs, "as_tensor")) else None
    if meta_src is not None:
        inputs = inputs.as_tensor()
    # drop-in input handling (utils.py:146-153 take any float dtype and any inputs/sw_device pair): half / bfloat16 volumes are
    # widened to fp32 for the kernels and the result is returned in the caller's dtype; a CPU volume with a ROCm `sw_device` is
    # moved to HBM once (the reference moves it window by window) and the result goes back to `device` or the inputs' device
    if torch.is_grad_enabled() and inputs.requires_grad:
        # the reference's result carries the autograd graph of the windows (tests/inferers/test_sliding_window_inference.py:124-139);
        # the kernels do not record one: such a call belongs to the reference path
        raise NotImplementedError("monai_amd: sliding_window_inference of an input that requires grad (autograd through the blend) is not on the HIP path")
    narrow = inputs.dtype if inputs.dtype in (torch.float16, torch.bfloat16) else None
    host_in = (not inputs.is_cuda) and sw_device is not None and torch.device(sw_device).type == "cuda"
    if narrow is not None or host_in:
        x = inputs.to(device=torch.device(sw_device) if host_in else inputs.device, dtype=torch.float32)
        out = sliding_window_inference(x, roi_size, sw_batch_size, predictor, overlap, mode, sigma_scale, padding_mode, cval, sw_device,
                                       device if device is not None else inputs.device, progress, roi_weight_map, process_fn, buffer_steps,
                                       buffer_dim, with_coord, *args, **kwargs)
        keys, parts = _flatten_struct(out)
        if narrow is not None:
            parts = [t.to(narrow) if t.is_floating_point() else t for t in parts]
        if meta_src is not None:
            parts = [_restore_meta(t, meta_src) for t in parts]
        return _pack_struct(parts, keys)
    _lib.require_device(inputs)
    compute_dtype = inputs.dtype
    batch_size, in_ch, *image_size_ = inputs.shape
    out_device = torch.device(device) if device is not None else inputs.device
    if sw_device is not None and torch.device(sw_device).type != inputs.device.type:
        raise RuntimeError("monai_amd: sw_device must be the ROCm device of the inputs (windows are gathered in HBM)")
    roi_size = fall_back_tuple(roi_size, image_size_)

    # pad when the image is smaller than the roi (utils.py:163-170), centred, last dim first
    image_size = tuple(max(image_size_[i], roi_size[i]) for i in range(num_spatial_dims))
    pad_size = []
    for k in range(inputs.dim() - 1, 1, -1):
        diff = max(roi_size[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad_size.extend([half, diff - half])
    if any(pad_size):
        inputs = F.pad(inputs, pad=pad_size, mode=look_up_option(padding_mode, _PAD_MODES, "padding_mode"), value=cval)
    inputs = inputs.contiguous()

    scan_interval = _get_scan_interval(image_size, roi_size, num_spatial_dims, overlap)
    starts = window_starts(image_size, roi_size, scan_interval)
    num_win = 1
    for s in starts:
        num_win *= len(s)

    # all-window logits that do not fit in HBM: slab by slab along the first spatial axis (see _slabwise)
    argmax_dtype = kwargs.pop("_monai_amd_argmax", None)      # fused AsDiscrete(argmax=True) epilogue (sliding_window_argmax below)
    slab_ok = (not kwargs.pop("_monai_amd_no_slabs", False) and not with_coord and process_fn is None and not any(pad_size)
               and len(starts[0]) > 1 and not buffered)
    if slab_ok:
        sub_kwargs = dict(kwargs, _monai_amd_no_slabs=True, _monai_amd_argmax=argmax_dtype)

        def _whole(x):
            return sliding_window_inference(x, roi_size, sw_batch_size, predictor, overlap, mode, sigma_scale, padding_mode, cval, sw_device,
                                            device, progress, roi_weight_map, None, buffer_steps, buffer_dim, False, *args, **sub_kwargs)

        try:
            return _whole(meta_src if meta_src is not None else inputs)
        except _LogitsDoNotFit as e:      # leave the handler before retrying: the traceback keeps the failed call's buffers alive
            need, budget = e.need, (0.9 * e.budget if _logits_budget(inputs.device) is None else e.budget)
        out = _slabwise(inputs, roi_size, starts[0], need, budget, _whole)
        if meta_src is not None:
            keys, parts = _flatten_struct(out)
            out = _pack_struct([_restore_meta(t, meta_src) for t in parts], keys)
        return out

    # importance map, always evaluated on the host in fp32 (bit-identical to the reference's CPU map)
    valid_patch_size = get_valid_patch_size(image_size, roi_size)
    imp_key = None          # cache key of a map this module evaluated itself (a caller's roi_weight_map is uploaded per call)
    if valid_patch_size == tuple(roi_size) and roi_weight_map is not None:
        imp = roi_weight_map
    else:
        try:
            imp, imp_key = _host_importance_map(valid_patch_size, mode, sigma_scale, compute_dtype)
        except Exception as e:  # same wrapping as the reference (utils.py:205-209)
            raise RuntimeError(
                f"patch size {valid_patch_size}, mode={mode}, sigma_scale={sigma_scale}, device={device}\n"
                "Seems to be OOM. Please try smaller patch size or mode='constant' instead of mode='gaussian'."
            ) from e
    imp = imp.to(dtype=compute_dtype)
    while imp.dim() > num_spatial_dims:
        imp = imp[0]

    roi3 = _to3(roi_size, 1)
    img3 = _to3(image_size, 1)
    grid3 = [[0]] * (3 - num_spatial_dims) + [list(s) for s in starts]
    dev = inputs.device

    # windows owned by this rank: all of them, or (window sharding on) its slot of every round -- monai_amd/parallel.py
    shard = parallel.window_shard(num_win)
    nb = _auto_batch(predictor, roi3, num_win, sw_batch_size, dev, world=shard.world, sharded=shard.sharded)
    nb = shard.agree_batch(nb, dev)
    my_rounds = shard.rounds(nb)
    fused = hasattr(predictor, "forward_into") and not with_coord and not args and not kwargs and process_fn is None
    win_buf = torch.empty((nb, in_ch) + roi3, dtype=compute_dtype, device=dev)

    windows_nd = None
    if with_coord:
        import itertools

        windows_nd = [tuple(slice(s, s + roi_size[d]) for d, s in enumerate(w)) for w in itertools.product(*starts)]

    mosaic = None      # fused single-GPU path: the logits in the mosaic layout (ops.LogitsMosaic) instead of window-major rows
    logits = None      # per output: [num_win_padded, K, *seg3]
    seg_shapes = None  # per output: spatial shape of one window's prediction (native dims)
    dict_keys = None
    outputs = None     # per output: [B, K, *out_img3]
    weights = None     # per output: importance map resampled to the prediction size (device)
    proc_weights = None  # process_fn: per output, the first batch's weight map (the reference's count map uses only that one)
    imp_dev = None
    zscales = None

    fused = fused and hasattr(predictor, "out_channels") and getattr(predictor, "window_sized_output", True)
    # buffer_steps with a callback in the loop: the predictor sees the reference's buffered batch schedule (sorted windows, batches cut at the slab ends)
    buffered_calls = buffered and not fused and (process_fn is not None or with_coord or not shard.sharded)
    if buffered_calls and shard.sharded:
        raise NotImplementedError("monai_amd: buffer_steps with process_fn / with_coord under window sharding is not on the HIP path")
    for b in range(batch_size):
        vol3 = inputs[b].reshape((in_ch,) + img3)
        steps = list(enumerate(my_rounds))
        if progress:
            try:
                from tqdm import tqdm

                steps = tqdm(steps)
            except ImportError:
                pass
        pending = []
        if buffered_calls:
            steps = []            # the windows of this image are predicted here, in the reference's buffered order; the blend below reads their rows
            keys_b, logits, count_map = _buffered_batches(b, vol3, in_ch, grid3, starts, roi_size, roi3, num_win, int(sw_batch_size), predictor, process_fn, with_coord,
                                                          imp, buffer_dim, int(buffer_steps), shard, nb, compute_dtype, dev, args, kwargs, logits)
            if b == 0:
                dict_keys = keys_b
                seg_shapes, zscales = [tuple(roi_size)], [None]
                if count_map is not None:
                    proc_weights = [count_map]
        for q, (w0, n) in steps:
            if fused and logits is None:       # sized from the engine's out_channels, not from a first prediction: a rank whose slot of the first round is empty allocates too
                k = int(predictor.out_channels)
                seg_shapes, zscales = [tuple(roi_size)], [None]
                mosaic = None if buffered else _alloc_mosaic(predictor, shard, argmax_dtype, k, grid3, roi3, compute_dtype, dev)
                logits = [mosaic if mosaic is not None else _alloc_logits(shard, nb, k, roi3, compute_dtype, dev)]
            if n > 0 and fused:
                ops.window_extract(vol3, grid3, w0, n, roi3, win_buf[:n])
                with _prof.span("sw_predictor"):
                    if mosaic is not None:     # the network's last kernel writes the windows straight into the mosaic layout
                        predictor.forward_into_windows(win_buf[:n], mosaic, w0)
                    else:
                        predictor.forward_into(win_buf[:n], logits[0][w0 : w0 + n])
            elif n > 0:
                ops.window_extract(vol3, grid3, w0, n, roi3, win_buf[:n])
                win_data = win_buf[:n].reshape((n, in_ch) + tuple(roi_size))
                if with_coord:
                    coords = [[slice(b, b + 1), slice(None)] + list(windows_nd[i]) for i in range(w0, w0 + n)]
                    seg_out = predictor(win_data, coords, *args, **kwargs)
                else:
                    seg_out = predictor(win_data, *args, **kwargs)
                dict_keys, segs = _flatten_struct(seg_out)
                w_batch = None
                if process_fn is not None:     # utils.py:232-238: the callback may edit the predictions and the weight map
                    if imp_dev is None:
                        imp_dev = imp.to(dev)
                    segs, w_t = process_fn(segs, win_data, imp_dev)
                    segs = tuple(segs) if isinstance(segs, (list, tuple)) else (segs,)
                    if w_t.dim() == num_spatial_dims:
                        w_t = w_t[None, None]
                    w_batch = w_t.to(dtype=compute_dtype, device=dev)
                if buffered and logits is None and (len(segs) != 1 or tuple(segs[0].shape[2:]) != tuple(roi_size)):
                    # known after the FIRST batch, not after all of them (the reference ignores further outputs there and fails on a resized one)
                    raise NotImplementedError("monai_amd: buffer_steps with several / multi-resolution predictor outputs is not on the HIP path")
                if logits is None:
                    seg_shapes = [tuple(s.shape[2:]) for s in segs]
                    zscales = [
                        None if sh == tuple(roi_size) else [o / float(i) for o, i in zip(sh, roi_size)] for sh in seg_shapes
                    ]
                    logits = [_alloc_logits(shard, nb, int(s.shape[1]), _to3(sh, 1), compute_dtype, dev) for s, sh in zip(segs, seg_shapes)]
                for ss, s in enumerate(segs):
                    if s.dtype != compute_dtype and s.is_floating_point():
                        # a predictor under torch.autocast returns half precision; the reference weights it in that precision and
                        # accumulates in compute_dtype (utils.py:286-288) -- here it is widened first (never less precise)
                        s = s.to(compute_dtype)
                    _lib.require_device(s)
                    dst = logits[ss][w0 : w0 + n]
                    if w_batch is None:
                        dst.copy_(s.reshape(dst.shape))
                        continue
                    if zscales[ss] is not None:      # cumulative nearest resampling, as utils.py:260-263
                        w_batch = F.interpolate(w_batch, seg_shapes[ss], mode=_NEAREST)
                    if w_batch.shape[0] != 1 or w_batch.shape[1] != 1:
                        raise RuntimeError("monai_amd: process_fn must return a weight map broadcastable over batch and channels "
                                           f"(got {tuple(w_batch.shape)}; the reference's count map needs [1, 1, *spatial])")
                    if proc_weights is None:
                        proc_weights = []
                    if len(proc_weights) <= ss:      # the count map is built from the FIRST batch's map (utils.py:270-275)
                        proc_weights.append(w_batch[0, 0].reshape(_to3(seg_shapes[ss], 1)).contiguous().clone())
                    torch.mul(s.reshape(dst.shape), w_batch.reshape((1, 1) + tuple(dst.shape[2:])), out=dst)   # `seg *= w_t`
            if shard.sharded:
                if logits is None:
                    raise RuntimeError("monai_amd: a rank without windows in the first round cannot size the logits buffer "
                                       "(fewer windows than ranks x windows per launch: lower sw_batch_size)")
                # this round's rows travel while the next round computes
                pending += [shard.gather_round(lg, q, nb) for lg in logits]

        if logits is None:
            raise RuntimeError("monai_amd: no windows were processed")
        with _prof.span("sw_gather_wait"):         # what the compute stream still has to wait for after its last round
            for work in pending:
                work.wait()
        gathered = logits

        if weights is None:  # importance map per output resolution (the reference resamples cumulatively, utils.py:260-263)
            weights, w_t = [], imp[None, None]
            for sh, z in zip(seg_shapes, zscales):
                if z is not None:
                    w_t = F.interpolate(w_t, sh, mode=_NEAREST)
                weights.append(_on_device(w_t[0, 0].reshape(_to3(sh, 1)).contiguous(), dev, cache_key=None if imp_key is None else imp_key + (tuple(sh),)))
            outputs = []
            for lg, z in zip(gathered, zscales):
                osz = [int(i * zz) for i, zz in zip(image_size, z)] if z else list(image_size)
                if argmax_dtype is not None:       # only the label map is ever written: [B, 1, ...]
                    outputs.append(torch.empty((batch_size, 1) + _to3(osz, 1), dtype=argmax_dtype, device=dev))
                else:
                    outputs.append(torch.empty((batch_size, lg.k if lg is mosaic else lg.shape[1]) + _to3(osz, 1), dtype=compute_dtype, device=dev))
        for ss, (lg, z) in enumerate(zip(gathered, zscales)):
            if z is None:
                g = grid3
            else:
                g = [[0]] * (3 - num_spatial_dims) + [[int(s * zz) for s in ax] for ax, zz in zip(starts, z)]
            nlog = num_win * lg.k * roi3[0] * roi3[1] * roi3[2] if lg is mosaic else lg[:num_win].numel()
            nbytes = 4.0 * nlog + outputs[ss][b].numel() * outputs[ss][b].element_size()  # logits read once + output written once
            with _prof.span("sw_blend", nbytes):
                if buffered:      # the reference's buffered summation order (single output at window resolution; checked after the first predictor batch)
                    ops.sw_blend_buffered(lg[:num_win], proc_weights[ss] if proc_weights is not None else weights[ss], outputs[ss][b], g, _to3(seg_shapes[ss], 1),
                                          buffer_dim + (3 - num_spatial_dims), int(buffer_steps), premultiplied=proc_weights is not None)
                elif lg is mosaic:
                    ops.sw_blend_mosaic(mosaic, _factored_map(imp_key, imp, roi3, mode, sigma_scale, dev) if imp_key is not None else weights[ss], outputs[ss][b])
                elif argmax_dtype is not None:
                    _blend_argmax(lg[:num_win], proc_weights[ss] if proc_weights is not None else weights[ss], outputs[ss][b, 0], g,
                                  _to3(seg_shapes[ss], 1), premultiplied=proc_weights is not None)
                elif proc_weights is not None:
                    ops.sw_blend(lg[:num_win], proc_weights[ss], outputs[ss][b], g, _to3(seg_shapes[ss], 1), premultiplied=True)
                else:
                    ops.sw_blend(lg[:num_win], weights[ss], outputs[ss][b], g, _to3(seg_shapes[ss], 1))

    # back to the caller's rank / crop the padding (utils.py:300-313) / output device
    finals = []
    for ss, o in enumerate(outputs):
        z = zscales[ss]
        osz = [int(i * zz) for i, zz in zip(image_size, z)] if z else list(image_size)
        o = o.reshape((batch_size, o.shape[1]) + tuple(osz))
        if any(pad_size):
            zoom = [sd / float(rd) for sd, rd in zip(o.shape[2:], roi_size)]
            cut = []
            for sp in range(num_spatial_dims):
                si = num_spatial_dims - sp - 1
                cut.insert(0, slice(int(round(pad_size[sp * 2] * zoom[si])), int(round((pad_size[sp * 2] + image_size_[si]) * zoom[si]))))
            o = o[(slice(None), slice(None), *cut)]
        if o.device != out_device:
            o = o.to(out_device)
        if meta_src is not None:
            o = _restore_meta(o, meta_src)
        finals.append(o)
    if any(pad_size):
        kwargs.update({"pad_size": pad_size})
    return _pack_struct(finals, dict_keys)


def _buffered_batches(b, vol3, in_ch, grid3, starts, roi_size, roi3, num_win, sw_batch_size, predictor, process_fn, with_coord, imp, buffer_dim, buffer_steps,
                      shard, nb, dtype, dev, args, kwargs, logits):
    """The predictor calls of the reference's buffered schedule for image `b` (monai/inferers/utils.py:215-253 with `_create_buffered_slices`, :324-348): windows
    stably sorted by their start along `buffer_dim`; flush boundaries after every min(#distinct starts, buffer_steps) distinct starts; batches of `sw_batch_size`
    sorted windows that END at a boundary; `with_coord` hands over the sorted slices; only the first of several outputs is kept (utils.py:244).  Every window's
    (weighted, when a `process_fn` returned the map) prediction goes to row = its row-major index of ONE all-window buffer: `mh_sw_blend_buffered_f32` then sums in
    the buffered order.  -> (dict keys of the predictor's output or None, [logits], the count's weight map or None = the importance map)."""
    import itertools

    import numpy as np

    nsp = len(roi_size)
    wins = np.asarray(list(itertools.product(*[range(len(s)) for s in starts])), dtype=np.int64)             # [num_win, nsp] per-axis window numbers, row-major
    start_along = np.asarray(starts[buffer_dim], dtype=np.int64)[wins[:, buffer_dim]]
    order = np.argsort(start_along, kind="mergesort")
    _, counts = np.unique(start_along[order], return_counts=True)
    b_ends = np.cumsum(counts).tolist()
    x = [0, *b_ends][:: min(len(b_ends), buffer_steps)]
    if x[-1] < b_ends[-1]:
        x.append(b_ends[-1])
    win_buf = torch.empty((sw_batch_size, in_ch) + tuple(roi3), dtype=dtype, device=dev)
    imp_dev = imp.to(dev)
    dict_keys, count_map = None, None
    if logits is None and hasattr(predictor, "out_channels"):
        # the fit decision BEFORE any predictor call where the class count is known: a stateful predictor / process_fn must not see a batch twice because the
        # all-window buffer turned out not to fit after the first one (the caller then retries in the plain order)
        logits = [_alloc_logits(shard, nb, int(predictor.out_channels), roi3, dtype, dev)]
    for gi in range(len(x) - 1):
        for g0 in range(x[gi], x[gi + 1], sw_batch_size):
            idx = [int(order[i]) for i in range(g0, min(g0 + sw_batch_size, x[gi + 1]))]
            k = 0
            while k < len(idx):                 # one gather launch per run of consecutive window indices (the whole batch when buffer_dim is the first axis)
                run = 1
                while k + run < len(idx) and idx[k + run] == idx[k] + run:
                    run += 1
                ops.window_extract(vol3, grid3, idx[k], run, roi3, win_buf[k : k + run])
                k += run
            win_data = win_buf[: len(idx)].reshape((len(idx), in_ch) + tuple(roi_size))
            if with_coord:
                coords = [[slice(b, b + 1), slice(None)] + [slice(int(starts[d][wins[w, d]]), int(starts[d][wins[w, d]]) + int(roi_size[d])) for d in range(nsp)] for w in idx]
                seg_out = predictor(win_data, coords, *args, **kwargs)
            else:
                seg_out = predictor(win_data, *args, **kwargs)
            dict_keys, segs = _flatten_struct(seg_out)
            w_t = None
            if process_fn is not None:
                segs, w_t = process_fn(segs, win_data, imp_dev)
                segs = tuple(segs) if isinstance(segs, (list, tuple)) else (segs,)
                if w_t.dim() == nsp:
                    w_t = w_t[None, None]
                w_t = w_t.to(dtype=dtype, device=dev)
            seg = segs[0]
            if tuple(seg.shape[2:]) != tuple(roi_size):
                raise NotImplementedError("monai_amd: buffer_steps with a predictor whose output is not window-sized is not on the HIP path (the reference fails there too)")
            if seg.dtype != dtype and seg.is_floating_point():
                seg = seg.to(dtype)
            _lib.require_device(seg)
            if logits is None:
                logits = [_alloc_logits(shard, nb, int(seg.shape[1]), roi3, dtype, dev)]
            elif int(seg.shape[1]) != int(logits[0].shape[1]):
                raise RuntimeError(f"monai_amd: the predictor returned {int(seg.shape[1])} channels, its `out_channels` says {int(logits[0].shape[1])}")
            for k, w in enumerate(idx):
                dst = logits[0][w]
                if w_t is None:
                    dst.copy_(seg[k].reshape(dst.shape))
                else:
                    torch.mul(seg[k].reshape(dst.shape), w_t[0 if w_t.shape[0] == 1 else k].reshape((-1,) + tuple(dst.shape[1:])), out=dst)      # `p * w_t`
        if gi == 0 and b == 0 and process_fn is not None:
            count_map = w_t[0, 0].reshape(tuple(roi3)).contiguous().clone()          # the count map is built at the FIRST flush from the map current then (utils.py:264-275)
    if dict_keys is not None:
        dict_keys = dict_keys[:1]
    return dict_keys, logits, count_map


# ---- importance maps: host evaluation and upload happen once per (patch size, mode, sigma, dtype), not once per call -------------------------
# The map is evaluated on the host (bit-identical to the reference's CPU map) and uploaded from pageable memory -- a synchronous copy that is
# stream-ordered behind everything already enqueued.  Issued per call, right in front of the blend, that copy made the host wait for the whole
# predictor queue and left the GPU idle while the blend's arguments were being built (0.4 ms per 512^3 volume, visible as the gap between the HIP-event
# span and the kernel's own duration in profiles/r03_bench_kernel_trace_stats_v2.txt).  Result-neutral: the same values, kept.
_HOST_MAPS: dict = {}
_DEVICE_MAPS: dict = {}
_MAPS_LOCK = threading.RLock()      # inferers may run concurrently in threads (a server, DataLoader workers): lookups and the clear-all eviction are atomic


def _host_importance_map(patch_size, mode, sigma_scale, dtype):
    key = (tuple(int(v) for v in patch_size), str(mode), tuple(float(v) for v in ensure_tuple(sigma_scale)), dtype)
    with _MAPS_LOCK:
        hit = _HOST_MAPS.get(key)
        if hit is None:
            if len(_HOST_MAPS) >= 8:
                _HOST_MAPS.clear()
                _DEVICE_MAPS.clear()
            hit = _HOST_MAPS[key] = compute_importance_map(patch_size, mode=mode, sigma_scale=sigma_scale, device="cpu", dtype=dtype)
    return hit, key


def _factored_map(imp_key, host_map, roi3, mode, sigma_scale, dev) -> torch.Tensor:
    """[gz | gy | gx | floor] on the device for the mosaic blend (it re-forms the importance map from its factors in registers: same fp32 values, no
    roi^3 map competing with the logits stream for L2), or the full map when the factorisation is not available (non-3-D roi, alignment)."""
    key = ("factored",) + imp_key + (str(dev),)
    with _MAPS_LOCK:
        hit = _DEVICE_MAPS.get(key)
        if hit is None:
            fac = importance_map_factors(imp_key[0], mode, sigma_scale) if len(imp_key[0]) == 3 and (roi3[0] + roi3[1]) % 4 == 0 else None
            if fac is None:     # the caller's own host map (never re-read from the cache: another thread may have evicted it)
                hit = _on_device(host_map.reshape(roi3).contiguous(), dev, cache_key=imp_key + (tuple(roi3),))
            else:
                hit = torch.cat([fac[0], fac[1], fac[2], torch.tensor([fac[3]], dtype=torch.float32)]).to(dev)
            _DEVICE_MAPS[key] = hit
    return hit


def _on_device(host_map: torch.Tensor, dev, cache_key=None) -> torch.Tensor:
    if cache_key is None:
        return host_map.to(dev)
    key = cache_key + (str(dev),)
    with _MAPS_LOCK:
        hit = _DEVICE_MAPS.get(key)
        if hit is None:
            if len(_DEVICE_MAPS) >= 16:
                _DEVICE_MAPS.clear()
            hit = _DEVICE_MAPS[key] = host_map.to(dev)
    return hit


def _finish_reference_result(result, private: dict):
    """a call that fell through to the reference's sliding_window_inference (CPU input, float64, autograd ...) with the fused-argmax switch set:
    AsDiscrete(argmax=True) of the reference's blended result, per output"""
    dtype = private.get("_monai_amd_argmax")
    if dtype is None:
        return result
    keys, parts = _flatten_struct(result)
    return _pack_struct([t.argmax(dim=1, keepdim=True).to(dtype) for t in parts], keys)


sliding_window_inference.__wrapped__._mh_after_reference = _finish_reference_result


def _window_stride(dense: int) -> int:
    """Floats between consecutive windows' logits.  The blend reads the K class blocks of up to 8 (27) covering windows of a
    voxel concurrently; with the dense stride (K * roi * 4 B, a multiple of 128 KiB at 96^3) those streams fall on the same HBM
    channels.  Large windows get a stride of 17 x 256 B modulo 128 KiB (measured with tools/ubench/hbm_stream.hip and
    tools/blend_bench.py, profiles/r02_*); MONAI_AMD_LOGITS_PAD (floats, multiple of 4) overrides, 0 = dense."""
    env = os.environ.get("MONAI_AMD_LOGITS_PAD")
    if env is not None and env != "":
        pad = max(0, int(env))
        return dense + pad - pad % 4
    if dense % 4 or dense * 4 < (1 << 20):
        return dense
    pad_bytes = (17 * 256 - dense * 4) % (1 << 17)
    return dense + pad_bytes // 4


def _alloc_logits(shard, nb: int, k: int, seg3, dtype, dev) -> torch.Tensor:
    """Logits of every window of the image, [num_win (padded to whole rounds when sharded), K, *seg3] as a view of one flat
    buffer with a padded window stride (`_window_stride`): the predictor writes its windows' rows, window sharding completes
    the others (`flat_rows`), the blend reads it once."""
    rows = shard.padded_windows(nb)
    dense = k * seg3[0] * seg3[1] * seg3[2]
    ws = _window_stride(dense)
    need = rows * ws * 4
    # the fit decision must be the same on every rank (a rank that went slab-wise alone would dead-lock the others' collectives):
    # under window sharding the budget is the MINIMUM over the ranks
    limit = _logits_budget(dev)
    if dev.type == "cuda":
        free, _ = torch.cuda.mem_get_info(dev)
        free = float(shard.agree_batch(int(free), dev))
        if need > 0.9 * free:
            raise _LogitsDoNotFit(need, free)
    if limit is not None and need > limit:
        raise _LogitsDoNotFit(need, limit)
    flat = torch.empty(rows * ws, dtype=dtype, device=dev)
    return flat.as_strided((rows, k) + tuple(seg3), (ws, seg3[0] * seg3[1] * seg3[2], seg3[1] * seg3[2], seg3[2], 1))


def _alloc_mosaic(predictor, shard, argmax_dtype, k: int, grid3, roi3, dtype, dev):
    """The mosaic logits layout (ops.LogitsMosaic: the windows of one residue class per axis as dense arrays, so the blend reads long runs instead of
    384-byte pieces of 8 ... 27 x K window blocks) when the path allows it: a predictor whose last kernel can write it (`forward_into_windows`), one GPU
    (window sharding gathers window-major rows), the plain blend (the fused-argmax epilogue reads window-major), a regular grid with <= 4 residue
    classes, K <= 8.  Same result bits either way; MONAI_AMD_LOGITS_LAYOUT=windows keeps the window-major buffer.  The same fit rule as _alloc_logits."""
    if (not hasattr(predictor, "forward_into_windows") or shard.sharded or argmax_dtype is not None or dtype != torch.float32
            or os.environ.get("MONAI_AMD_LOGITS_LAYOUT") == "windows" or not ops.LogitsMosaic.supported(grid3, roi3, k)):
        return None
    layout = ops.LogitsMosaic(grid3, roi3, k, dev, dtype, allocate=False)
    need = 4.0 * layout.total
    limit = _logits_budget(dev)
    if dev.type == "cuda":
        free, _ = torch.cuda.mem_get_info(dev)
        if need > 0.9 * free:
            raise _LogitsDoNotFit(need, float(free))
    if limit is not None and need > limit:
        raise _LogitsDoNotFit(need, limit)
    return layout.allocate()


def _blend_argmax(logits, imp, labels, grid, seg3, premultiplied: bool) -> None:
    """labels [D, H, W] = argmax over the classes of the blended logits, in ONE pass when the window grid has
    dense_patch_slices' regular form (mh_sw_blend_argmax_f32); otherwise blend, then the channel argmax kernel."""
    try:
        ops.sw_blend_argmax(logits, imp, labels, grid, seg3, int(logits.shape[1]), premultiplied=premultiplied)
    except RuntimeError as e:
        if "irregular window starts" not in str(e):
            raise
        tmp = torch.empty((int(logits.shape[1]),) + tuple(labels.shape), dtype=logits.dtype, device=logits.device)
        ops.sw_blend(logits, imp, tmp, grid, seg3, premultiplied=premultiplied)
        labels.copy_(ops.channel_reduce("argmax", tmp)[0].to(labels.dtype))


def sliding_window_argmax(inputs, roi_size, sw_batch_size, predictor, *args, labels_dtype=torch.float32, **kwargs):
    """``sliding_window_inference`` followed by ``AsDiscrete(argmax=True)`` -- the post-processing step segmentation bundles put
    behind the inferer (monai/transforms/post/array.py:132-237: ``torch.argmax(img, dim=0, keepdim=True)`` converted to float32)
    -- with the argmax fused into the blend's epilogue: every voxel's K blended values are formed in registers exactly as the
    blend forms them and only the label is written (K x 4 B of output traffic per voxel -> 4 B, or 1 B with
    ``labels_dtype=torch.uint8``).  Same positional / keyword arguments as ``sliding_window_inference``; returns
    ``[B, 1, *spatial]`` labels (tuple / dict outputs: one label map per output), bit-identical to
    ``AsDiscrete(argmax=True)`` applied to each batch element of the unfused result (ties -> first index, NaN maximal)."""
    if labels_dtype not in (torch.float32, torch.uint8):
        raise ValueError("sliding_window_argmax: labels_dtype must be torch.float32 (AsDiscrete's output dtype) or torch.uint8")
    return sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, *args, _monai_amd_argmax=labels_dtype, **kwargs)


def flat_rows(logits: torch.Tensor, r0: int, r1: int) -> torch.Tensor:
    """Rows [r0, r1) of an `_alloc_logits` buffer as ONE contiguous 1-D tensor (padding included): what a collective sends."""
    ws = logits.stride(0)
    base = logits._base if logits._base is not None else logits
    return base.reshape(-1)[r0 * ws : r1 * ws]


class _LogitsDoNotFit(RuntimeError):
    """The all-window logits buffer exceeds the memory budget: the caller retries slab by slab (`_slabwise`)."""

    def __init__(self, need: int, budget: float):
        super().__init__(
            f"monai_amd: the all-window logits buffer needs {need / 2**30:.2f} GiB, the budget is {budget / 2**30:.2f} GiB of HBM "
            "(and the volume cannot be cut into slabs along its first spatial axis: a single row of windows does not fit, the image is padded "
            "up to the roi, or the call uses with_coord / process_fn, whose semantics are tied to the whole volume)"
        )
        self.need, self.budget = need, budget


def _logits_budget(dev):
    """Explicit cap on the logits buffer (bytes) from MONAI_AMD_MAX_LOGITS_BYTES -- tests use it to force the slab-wise
    path on small volumes; without it the cap is 90 % of the free HBM (checked in _alloc_logits)."""
    env = os.environ.get("MONAI_AMD_MAX_LOGITS_BYTES")
    return float(env) if env else None


def _slabwise(inputs, roi_size, starts0, need: int, budget: float, call):
    """Volumes whose all-window logits do not fit in HBM: cut the FIRST spatial axis into slabs of whole window rows and run
    the same inference on each slab's sub-volume -- the span of every window row that intersects the slab, so each owned
    output plane sees exactly the windows it sees in the whole volume, at the same positions and in the same order
    (results are bit-identical; the rows shared by two neighbouring sub-volumes are computed twice: one extra row per slab at
    overlap 0.5).  `call(sub_inputs)` runs the normal path; returns the packed outputs of the whole volume."""
    rows = len(starts0)
    roi0 = int(roi_size[0])
    size0 = int(inputs.shape[2])
    per_row = need / rows
    first_of = lambda a: min(r for r in range(rows) if starts0[r] + roi0 > starts0[a])  # noqa: E731
    # greedy grouping of owned rows [a, b): owned + re-computed rows must fit in the budget
    groups, a = [], 0
    while a < rows:
        b = a + 1
        while b < rows and (b + 1 - first_of(a)) * per_row <= budget:
            b += 1
        if (b - first_of(a)) * per_row > budget:
            raise _LogitsDoNotFit(int((b - first_of(a)) * per_row), budget)
        groups.append((a, b))
        a = b
    if len(groups) < 2:
        raise _LogitsDoNotFit(need, budget)
    finals, keys = None, None
    for a, b in groups:
        r0 = first_of(a)
        lo, hi = int(starts0[r0]), int(starts0[b - 1]) + roi0
        own_lo, own_hi = int(starts0[a]), (int(starts0[b]) if b < rows else size0)
        res = call(inputs[:, :, lo:hi])
        keys, parts = _flatten_struct(res)
        if finals is None:
            finals = []
            for t in parts:
                scale = t.shape[2] / float(hi - lo)
                finals.append(torch.empty(tuple(t.shape[:2]) + (int(round(size0 * scale)),) + tuple(t.shape[3:]), dtype=t.dtype, device=t.device))
        for t, f in zip(parts, finals):
            scale = t.shape[2] / float(hi - lo)
            p0, p1, q0 = own_lo * scale, own_hi * scale, lo * scale
            if abs(p0 - round(p0)) > 1e-6 or abs(p1 - round(p1)) > 1e-6 or abs(q0 - round(q0)) > 1e-6:
                raise RuntimeError("monai_amd: slab-wise blending needs output planes aligned with the window rows "
                                   f"(output/input scale {scale} along the first spatial axis)")
            p0, p1, q0 = int(round(p0)), int(round(p1)), int(round(q0))
            f[:, :, p0:p1] = t[:, :, p0 - q0 : p1 - q0]
        del res, parts
    return _pack_struct(finals, keys)


def _restore_meta(out: torch.Tensor, src):
    """Give the result the input's MetaTensor type and metadata (the reference's ``convert_to_dst_type(final_output,
    temp_meta)``, utils.py:316-319) when the input was a MetaTensor-like tensor subclass."""
    try:
        res = type(src)(out)
        if hasattr(res, "copy_meta_from"):
            res.copy_meta_from(src, copy_attr=False)
        return res
    except Exception:
        return out
