"""Oracle (test infrastructure): sliding-window inference, restated from the reference.

Reference functions followed (paths relative to /root/reference):
  * ``sliding_window_inference``      monai/inferers/utils.py:42-321   (non-buffered path)
  * ``_get_scan_interval``            monai/inferers/utils.py:363-384
  * ``_compute_coords``               monai/inferers/utils.py:351-360
  * ``_flatten_struct/_pack_struct``  monai/inferers/utils.py:387-405
  * ``dense_patch_slices``            monai/data/utils.py:166-206
  * ``get_valid_patch_size``          monai/data/utils.py:343-354
  * ``compute_importance_map``        monai/data/utils.py:1084-1134

The blend is done exactly as the reference does it -- scatter order, in-place ``*=`` on the
predictor output, ``+=`` into the output, one final ``/=`` by the count map -- so that on the same
predictor the result is bitwise identical to the reference (checked in tests/test_oracle_golden.py).
"""

from __future__ import annotations

import itertools
import math
from collections.abc import Mapping, Sequence

import torch
import torch.nn.functional as F


def _rep(v, n):
    if isinstance(v, (str, bytes)) or not isinstance(v, Sequence):
        return (v,) * n
    v = tuple(v)
    if len(v) != n:
        raise ValueError(f"Sequence must have length {n}, got {len(v)}.")
    return v


def resolve_roi_size(roi_size, image_size):
    """``fall_back_tuple(roi_size, image_size)`` -- monai/utils/misc.py:256-299: a component that is
    None / 0 / negative falls back to the image dimension."""
    roi = _rep(roi_size, len(image_size))
    return tuple(int(r) if (r and r > 0) else int(d) for r, d in zip(roi, image_size))


def get_scan_interval(image_size, roi_size, overlap):
    """monai/inferers/utils.py:363-384: ``int(roi * (1 - overlap))``, at least 1; the whole roi when
    the roi spans the image."""
    out = []
    for i, r, o in zip(image_size, roi_size, overlap):
        if r == i:
            out.append(int(r))
        else:
            step = int(r * (1 - o))
            out.append(step if step > 0 else 1)
    return tuple(out)


def dense_patch_starts(image_size, patch_size, scan_interval):
    """Per-axis window start indices, monai/data/utils.py:166-206.

    Window ``k`` along an axis starts at ``k * interval`` pulled back so it ends inside the image;
    the count is the first ``k`` whose window reaches the image end, plus one.  The full window list
    is the cartesian product in row-major order (last axis fastest, ``np.meshgrid(indexing="ij")``).
    """
    patch_size = tuple(min(m, p or m) for m, p in zip(image_size, patch_size))  # get_valid_patch_size
    starts = []
    for size, patch, step in zip(image_size, patch_size, scan_interval):
        if step == 0:
            count = 1
        else:
            upper = int(math.ceil(float(size) / step))
            hit = next((d for d in range(upper) if d * step + patch >= size), None)
            count = hit + 1 if hit is not None else 1
        axis = []
        for k in range(count):
            s = k * step
            s -= max(s + patch - size, 0)
            axis.append(s)
        starts.append(axis)
    return starts, patch_size


def compute_importance_map(patch_size, mode="constant", sigma_scale=0.125, dtype=torch.float32):
    """monai/data/utils.py:1084-1134, computed with torch on the CPU (fp32)."""
    mode = str(getattr(mode, "value", mode)).lower()
    if mode == "constant":
        imp = torch.ones(tuple(patch_size), dtype=torch.float)
    elif mode == "gaussian":
        sig = _rep(sigma_scale, len(patch_size))
        sigmas = [p * s for p, s in zip(patch_size, sig)]
        imp = None
        for i, n in enumerate(patch_size):
            x = torch.arange(start=-(n - 1) / 2.0, end=(n - 1) / 2.0 + 1, dtype=torch.float)
            x = torch.exp(x**2 / (-2 * sigmas[i] ** 2))
            imp = imp.unsqueeze(-1) * x[(None,) * i] if i > 0 else x
    else:
        raise ValueError(f"Unsupported mode: {mode}")
    floor = max(torch.min(imp).item(), 1e-3)
    return torch.clamp_(imp.to(torch.float), min=floor).to(dtype)


def _flatten(out):
    if isinstance(out, torch.Tensor):
        return None, (out,)
    if isinstance(out, Mapping):
        keys = sorted(out.keys())
        return keys, tuple(out[k] for k in keys)
    return None, tuple(out)


def sliding_window_inference(
    inputs: torch.Tensor,
    roi_size,
    sw_batch_size: int,
    predictor,
    overlap=0.25,
    mode="constant",
    sigma_scale=0.125,
    padding_mode="constant",
    cval=0.0,
    roi_weight_map=None,
    buffer_steps=None,
    buffer_dim=-1,
):
    """monai/inferers/utils.py:42-321 on CPU tensors: the non-buffered path, and (buffer_steps > 0) the buffered one -- see `_buffered` below."""
    if buffer_steps is not None and buffer_steps > 0:
        return _buffered(inputs, roi_size, sw_batch_size, predictor, overlap, mode, sigma_scale, padding_mode, cval, roi_weight_map, int(buffer_steps), int(buffer_dim))
    nsp = inputs.dim() - 2
    overlap = _rep(overlap, nsp)
    for o in overlap:
        if o < 0 or o >= 1:
            raise ValueError(f"overlap must be >= 0 and < 1, got {overlap}.")
    dtype = inputs.dtype
    batch, _, *orig_size = inputs.shape
    roi = resolve_roi_size(roi_size, orig_size)

    image_size = tuple(max(orig_size[i], roi[i]) for i in range(nsp))
    pad = []
    for k in range(inputs.dim() - 1, 1, -1):  # utils.py:163-170, last dim first, centred
        diff = max(roi[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad.extend([half, diff - half])
    if any(pad):
        inputs = F.pad(inputs, pad=pad, mode=padding_mode, value=cval)

    interval = get_scan_interval(image_size, roi, overlap)
    starts, patch = dense_patch_starts(image_size, roi, interval)
    windows = [tuple(slice(s, s + patch[d]) for d, s in enumerate(w)) for w in itertools.product(*starts)]
    num_win = len(windows)
    total = num_win * batch

    if patch == tuple(roi) and roi_weight_map is not None:
        imp = roi_weight_map
    else:
        imp = compute_importance_map(patch, mode=mode, sigma_scale=sigma_scale, dtype=dtype)
    if imp.dim() == nsp:
        imp = imp[None, None]
    imp = imp.to(dtype)

    outputs, counts, keys = [], [], None
    for g in range(0, total, sw_batch_size):
        idxs = range(g, min(g + sw_batch_size, total))
        where = [[slice(i // num_win, i // num_win + 1), slice(None)] + list(windows[i % num_win]) for i in idxs]
        if sw_batch_size > 1:
            win = torch.cat([inputs[tuple(w)] for w in where])
        else:
            win = inputs[tuple(where[0])]
        keys, segs = _flatten(predictor(win))
        segs = list(segs)
        w_t = imp
        for ss in range(len(segs)):
            seg_shape = tuple(segs[ss].shape[2:])
            z = None
            if seg_shape != tuple(roi):
                z = [o / float(i) for o, i in zip(seg_shape, roi)]
                w_t = F.interpolate(w_t, seg_shape, mode="nearest-exact")
            if len(outputs) <= ss:
                oshape = [batch, segs[ss].shape[1]]
                oshape += [int(i * s) for i, s in zip(image_size, z)] if z else list(image_size)
                outputs.append(torch.zeros(oshape, dtype=dtype))
                counts.append(torch.zeros([1, 1] + oshape[2:], dtype=dtype))
                for w in windows:
                    if z is not None:
                        w = tuple(slice(int(s.start * zz), int(s.stop * zz)) for s, zz in zip(w, z))
                    counts[-1][(slice(None), slice(None), *w)] += w_t
            segs[ss] *= w_t
            for wh, p in zip(where, segs[ss]):
                wh = list(wh)
                if z:
                    for ax in range(2, len(wh)):
                        wh[ax] = slice(int(wh[ax].start * z[ax - 2]), int(wh[ax].stop * z[ax - 2]))
                outputs[ss][tuple(wh)] += p

    for ss in range(len(outputs)):
        outputs[ss] /= counts[ss]

    if any(pad):  # crop the padding back off, utils.py:300-313
        for ss, o in enumerate(outputs):
            zoom = [sd / rd for sd, rd in zip(o.shape[2:], roi)]
            cut = []
            for sp in range(nsp):
                si = nsp - sp - 1
                cut.insert(
                    0,
                    slice(int(round(pad[sp * 2] * zoom[si])), int(round((pad[sp * 2] + orig_size[si]) * zoom[si]))),
                )
            outputs[ss] = o[(slice(None), slice(None), *cut)]

    if keys is not None:
        return dict(zip(keys, outputs))
    return outputs[0] if len(outputs) == 1 else tuple(outputs)


def _buffered(inputs, roi_size, sw_batch_size, predictor, overlap, mode, sigma_scale, padding_mode, cval, roi_weight_map, buffer_steps, buffer_dim):
    """The buffered schedule of the reference on CPU tensors (single-output predictors), monai/inferers/utils.py:
      * :142-143  a negative `buffer_dim` counts from the last spatial axis;
      * :324-348  `_create_buffered_slices`: windows stably sorted by their start along `buffer_dim`; flush boundaries after every
                  `min(#distinct starts, buffer_steps)` distinct starts; a slab spans [first window's start, last window's end) of its group;
      * :239-253  per group a zero slab buffer `[1, K, *image with the slab's extent]`, `buffer[win] += p * w_t` window by window (sorted order);
      * :264-275  the count map adds `w_t` over ALL windows in the sorted order;
      * :276-284  the finished slab is added to the zero-initialised output (`output[slab] += buffer`: the CPU run has non_blocking = False);
      * :297-298  one final `/=` by the count map."""
    import numpy as np

    nsp = inputs.dim() - 2
    if buffer_dim < -nsp or buffer_dim > nsp:
        raise ValueError(f"buffer_dim must be in [{-nsp}, {nsp}], got {buffer_dim}.")
    if buffer_dim < 0:
        buffer_dim += nsp
    overlap = _rep(overlap, nsp)
    dtype = inputs.dtype
    batch, _, *orig_size = inputs.shape
    roi = resolve_roi_size(roi_size, orig_size)
    image_size = tuple(max(orig_size[i], roi[i]) for i in range(nsp))
    pad = []
    for k in range(inputs.dim() - 1, 1, -1):
        diff = max(roi[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad.extend([half, diff - half])
    if any(pad):
        inputs = F.pad(inputs, pad=pad, mode=padding_mode, value=cval)
    interval = get_scan_interval(image_size, roi, overlap)
    starts, patch = dense_patch_starts(image_size, roi, interval)
    wins = np.asarray([[(s, s + patch[d]) for d, s in enumerate(w)] for w in itertools.product(*starts)])      # [num_win, nsp, 2]
    wins = wins[np.argsort(wins[:, buffer_dim, 0], kind="mergesort")]
    num_win = len(wins)
    along = wins[:, buffer_dim]
    _, counts = np.unique(along[:, 0], return_counts=True)
    b_ends = np.cumsum(counts).tolist()
    x = [0, *b_ends][:: min(len(b_ends), int(buffer_steps))]
    if x[-1] < b_ends[-1]:
        x.append(b_ends[-1])
    groups = [(x[i], x[i + 1]) for i in range(len(x) - 1)]

    if patch == tuple(roi) and roi_weight_map is not None:
        imp = roi_weight_map
    else:
        imp = compute_importance_map(patch, mode=mode, sigma_scale=sigma_scale, dtype=dtype)
    if imp.dim() == nsp:
        imp = imp[None, None]
    imp = imp.to(dtype)

    output = count = None
    for b in range(batch):
        for g0, g1 in groups:
            c_start, c_end = int(along[g0, 0]), int(along[g1 - 1, 1])
            buf = None
            for i0 in range(g0, g1, sw_batch_size):
                idx = range(i0, min(i0 + sw_batch_size, g1))
                sl = [tuple(slice(int(a), int(e)) for a, e in wins[i]) for i in idx]
                win = torch.cat([inputs[(slice(b, b + 1), slice(None)) + s] for s in sl]) if sw_batch_size > 1 else inputs[(slice(b, b + 1), slice(None)) + sl[0]]
                seg = predictor(win)
                if not isinstance(seg, torch.Tensor):
                    raise ValueError("oracle: the buffered schedule is restated for single-tensor predictors")
                if buf is None:
                    sp = list(image_size)
                    sp[buffer_dim] = c_end - c_start
                    buf = torch.zeros([1, seg.shape[1], *sp], dtype=dtype)
                for p, s in zip(seg, sl):
                    s = list(s)
                    off = s[buffer_dim].start - c_start
                    s[buffer_dim] = slice(off, off + roi[buffer_dim])
                    buf[(slice(0, 1), slice(None), *s)] += p * imp
            if output is None:
                output = torch.zeros([batch, buf.shape[1], *image_size], dtype=dtype)
                count = torch.zeros([1, 1, *image_size], dtype=dtype)
                for i in range(num_win):
                    count[(slice(None), slice(None), *(slice(int(a), int(e)) for a, e in wins[i]))] += imp
            o = [slice(b, b + 1), slice(None)] + [slice(None)] * nsp
            o[buffer_dim + 2] = slice(c_start, c_end)
            output[tuple(o)] += buf
    output /= count
    if any(pad):
        cut = []
        for sp in range(nsp):
            si = nsp - sp - 1
            cut.insert(0, slice(pad[sp * 2], pad[sp * 2] + orig_size[si]))
        output = output[(slice(None), slice(None), *cut)]
    return output
