"""TEST INFRASTRUCTURE: the oracle's per-window network on several host processes at once.

The CPU reference of the headline configuration is 1000 windows x ~0.6 s of oneDNN work -- ten minutes on one group of 32 threads, while a GPU
box has 128-256 hardware threads.  Windows are independent until the blend, so `PoolPredictor` computes the predictions of consecutive window
batches in `procs` worker processes (each `threads` ATen threads) and hands them to `oracle.sliding_window_inference` IN ORDER: the blend -- the
only order-dependent part of the reference (monai/inferers/utils.py:264-298: `seg *= w`, `out[win] += seg`, `cnt[win] += w`, `out /= cnt`) -- still
runs in the calling process, window by window, exactly as before.  The per-window values are those of `predict_batch` (the same ATen CPU operators
as the single-process oracle); nothing here is ever imported by monai_amd/.

A worker re-derives its windows from the batch index (the same `dense_patch_starts` walk as the caller), so only an index travels to it and a
`[sw_batch, K, *roi]` tensor travels back.
"""

from __future__ import annotations

import itertools
import os
from typing import Callable, Optional

import torch

_STATE: dict = {}


def _init(vol, roi, sw_batch, overlap, threads, factory, factory_args):
    """worker initialiser: the volume arrives once (shared memory), the predictor is built once per worker"""
    from . import sliding_window as osw

    torch.set_num_threads(int(threads))
    image_size = tuple(vol.shape[2:])
    interval = osw.get_scan_interval(image_size, roi, overlap)
    starts, patch = osw.dense_patch_starts(image_size, roi, interval)
    _STATE.update(vol=vol, windows=[tuple(slice(s, s + patch[d]) for d, s in enumerate(w)) for w in itertools.product(*starts)],
                  sw_batch=int(sw_batch), predict=factory(*factory_args))


def _work(g: int):
    """predictions of window batch `g` (windows g .. g + sw_batch - 1 of image 0), as the oracle's loop would cut them"""
    vol, windows, nb = _STATE["vol"], _STATE["windows"], _STATE["sw_batch"]
    idx = range(g, min(g + nb, len(windows)))
    win = torch.cat([vol[(slice(0, 1), slice(None)) + windows[i]] for i in idx]) if nb > 1 else vol[(slice(0, 1), slice(None)) + windows[idx[0]]]
    with torch.no_grad():
        return _STATE["predict"](win)


class PoolPredictor:
    """`predictor(win)` for `oracle.sliding_window_inference` that returns the pool's result for the next batch (the oracle asks for the batches in
    ascending order, exactly once each).  `factory(*factory_args)` must be a picklable module-level callable returning the per-batch predictor."""

    def __init__(self, vol: torch.Tensor, roi, sw_batch: int, overlap, factory: Callable, factory_args=(), procs: Optional[int] = None, threads: Optional[int] = None):
        import torch.multiprocessing as mp

        ncpu = os.cpu_count() or 1
        self.threads = int(threads or min(32, ncpu))
        self.procs = int(procs or max(1, min(8, ncpu // self.threads)))
        from . import sliding_window as osw

        image_size = tuple(vol.shape[2:])
        starts, _ = osw.dense_patch_starts(image_size, tuple(roi), osw.get_scan_interval(image_size, tuple(roi), tuple(overlap)))
        self.num_win = 1
        for s in starts:
            self.num_win *= len(s)
        self.sw_batch = int(sw_batch)
        self.batch_timeout_s = float(os.environ.get("ORACLE_POOL_BATCH_TIMEOUT_S", "900"))
        if vol.shape[0] != 1:
            raise ValueError("PoolPredictor: one image per call")
        vol = vol.contiguous().share_memory_()
        ctx = mp.get_context("spawn")          # fork after ATen has started its thread pool is not safe
        self._pool = ctx.Pool(self.procs, initializer=_init, initargs=(vol, tuple(roi), self.sw_batch, tuple(overlap), self.threads, factory, tuple(factory_args)))
        self._it = self._pool.imap(_work, range(0, self.num_win, self.sw_batch), chunksize=1)

    def __call__(self, win: torch.Tensor) -> torch.Tensor:
        out = self._it.next(timeout=self.batch_timeout_s)        # a dead worker must not hang the caller: multiprocessing.TimeoutError instead
        if out.shape[0] != win.shape[0]:
            raise RuntimeError(f"PoolPredictor: batch of {win.shape[0]} windows asked, {out.shape[0]} computed (the caller's batching differs from the workers')")
        return out

    def close(self):
        self._pool.terminate()
        self._pool.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---- picklable predictor factories -------------------------------------------------------------------------------------------
def basic_unet_factory(state_dict):
    from .basic_unet import basic_unet_forward

    return lambda w: basic_unet_forward(state_dict, w)


def basic_unet_factory_no_onednn(state_dict):
    """the same network with oneDNN switched off in this worker: ATen's native CPU convolution -- another summation order of the same fp32 arithmetic
    (bench.py: reference_self_spread)"""
    from .basic_unet import basic_unet_forward

    torch.backends.mkldnn.enabled = False
    return lambda w: basic_unet_forward(state_dict, w)


def unetr_factory(state_dict):
    from .unetr import unetr_forward

    return lambda w: unetr_forward(state_dict, w)
