"""Optional per-launch timing with HIP events on torch's current stream (the stream every kernel of this package
is enqueued on).  Inactive unless ``start()`` was called -- bench.py uses it to measure the average duration of the
dominant kernels inside the timed region.  Without a GPU (the emulator harness test of bench.py) the spans are host timers."""

from __future__ import annotations

import contextlib
from collections import defaultdict

import torch

_spans = None  # name -> list[(start_event, end_event, work)]
_coarse = False
_COARSE_NAMES = ("sw_blend", "sw_gather_wait")      # spans on the caller's stream while no other stream of the package has work in flight


def start(coarse: bool = False) -> None:
    """coarse=True: only the spans that stay meaningful when the rounds of windows run on several streams (a launch's begin-end interval then covers other
    streams' kernels as well): the inferer keeps its multi-stream schedule.  With the full set active the inferer runs one stream, so every span is a kernel's own time."""
    global _spans, _coarse
    _spans, _coarse = defaultdict(list), bool(coarse)


def active() -> bool:
    return _spans is not None


def overlapped_ok() -> bool:
    return _coarse


def stop() -> dict:
    """-> {name: {"launches": n, "ms_avg": t, "ms_total": T, "work": total work units}} (synchronises)."""
    global _spans, _coarse
    spans, _spans, _coarse = _spans, None, False
    out = {}
    if spans:
        gpu = torch.cuda.is_available()
        if gpu:
            torch.cuda.synchronize()
        for name, evs in spans.items():
            total = sum(a.elapsed_time(b) if gpu else 1e3 * (b - a) for a, b, _ in evs)
            out[name] = {"launches": len(evs), "ms_total": total, "ms_avg": total / len(evs), "work": float(sum(w for _, _, w in evs))}
    return out


@contextlib.contextmanager
def span(name: str, work: float = 0.0):
    if _spans is None or (_coarse and name not in _COARSE_NAMES):
        yield
        return
    if not torch.cuda.is_available():
        import time

        t0 = time.perf_counter()
        try:
            yield
        finally:
            _spans[name].append((t0, time.perf_counter(), work))
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    try:
        yield
    finally:
        b.record()
        _spans[name].append((a, b, work))
