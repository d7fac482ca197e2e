"""Optional per-launch timing with HIP events on torch's current stream (the stream every kernel of this package
is enqueued on).  Inactive unless ``start()`` was called -- bench.py uses it to measure the average duration of the
dominant kernels inside the timed region.  Without a GPU (the emulator harness test of bench.py) the spans are host timers."""

from __future__ import annotations

import contextlib
from collections import defaultdict

import torch

_spans = None  # name -> list[(start_event, end_event, work)]


def start() -> None:
    global _spans
    _spans = defaultdict(list)


def stop() -> dict:
    """-> {name: {"launches": n, "ms_avg": t, "ms_total": T, "work": total work units}} (synchronises)."""
    global _spans
    spans, _spans = _spans, None
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
    if _spans is None:
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
