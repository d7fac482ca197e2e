"""Window sharding across the GPUs of one node.

The sliding-window path shards naturally: windows are independent until the blend.  With sharding enabled the window
index space is cut into ROUNDS of ``world x n`` consecutive windows (n = nb, the windows per predictor launch, in the main rounds;
about nb / 4 in the tail rounds at the end -- ``WindowShard.schedule``); in a round starting at row ``base``
rank r (one process per GPU, ``torch.distributed`` backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests) runs the
predictor on windows ``[base + r*n, +n)`` and writes their logits straight into its slice of the full
``[rows, K, roi]`` buffer (row == window index); one ``all_gather_into_tensor`` per round -- issued asynchronously, so it
travels over xGMI while the next round computes -- completes that round's rows on every rank.  After the last round
every rank holds every window's logits and the deterministic gather blend runs unchanged: the result is identical to
the single-GPU result on every rank (the blend never depended on who computed a window).  The reference has no
counterpart (monai/utils/dist.py:59-140 only gathers metrics).

Nothing here runs unless ``enable_window_sharding()`` was called (bench.py does, for --gpus > 1): the public
``SlidingWindowInferer`` signature is unchanged.  With ONE rank the schedule degenerates to the unsharded loop and no collective is issued --
unless the sharding was enabled with ``force=True`` (or MONAI_AMD_SHARD_FORCE=1 is set): then a one-rank group takes exactly the code a
larger one takes (round schedule with its tail, padded row buffer, the in-place probe, one asynchronous ``all_gather_into_tensor`` per round,
``_Pending.wait``), which is how the RCCL path is exercised on a one-GPU box (tests/test_e2e_gpu.py, ``bench.py --force-shard``).
"""

from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

_GROUP = None
_ENABLED = False
_FORCE = False


def enable_window_sharding(group=None, force: bool = False) -> None:
    """Shard the windows of every following sliding_window_inference call over `group` (default: WORLD).
    force: a group of ONE rank runs the sharded code too (rounds, row padding, the collectives) instead of the unsharded loop."""
    global _GROUP, _ENABLED, _FORCE
    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("monai_amd.parallel: torch.distributed is not initialised")
    _GROUP, _ENABLED, _FORCE = group, True, bool(force) or os.environ.get("MONAI_AMD_SHARD_FORCE") == "1"


def disable_window_sharding() -> None:
    global _GROUP, _ENABLED, _FORCE
    _GROUP, _ENABLED, _FORCE = None, False, False


class window_sharding:
    """Scoped form of enable / disable_window_sharding: `with parallel.window_sharding(group): out = inferer(vol, net)` -- the previous state (off, or an outer scope's
    group) is restored on exit, also when the body raises.  The switch itself stays a property of the process (one inference stream per rank, the layout bench.py and
    the reference's one-process-per-GPU launchers use); two threads of one rank that shard over different groups must serialise their scopes."""

    def __init__(self, group=None, force: bool = False):
        self.group, self.force = group, force

    def __enter__(self):
        self._saved = (_GROUP, _ENABLED, _FORCE)
        enable_window_sharding(self.group, self.force)
        return self

    def __exit__(self, *exc):
        global _GROUP, _ENABLED, _FORCE
        _GROUP, _ENABLED, _FORCE = self._saved
        return False


# ---- in-place all-gather: probed once per process group -------------------------------------------------------------------------
# RCCL / NCCL run all_gather in place when the send buffer is exactly the rank's own slot of the receive buffer (recv + rank * count);
# torch.distributed documents no aliasing guarantee, and other backends (or a future release) may reject or mis-handle it.  One tiny
# probe collective per group decides: rows come back right -> the in-place form (no send copy); an exception or wrong rows -> every
# round sends a private copy of its rows.  All ranks take the same branch (the verdict is all-reduced).
_INPLACE_OK: dict = {}


def _inplace_gather_ok(group, device) -> bool:
    key = (id(group), str(device))
    if os.environ.get("MONAI_AMD_GATHER_INPLACE") == "0":       # force the out-of-place form (tests exercise both)
        return False
    if key not in _INPLACE_OK:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ok = 1
        try:
            buf = torch.full((world * 64,), -1.0, dtype=torch.float32, device=device)
            mine = buf[rank * 64:(rank + 1) * 64]
            mine.copy_(torch.arange(64, dtype=torch.float32, device=device) + 1000.0 * rank)
            dist.all_gather_into_tensor(buf, mine, group=group)
            want = (torch.arange(64, dtype=torch.float32, device=device)[None] + 1000.0 * torch.arange(world, dtype=torch.float32, device=device)[:, None]).reshape(-1)
            ok = int(torch.equal(buf, want))
        except Exception:      # a backend that refuses aliased buffers
            ok = 0
        t = torch.tensor([ok], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        _INPLACE_OK[key] = bool(int(t.item()))
    return _INPLACE_OK[key]


class _Pending:
    """An in-flight collective together with the send buffer it reads."""

    def __init__(self, work, keep):
        self.work, self.keep = work, keep

    def wait(self):
        self.work.wait()
        self.keep = None


@dataclass
class WindowShard:
    """Rank-local view of the window index space [0, num_win)."""

    num_win: int
    world: int
    rank: int
    chunk: int   # windows per rank (equal on every rank; the last ranks may own fewer real windows)
    lo: int      # first global window index of this rank (contiguous partition, see `partition`)
    hi: int      # one past the last REAL window of this rank
    group: Optional[object] = None
    force: bool = False      # one rank, but the sharded code (see the module docstring)

    @property
    def base(self) -> int:
        return self.lo

    @property
    def sharded(self) -> bool:
        """does this call run the round schedule and its collectives?"""
        return self.world > 1 or self.force

    def all_gather(self, local: torch.Tensor) -> torch.Tensor:
        """local [chunk, ...] -> [world * chunk, ...] with global window w at row w (identity when not sharded)."""
        if not self.sharded:
            return local
        out = torch.empty((self.world * self.chunk,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    # ---- round-interleaved schedule: communication of round q overlaps the computation of round q + 1 -------------
    def agree_batch(self, nb: int, device) -> int:
        """Windows per predictor launch, identical on every rank (the minimum of the ranks' own choices)."""
        if not self.sharded:
            return nb
        t = torch.tensor([int(nb)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return int(t.item())

    def schedule(self, nb: int) -> list:
        """[(first row, windows per rank)] of every round: the window index space [0, num_win) is cut into MAIN rounds of `world * nb` consecutive windows and, at its end,
        TAIL rounds of `world * tail` windows (tail ~ nb / 4).  In a round starting at row `base` rank r owns rows [base + r * n, base + (r + 1) * n); rows beyond
        num_win are padding (nobody computes them).  Why a tail: a round's all-gather travels while the NEXT round computes, so what the last round sends is exposed in
        front of the blend -- with 8 GPUs and 63 windows per launch that is 504 windows = 7.8 GB per rank (~23 ms over xGMI against ~107 ms of compute per rank);
        five tail rounds of 13 windows per rank leave 1.6 GB (~5 ms) exposed and cost the deep U-Net levels a few per cent of fill on 1/8 of the windows.
        A pure function of (num_win, world, nb): every rank derives the same schedule.  MONAI_AMD_TAIL_ROUNDS=0 keeps equal rounds (measurement switch)."""
        if not self.sharded:
            return [(w0, min(nb, self.num_win - w0)) for w0 in range(0, self.num_win, nb)]
        span = self.world * nb
        tail = nb // 4 if (nb >= 8 and os.environ.get("MONAI_AMD_TAIL_ROUNDS") != "0") else 0
        # the tail also has to hide the LAST MAIN round's gather behind its own compute: it keeps between half a round and one and a half (a round's rows cross xGMI in
        # about the time 0.4 rounds of windows take to compute at 8 GPUs)
        main = max(0, (2 * self.num_win - span) // (2 * span)) if tail else -(-self.num_win // span)
        out = [(q * span, nb) for q in range(main)]
        rem = self.num_win - main * span
        if rem > 0:       # tail > 0: span / 2 <= rem < 3 span / 2 windows (fewer when the whole image is smaller) in T rounds of (nearly) equal size
            t_rounds = -(-rem // (self.world * tail))
            per = -(-rem // (self.world * t_rounds))
            base = main * span
            out += [(base + t * self.world * per, per) for t in range(t_rounds)]
        return out

    def rounds(self, nb: int):
        """(first global window, number of real windows) of THIS rank in each round of `schedule(nb)`"""
        if not self.sharded:
            return self.schedule(nb)
        out = []
        for base, n in self.schedule(nb):
            w0 = base + self.rank * n
            out.append((w0, max(0, min(n, self.num_win - w0))))
        return out

    def padded_windows(self, nb: int) -> int:
        """rows of the all-window logits buffer: the windows, padded to whole rounds when sharded (row index == window index)"""
        if not self.sharded:
            return self.num_win
        base, n = self.schedule(nb)[-1]
        return base + self.world * n

    def gather_round(self, full: torch.Tensor, q: int, nb: int):
        """Complete the rows of round `q` of `schedule(nb)` on every rank (this rank has written its own slot of them).
        `full` is an inferer logits buffer (rows `stride(0)` floats apart inside one flat allocation): the round's rows are one
        contiguous span, this rank's rows are its own slot of that span, so the all-gather runs IN PLACE -- no send copy.
        Returns the async work handle (None when not sharded)."""
        if not self.sharded:
            return None
        from .inferers.utils import flat_rows

        base, n = self.schedule(nb)[q]
        out = flat_rows(full, base, base + self.world * n)
        mine = flat_rows(full, base + self.rank * n, base + (self.rank + 1) * n)
        # the slot arithmetic the in-place form rests on: equal contiguous slots, this rank's rows exactly at recv + rank * count
        if out.numel() != self.world * mine.numel() or mine.data_ptr() != out.data_ptr() + self.rank * mine.numel() * mine.element_size():
            raise RuntimeError("monai_amd.parallel: the round's rows are not equal contiguous slots of the logits buffer")
        if _inplace_gather_ok(self.group, full.device):
            work = dist.all_gather_into_tensor(out, mine, group=self.group, async_op=True)
            return _Pending(work, None)
        send = mine.clone()                 # out-of-place fallback: a private send buffer, kept alive until the collective is done
        work = dist.all_gather_into_tensor(out, send, group=self.group, async_op=True)
        return _Pending(work, send)


def partition(num_win: int, world: int, rank: int, group=None, force: bool = False) -> WindowShard:
    """Contiguous equal chunks of ceil(num_win / world) windows: rank r owns [r*chunk, min((r+1)*chunk, num_win))."""
    chunk = (num_win + world - 1) // world
    lo = min(rank * chunk, num_win)
    hi = min(lo + chunk, num_win)
    return WindowShard(num_win, world, rank, chunk, lo, hi, group, force)


def window_shard(num_win: int) -> WindowShard:
    if not _ENABLED:
        return WindowShard(num_win, 1, 0, num_win, 0, num_win, None)
    world = dist.get_world_size(_GROUP)
    rank = dist.get_rank(_GROUP)
    return partition(num_win, world, rank, _GROUP, _FORCE)


def inplace_gather_verdicts() -> dict:
    """{(group id, device): did the in-place all-gather probe pass} for every group probed so far in this process (tests and bench.py report it)"""
    return dict(_INPLACE_OK)
