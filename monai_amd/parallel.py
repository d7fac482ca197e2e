"""Window sharding across the GPUs of one node.

The sliding-window path shards naturally: windows are independent until the blend.  With sharding enabled each
rank (one process per GPU, ``torch.distributed`` backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests) runs the
predictor on one contiguous, equal-sized range of window indices and a single ``all_gather_into_tensor`` of the
per-window logits rebuilds the full ``[num_win, K, roi]`` buffer on every rank; the deterministic gather blend then
runs unchanged, so the result is identical to the single-GPU result on every rank.  The reference has no
counterpart (monai/utils/dist.py:59-140 only gathers metrics).

Nothing here runs unless ``enable_window_sharding()`` was called (bench.py does, for --gpus > 1): the public
``SlidingWindowInferer`` signature is unchanged.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

_GROUP = None
_ENABLED = False


def enable_window_sharding(group=None) -> None:
    """Shard the windows of every following sliding_window_inference call over `group` (default: WORLD)."""
    global _GROUP, _ENABLED
    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("monai_amd.parallel: torch.distributed is not initialised")
    _GROUP, _ENABLED = group, True


def disable_window_sharding() -> None:
    global _GROUP, _ENABLED
    _GROUP, _ENABLED = None, False


@dataclass
class WindowShard:
    """Rank-local view of the window index space [0, num_win)."""

    num_win: int
    world: int
    rank: int
    chunk: int   # windows per rank (equal on every rank; the last ranks may own fewer real windows)
    lo: int      # first global window index of this rank
    hi: int      # one past the last REAL window of this rank
    group: Optional[object] = None

    @property
    def base(self) -> int:
        return self.lo

    def all_gather(self, local: torch.Tensor) -> torch.Tensor:
        """local [chunk, ...] -> [world * chunk, ...] with global window w at row w (identity when world == 1)."""
        if self.world == 1:
            return local
        out = torch.empty((self.world * self.chunk,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out


def partition(num_win: int, world: int, rank: int, group=None) -> WindowShard:
    """Contiguous equal chunks of ceil(num_win / world) windows: rank r owns [r*chunk, min((r+1)*chunk, num_win))."""
    chunk = (num_win + world - 1) // world
    lo = min(rank * chunk, num_win)
    hi = min(lo + chunk, num_win)
    return WindowShard(num_win, world, rank, chunk, lo, hi, group)


def window_shard(num_win: int) -> WindowShard:
    if not _ENABLED:
        return WindowShard(num_win, 1, 0, num_win, 0, num_win, None)
    world = dist.get_world_size(_GROUP)
    rank = dist.get_rank(_GROUP)
    return partition(num_win, world, rank, _GROUP)
