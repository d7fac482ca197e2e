"""N > 1 path on the CPU: two ranks (gloo, 127.0.0.1) run the REAL product path -- window sharding, all-gather of
the per-window logits, replicated deterministic blend -- with the kernels provided by the SIMT-emulator build.
Every rank must reproduce the single-process result bit for bit (same kernels, same summation order), and that
result must match the CPU oracle within the 1e-4 logit tolerance."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


# an eighth of the default widths: the test is about sharding / gathering / blending, not about the convolutions (which other tests
# cover at full width), and the SIMT emulator pays for every flop
FEATURES = (8, 8, 16, 32, 64, 8)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, roi, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        here = os.path.dirname(os.path.abspath(__file__))
        sys.path[:0] = [here, os.path.dirname(here)]
        from emu_backend import emu_backend

        from monai_amd import parallel
        from monai_amd.inferers import SlidingWindowInferer
        from monai_amd.networks.nets import BasicUNet

        with emu_backend():
            torch.manual_seed(1)
            net = BasicUNet(3, 1, 3, features=FEATURES).eval()
            torch.manual_seed(3)
            x = torch.rand(shape)
            inf = SlidingWindowInferer(roi_size=roi, sw_batch_size=2, overlap=0.5, mode="gaussian")
            single = inf(x, net).clone()
            parallel.enable_window_sharding()
            sharded = inf(x, net).clone()
            os.environ["MONAI_AMD_SW_BATCH"] = "2"      # 2 windows per launch -> 2 rounds of 4, the last one half padding
            sharded2 = inf(x, net).clone()
            del os.environ["MONAI_AMD_SW_BATCH"]
            assert torch.equal(sharded, sharded2), "the result must not depend on the round size"
            # the all-gather of a round runs in place on the rank's own rows when a probe collective says the backend handles aliased
            # buffers (parallel._inplace_gather_ok), out of place through a private send copy otherwise: same bits either way
            assert parallel._inplace_gather_ok(None, x.device) in (True, False)
            os.environ["MONAI_AMD_GATHER_INPLACE"] = "0"
            staged = inf(x, net).clone()
            del os.environ["MONAI_AMD_GATHER_INPLACE"]
            assert torch.equal(staged, sharded), "in-place and out-of-place round gathers must agree"
            # volumes whose logits exceed the budget go slab by slab -- also under sharding (every slab's windows are sharded, the
            # fit decision is collective): force it with a cap of two window rows and compare with the unsharded result
            per_win = 3 * roi[0] * roi[1] * roi[2] * 4
            os.environ["MONAI_AMD_MAX_LOGITS_BYTES"] = str(2 * 2 * per_win + 64)
            slabbed = inf(x, net).clone()
            del os.environ["MONAI_AMD_MAX_LOGITS_BYTES"]
            assert torch.equal(slabbed, sharded), "slab-wise processing under window sharding must not change a bit"
            parallel.disable_window_sharding()
            shard = parallel.partition(7, world, rank)
        ret[rank] = (single, sharded, (shard.lo, shard.hi, shard.chunk))
    finally:
        dist.destroy_process_group()


def test_two_rank_window_sharding_matches_single_process():
    import oracle
    from oracle import sliding_window as osw

    shape, roi = (1, 1, 64, 24, 16), (32, 16, 16)   # 3 x 2 x 1 = 6 windows -> 3 per rank
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), shape, roi, ret), nprocs=2, join=True)
    (s0, d0, p0), (s1, d1, p1) = ret[0], ret[1]
    assert torch.equal(s0, s1)                      # both ranks see the same single-process result
    assert torch.equal(d0, s0) and torch.equal(d1, s0)  # sharded == unsharded, bit for bit, on every rank
    assert p0 == (0, 4, 4) and p1 == (4, 7, 4)      # uneven split: equal chunks, last rank owns fewer real windows

    torch.manual_seed(1)
    sd = oracle.make_basic_unet_state(1, 3, features=FEATURES)
    torch.manual_seed(3)
    x = torch.rand(shape)
    with torch.no_grad():
        ref = osw.sliding_window_inference(x, roi, 2, lambda w: oracle.basic_unet_forward(sd, w), overlap=0.5, mode="gaussian")
    assert (d0 - ref).abs().max().item() < 1e-4


def _worker_n(rank, world, port, shape, roi, ret):
    """world sizes 3 and 4: a padded last round, ranks that propose DIFFERENT windows-per-launch (the minimum must win on every rank,
    or the slot arithmetic dead-locks / mis-places rows), the slab-wise path under sharding -- each against the single-process result, bitwise"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        here = os.path.dirname(os.path.abspath(__file__))
        sys.path[:0] = [here, os.path.dirname(here)]
        from emu_backend import emu_backend

        from monai_amd import parallel
        from monai_amd.inferers import SlidingWindowInferer
        from monai_amd.networks.nets import BasicUNet

        with emu_backend():
            torch.manual_seed(1)
            net = BasicUNet(3, 1, 3, features=FEATURES).eval()
            torch.manual_seed(3)
            x = torch.rand(shape)
            inf = SlidingWindowInferer(roi_size=roi, sw_batch_size=2, overlap=0.5, mode="gaussian")
            single = inf(x, net).clone()
            parallel.enable_window_sharding()
            res = {}
            os.environ["MONAI_AMD_SW_BATCH"] = "1"                 # 10 windows, rounds of `world`: the last round is padded (3: 12 rows, 4: 12 rows)
            res["nb1"] = inf(x, net).clone()
            os.environ["MONAI_AMD_SW_BATCH"] = str(2 + rank)       # every rank proposes another batch: the agreed one is the minimum (2)
            res["disagree"] = inf(x, net).clone()
            os.environ["MONAI_AMD_SW_BATCH"] = "2"
            per_win = 3 * roi[0] * roi[1] * roi[2] * 4
            # 8 windows' worth of logits: the 10 windows (padded to whole rounds of world x 2 = 12 rows) do not fit -> slab by slab, every slab's windows sharded
            os.environ["MONAI_AMD_MAX_LOGITS_BYTES"] = str(8 * per_win + 64)
            res["slabs"] = inf(x, net).clone()
            del os.environ["MONAI_AMD_MAX_LOGITS_BYTES"], os.environ["MONAI_AMD_SW_BATCH"]
            parallel.disable_window_sharding()
        ret[rank] = (single, res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [3, pytest.param(4, marks=pytest.mark.heavy_emu)])      # world 4: MONAI_AMD_HEAVY_EMU=1 (four emulator processes; bench.py --gpus 4 runs in test_bench_harness.py)
def test_window_sharding_world_sizes_3_and_4(world):
    shape, roi = (1, 1, 96, 24, 16), (32, 16, 16)    # 5 x 2 x 1 = 10 windows
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_n, args=(world, _free_port(), shape, roi, ret), nprocs=world, join=True)
    single = ret[0][0]
    for r in range(world):
        s, res = ret[r]
        assert torch.equal(s, single)
        for name, t in res.items():
            assert torch.equal(t, single), f"rank {r} of {world}: {name} differs from the single-process result"


def _worker_1000(rank, world, port, ret):
    """the headline's window count under sharding: 10 x 10 x 10 = 1000 windows (8^3 at overlap 0.5 over 44^3), a toy per-window predictor through
    the generic path (sw_batch_size 60 -> the schedule of the 8-GPU run: one main round of 8 x 60, five tail rounds of 8 x 13, 40 padded rows), every rank's result
    against the single-process one, bitwise"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        here = os.path.dirname(os.path.abspath(__file__))
        sys.path[:0] = [here, os.path.dirname(here)]
        from emu_backend import emu_backend

        from monai_amd import parallel
        from monai_amd.inferers import sliding_window_inference

        calls = []

        def toy(w):      # two classes, per-window statistics inside (a window's logits depend on nothing but the window)
            calls.append(int(w.shape[0]))
            m = w.mean(dim=(2, 3, 4), keepdim=True)
            return torch.cat([w * 1.5 - m, (w - m) * (w - m) + 0.25], dim=1)

        with emu_backend():
            torch.manual_seed(21)
            x = torch.rand(1, 1, 44, 44, 44)
            single = sliding_window_inference(x, (8, 8, 8), 60, toy, overlap=0.5, mode="gaussian").clone()
            n_single = list(calls)
            del calls[:]
            parallel.enable_window_sharding()
            sharded = sliding_window_inference(x, (8, 8, 8), 60, toy, overlap=0.5, mode="gaussian").clone()
            parallel.disable_window_sharding()
        ret[rank] = (single, sharded, n_single, list(calls))
    finally:
        dist.destroy_process_group()


def test_world_size_8_with_1000_windows_bitwise():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_1000, args=(8, _free_port(), ret), nprocs=8, join=True)
    single = ret[0][0]
    assert sum(ret[0][2]) == 1000
    for r in range(8):
        s, d, _, calls = ret[r]
        assert torch.equal(s, single) and torch.equal(d, single), f"rank {r}: sharded != single process"
        assert calls[0] == 60 and len(calls) <= 6 and sum(calls) in (125, 124, 126, 120, 128)      # this rank's share of the 1000 windows in <= 6 launches
    assert sum(sum(ret[r][3]) for r in range(8)) == 1000


def test_partition_covers_every_window_once():
    from monai_amd import parallel

    for num_win in (1, 5, 8, 1000, 1001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                p = parallel.partition(num_win, world, r)
                assert p.hi - p.lo <= p.chunk and p.chunk * world >= num_win
                seen += list(range(p.lo, p.hi))
            assert seen == list(range(num_win))


def test_round_schedule_covers_every_window_once():
    """The round-interleaved schedule (monai_amd/parallel.py: WindowShard.schedule): main rounds of world x nb windows, shorter tail rounds at the end; every real window
    is computed by exactly one rank, each rank's slot of a round starting at row `base` is [base + rank * n, +n), rounds are contiguous, the padded row count is
    where the last round ends, the tail keeps between half a round and one and a half, and the busiest rank is within a launch of the ideal share."""
    from monai_amd import parallel

    for num_win in (1, 7, 200, 1000, 1001):
        for world in (1, 2, 3, 8):
            for nb in (1, 4, 32, 60):
                seen = []
                probe = parallel.partition(num_win, world, 0)
                sched = probe.schedule(nb)
                padded = probe.padded_windows(nb)
                assert sched[0][0] == 0 and all(sched[q + 1][0] == sched[q][0] + (world * sched[q][1] if world > 1 else sched[q][1]) for q in range(len(sched) - 1))
                assert padded >= num_win and sched[-1][0] < num_win and all(1 <= n <= nb for _, n in sched)
                if world > 1:
                    assert padded == sched[-1][0] + world * sched[-1][1]
                    tail = [n for _, n in sched if n < nb]
                    if nb >= 8 and num_win >= world * nb:      # a real tail: short rounds, at least two of them, after at most ... full ones
                        assert len(tail) >= 2 and max(tail) <= nb // 4 + 1 and world * nb // 2 <= num_win - sum(world * n for _, n in sched if n == nb) < 3 * world * nb // 2 + 1
                for r in range(world):
                    sh = parallel.partition(num_win, world, r)
                    rounds = sh.rounds(nb)
                    assert len(rounds) == len(sched)
                    for (base, per), (w0, n) in zip(sched, rounds):
                        assert w0 == (base + r * per if world > 1 else base) and 0 <= n <= per and w0 + n <= max(num_win, w0)
                        seen += list(range(w0, w0 + n))
                assert sorted(seen) == list(range(num_win))
    # the headline at 8 GPUs: 125 windows per rank, the last round 13 windows per rank (1.6 GB of logits per rank exposed in front of the blend instead of 7.8 GB)
    sh = parallel.partition(1000, 8, 0)
    assert [n for _, n in sh.schedule(60)] == [60, 13, 13, 13, 13, 13] and sum(n for _, n in sh.rounds(60)) == 125
    os.environ["MONAI_AMD_TAIL_ROUNDS"] = "0"
    try:
        assert [n for _, n in sh.schedule(60)] == [60, 60, 60] and sh.padded_windows(60) == 1440
    finally:
        del os.environ["MONAI_AMD_TAIL_ROUNDS"]


def test_scoped_switches_restore_the_previous_state():
    """`parallel.window_sharding(group)` and `config.conv_algo_scope(name)`: scoped forms of the two process-wide host switches -- nested scopes restore the outer
    state, an exception inside the body restores it too, an unknown family is refused before anything changes."""
    from monai_amd import config, parallel

    saved = config.CONV_ALGO
    try:
        config.CONV_ALGO = "auto"
        with config.conv_algo_scope("fp32"):
            assert config.conv_algo() == config.CONV_ALGOS["fp32"]
            with config.conv_algo_scope("h2"):
                assert config.conv_algo() == config.CONV_ALGOS["h2"]
            assert config.CONV_ALGO == "fp32"
        assert config.CONV_ALGO == "auto"
        with pytest.raises(RuntimeError):
            with config.conv_algo_scope("direct"):
                raise RuntimeError("body failed")
        assert config.CONV_ALGO == "auto"
        with pytest.raises(ValueError):
            config.conv_algo_scope("no-such-family")
    finally:
        config.CONV_ALGO = saved

    assert not parallel._ENABLED
    with pytest.raises(RuntimeError):          # no process group: refused, and nothing is left switched on
        with parallel.window_sharding():
            pass
    assert not parallel._ENABLED and parallel._GROUP is None
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        with parallel.window_sharding():
            assert parallel._ENABLED and parallel._GROUP is None
            sub = dist.new_group([0])
            with parallel.window_sharding(sub):
                assert parallel._GROUP is sub
            assert parallel._ENABLED and parallel._GROUP is None
        assert not parallel._ENABLED
    finally:
        dist.destroy_process_group()


def _worker_forced(rank, world, port, ret):
    """ONE rank with the sharding forced (parallel.window_sharding(force=True)): the round schedule with its tail, the padded row buffer, the in-place probe and one
    asynchronous all-gather per round run although there is nobody to exchange with -- the form the RCCL path is exercised in on a one-GPU box"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        here = os.path.dirname(os.path.abspath(__file__))
        sys.path[:0] = [here, os.path.dirname(here)]
        from emu_backend import emu_backend

        from monai_amd import parallel
        from monai_amd.inferers import SlidingWindowInferer, sliding_window_inference
        from monai_amd.networks.nets import BasicUNet

        gathers = []
        real = parallel.WindowShard.gather_round

        def counting(self, full, q, nb):
            work = real(self, full, q, nb)
            gathers.append((q, work is not None))
            return work

        parallel.WindowShard.gather_round = counting

        def toy(w):
            m = w.mean(dim=(2, 3, 4), keepdim=True)
            return torch.cat([w * 1.5 - m, (w - m) * (w - m) + 0.25], dim=1)

        with emu_backend():
            torch.manual_seed(21)
            x = torch.rand(1, 1, 28, 28, 28)               # 6^3 = 216 windows of 8^3 at overlap 0.5
            single = sliding_window_inference(x, (8, 8, 8), 32, toy, overlap=0.5, mode="gaussian").clone()
            assert not gathers
            with parallel.window_sharding(force=True):
                sh = parallel.window_shard(216)
                assert sh.sharded and sh.world == 1 and [n for _, n in sh.schedule(32)] == [32] * 6 + [8, 8, 8]      # main rounds, then the tail (nb / 4)
                forced = sliding_window_inference(x, (8, 8, 8), 32, toy, overlap=0.5, mode="gaussian").clone()
            n_toy = len(gathers)
            verdicts = parallel.inplace_gather_verdicts()
            # the fused engine path (window-major rows instead of the mosaic under sharding)
            torch.manual_seed(1)
            net = BasicUNet(3, 1, 3, features=FEATURES).eval()
            torch.manual_seed(3)
            y = torch.rand(1, 1, 64, 24, 16)
            inf = SlidingWindowInferer(roi_size=(32, 16, 16), sw_batch_size=2, overlap=0.5, mode="gaussian")
            single_net = inf(y, net).clone()
            with parallel.window_sharding(force=True):
                forced_net = inf(y, net).clone()
            assert not parallel._ENABLED and not parallel._FORCE
        ret[rank] = (torch.equal(single, forced), torch.equal(single_net, forced_net), n_toy, len(gathers), all(g[1] for g in gathers), len(verdicts))
    finally:
        dist.destroy_process_group()


def test_one_rank_forced_sharding_runs_the_collective_path_bitwise():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_forced, args=(1, _free_port(), ret), nprocs=1, join=True)
    same_toy, same_net, n_toy, n_all, all_async, n_verdicts = ret[0]
    assert same_toy and same_net, "forced one-rank sharding must not change a bit"
    assert n_toy == 9 and n_all > n_toy and all_async, "every round must have issued its all-gather"
    assert n_verdicts >= 1, "the in-place probe must have reached a verdict"
