"""-m gpu: end-to-end parity of the product path on the MI355X against (a) the golden outputs of the real
reference (tests/golden) and (b) the CPU oracle on bench-shaped windows, plus size-independent properties at the
full BASELINE.json size."""
import numpy as np
import os

import pytest
import torch

import e2e_cases as ec
import oracle
from oracle import sliding_window as osw

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_blend_only_bitwise_vs_reference():
    ec.case_blend_only_vs_golden(DEV)


def test_net_single_window_vs_reference():
    print(ec.case_net_single_window_vs_golden(DEV))


def test_sliding_window_net5_vs_reference():
    print(ec.case_sliding_window_net5_vs_golden(DEV))


def test_config0_vs_reference():
    """BASELINE.json configs[0]"""
    print(ec.case_config0_vs_golden(DEV))


def test_bench_shaped_windows_vs_oracle():
    """96^3 windows (the tile configurations the 512^3 bench uses), 144x96x96 volume = 2 windows, vs the CPU oracle."""
    from monai_amd.inferers import SlidingWindowInferer

    net, sd = ec.make_net(1, 1, 5, DEV)
    torch.manual_seed(5)
    x = torch.rand(1, 1, 144, 96, 96)
    y = SlidingWindowInferer(roi_size=(96, 96, 96), sw_batch_size=4, overlap=0.5, mode="gaussian")(x.to(DEV), net)
    with torch.no_grad():
        ref = osw.sliding_window_inference(x, (96, 96, 96), 4, lambda w: oracle.basic_unet_forward(sd, w), overlap=0.5, mode="gaussian")
    r = ec.report(y.cpu(), ref)
    print(r)
    assert r["max_abs"] < ec.LOGIT_TOL, r
    assert r["argmax_mismatch"] == 0 or r["max_margin_at_mismatch"] < 2 * ec.LOGIT_TOL, r


def test_headline_many_windows_vs_oracle_192():
    """The COMPLETE inferer on a volume with 27 windows of 96^3 (3 per axis, overlap 0.5, gaussian blend) -- the headline
    configuration's window size, tile configurations and blend pattern, on the reference's synthetic CT volume
    (create_test_image_3d, oracle/synthetic.py) -- product path vs the CPU oracle, with the parity rule of oracle/parity.py:
    max |dlogit| <= 1e-4 and every argmax mismatch inside the oracle's own top-2 margin < 2 max|dlogit|."""
    from monai_amd.inferers import SlidingWindowInferer
    from oracle import synthetic

    net, sd = ec.make_net(1, 1, 5, DEV)
    x = torch.from_numpy(synthetic.benchmark_volume(192))[None, None]
    y = SlidingWindowInferer(roi_size=(96, 96, 96), sw_batch_size=4, overlap=0.5, mode="gaussian", sigma_scale=0.125)(x.to(DEV), net)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        ref = osw.sliding_window_inference(x, (96, 96, 96), 4, lambda w: oracle.basic_unet_forward(sd, w), overlap=0.5, mode="gaussian")
    rep = oracle.assert_label_parity(y, ref, tol=ec.LOGIT_TOL, what="192^3 / 27 windows of 96^3")
    print(rep)


def test_headline_125_windows_vs_oracle_288_both_arithmetics():
    """the COMPLETE inferer on 288^3 = 125 windows of 96^3 (5 per axis: interior voxels covered by 8 windows, every tile
    configuration and blend pattern of the 512^3 headline) against the CPU oracle, for BOTH arithmetic families -- the default (split-precision
    convolutions scaled by their records' magnitude bounds) and the exact-fp32 kernels -- with the parity rule of oracle/parity.py."""
    from monai_amd import config
    from monai_amd.inferers import SlidingWindowInferer
    from oracle import synthetic

    net, sd = ec.make_net(1, 1, 5, DEV)
    x = torch.from_numpy(synthetic.benchmark_volume(288))[None, None]
    inf = SlidingWindowInferer(roi_size=(96, 96, 96), sw_batch_size=4, overlap=0.5, mode="gaussian", sigma_scale=0.125)
    got = {}
    saved = config.CONV_ALGO
    try:
        for algo in ("auto", "fp32"):
            config.CONV_ALGO = algo
            got[algo] = inf(x.to(DEV), net).cpu()
    finally:
        config.CONV_ALGO = saved
    torch.set_num_threads(min(32, torch.get_num_threads()))
    with torch.no_grad():
        ref = osw.sliding_window_inference(x, (96, 96, 96), 4, lambda w: oracle.basic_unet_forward(sd, w), overlap=0.5, mode="gaussian")
    for algo in ("auto", "fp32"):
        print(algo, oracle.assert_label_parity(got[algo], ref, tol=ec.LOGIT_TOL, what=f"288^3 / 125 windows of 96^3 / {algo}"))


def test_mosaic_layout_equals_window_major():
    ec.case_mosaic_layout_equals_window_major(DEV)


def test_nets_with_trained_like_affine_spreads():
    """gamma in +-[1e-3, 1e3], |beta| to ~1e3 (activations far beyond fp16's 65504 inside the nets), one raw-CT-valued window: the default path vs the
    CPU oracle at 1e-4 of the logit scale, and never worse than 4x the exact-fp32 kernels' own distance from it"""
    print(ec.case_nets_with_spread_affine(DEV, window=(64, 64, 64)))
    print(ec.case_nets_with_spread_affine(DEV, window=(96, 96, 96), nets=("basic_unet",)))


def test_net_nonfinite_inputs_like_the_reference():
    ec.case_net_nonfinite_inputs(DEV, window=(64, 64, 64))


def test_slice_inferer_and_adapt_on_device():
    """SURVEY 8 row a9 on the MI355X: SliceInferer drives a 2-D predictor over every slice of a 3-D volume (one window per slice --
    more than 160 windows on the slice axis), SlidingWindowInfererAdapt falls through to the plain inferer when nothing overflows."""
    from monai_amd.inferers import SliceInferer, SlidingWindowInfererAdapt

    torch.manual_seed(11)
    conv = torch.nn.Conv2d(1, 3, 3, padding=1).eval().to(DEV)
    x = torch.rand(1, 1, 200, 40, 48, device=DEV)
    with torch.no_grad():
        for sd, roi in ((0, (40, 48)), (1, (200, 48)), (2, (200, 40))):
            r = SliceInferer(roi_size=roi, spatial_dim=sd, sw_batch_size=16)(x, conv)
            exp = torch.stack([conv(x.select(sd + 2, i)) for i in range(x.shape[sd + 2])], dim=sd + 2)
            assert r.shape == exp.shape and float((r - exp).abs().max()) < 1e-5, (sd, float((r - exp).abs().max()))
        # overlapping 2-D windows inside each slice: equals the reference blend of the same per-window predictions (CPU oracle)
        r = SliceInferer(roi_size=(24, 32), spatial_dim=0, sw_batch_size=8, overlap=0.5, mode="gaussian")(x[:, :, :6], conv)
        cpu_conv = torch.nn.Conv2d(1, 3, 3, padding=1).eval()
        cpu_conv.load_state_dict({k: v.cpu() for k, v in conv.state_dict().items()})
        ref = osw.sliding_window_inference(x[:, :, :6].cpu(), (1, 24, 32), 8, lambda w: cpu_conv(w.squeeze(2)).unsqueeze(2), overlap=0.5, mode="gaussian")
        assert float((r.cpu() - ref).abs().max()) < 1e-5
        a = SlidingWindowInfererAdapt((16, 16, 16), 4, overlap=0.25)(x[:, :, :32], lambda w: w * 2.0 + 1.0)
        assert float((a - (x[:, :, :32] * 2.0 + 1.0)).abs().max()) < 1e-6 and a.is_cuda


def test_fused_argmax_epilogue():
    assert ec.case_fused_argmax_epilogue(DEV)


def test_fused_and_separate_instnorm_statistics_agree():
    net, _ = ec.make_net(1, 1, 5, DEV)
    torch.manual_seed(6)
    x = torch.rand(2, 1, 32, 48, 64, device=DEV)
    a = net(x).clone()
    net.fused_stats = False
    b = net(x)
    assert (a - b).abs().max().item() < 2e-5


def test_batch_independence_and_determinism():
    """InstanceNorm is per sample: the engine's larger window batches must not change any window's logits."""
    net, _ = ec.make_net(1, 1, 5, DEV)
    torch.manual_seed(7)
    x = torch.rand(5, 1, 32, 32, 32, device=DEV)
    full = net(x).clone()
    for i in range(5):
        assert torch.equal(net(x[i : i + 1]), full[i : i + 1])
    assert torch.equal(net(x), full)


def test_full_size_properties_512():
    """BASELINE.json configs[1] size: 512^3, 96^3 windows, overlap 0.5 (1000 windows).  Size-independent checks:
    (1) a predictor that returns a constant per class gives back exactly that constant everywhere (partition of
    unity of the normalised blend, both modes); (2) the K-channel blend of an affine function of the window data is
    that function of the volume up to rounding; (3) the fused network path is finite and matches itself across two runs."""
    from monai_amd.inferers import SlidingWindowInferer, sliding_window_inference

    torch.manual_seed(0)
    vol = torch.rand(1, 1, 512, 512, 512, device=DEV)

    def affine_pred(w):
        return torch.cat([w * (k + 1.0) - 0.25 * k for k in range(5)], dim=1)

    y = sliding_window_inference(vol, (96, 96, 96), 20, affine_pred, overlap=0.5, mode="gaussian")
    for k in range(5):
        assert (y[:, k : k + 1] - (vol * (k + 1.0) - 0.25 * k)).abs().max().item() < 2e-6 * (k + 1)
    del y

    def const_pred(w):
        return torch.stack([torch.full_like(w[:, 0], 0.5 + k) for k in range(3)], dim=1)

    y = sliding_window_inference(vol, (96, 96, 96), 20, const_pred, overlap=0.5, mode="constant")
    for k in range(3):
        assert (y[:, k] - (0.5 + k)).abs().max().item() < 1e-6
    del y

    net, _ = ec.make_net(1, 1, 5, DEV)
    inf = SlidingWindowInferer(roi_size=(96, 96, 96), sw_batch_size=4, overlap=0.5, mode="gaussian")
    a = inf(vol, net)
    assert a.shape == (1, 5, 512, 512, 512) and bool(torch.isfinite(a).all())
    s1 = a.double().sum().item()
    lab = a.argmax(1)
    del a
    b = inf(vol, net)
    s2 = b.double().sum().item()
    # deterministic: no floating-point atomics anywhere
    assert s2 == s1 and torch.equal(b.argmax(1), lab), f"two runs differ: sums {s1!r} vs {s2!r}, {int((b.argmax(1) != lab).sum())} labels"


def test_buffered_schedule_bitwise_vs_reference():
    """`buffer_steps` / `buffer_dim` -- the summation order of the reference's buffered schedule, bit for bit (tests/golden/buffered.npz)"""
    assert ec.case_buffered_blend_vs_golden(DEV) >= 8


def test_conv_engine_splits_couts_16_mod_32():
    print(ec.case_conv_cout_16_mod_32_split(DEV))
    print(ec.case_conv_cout_16_mod_32_split(DEV, cin=48, cout=48, dims=(24, 32, 32), n=2))


def test_unetr_small_vs_reference():
    print(ec.case_unetr_small_vs_golden(DEV))


def test_unetr_vitb_window_vs_reference():
    """BASELINE.json configs[3]: ViT-B/16 UNETR on a 96^3 window -- exercises the MFMA attention kernel at S = 216."""
    print(ec.case_unetr_vitb_vs_golden(DEV))


def test_unetr_sliding_window_192():
    """UNETR through the SlidingWindowInferer (8 windows of 96^3 over a 144^3 volume) vs the CPU oracle."""
    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.networks.nets import UNETR
    from oracle import unetr as ou

    torch.manual_seed(1)
    net = UNETR(in_channels=1, out_channels=5, img_size=(96, 96, 96)).eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(DEV)
    torch.manual_seed(33)
    x = torch.rand(1, 1, 144, 96, 96)
    y = SlidingWindowInferer(roi_size=(96, 96, 96), sw_batch_size=4, overlap=0.5, mode="gaussian")(x.to(DEV), net)
    with torch.no_grad():
        ref = osw.sliding_window_inference(x, (96, 96, 96), 4, lambda w: ou.unetr_forward(sd, w), overlap=0.5, mode="gaussian")
    r = ec.report(y.cpu(), ref)
    print(r)
    assert r["max_abs"] < ec.LOGIT_TOL, r


def test_unet_vs_reference():
    print(ec.case_unet_vs_golden(DEV, names=("res2", "plain", "mixed")))


def test_unet_sliding_window_vs_oracle():
    """MONAI UNet (16..256, 2 residual units) through the SlidingWindowInferer: 8 windows of 96^3 vs the CPU oracle."""
    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.networks.nets import UNet
    from oracle import unet as ou

    ch, st = (16, 32, 64, 128, 256), (2, 2, 2, 2)
    torch.manual_seed(7)
    net = UNet(spatial_dims=3, in_channels=1, out_channels=5, channels=ch, strides=st, num_res_units=2).eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(DEV)
    torch.manual_seed(34)
    x = torch.rand(1, 1, 144, 96, 144)
    y = SlidingWindowInferer(roi_size=(96, 96, 96), sw_batch_size=4, overlap=0.5, mode="gaussian")(x.to(DEV), net)
    with torch.no_grad():
        ref = osw.sliding_window_inference(x, (96, 96, 96), 4, lambda w: ou.unet_forward(sd, w, ch, st, 2), overlap=0.5, mode="gaussian")
    r = ec.report(y.cpu(), ref)
    print(r)
    assert r["max_abs"] < ec.LOGIT_TOL, r


def test_basic_unet_odd_window_vs_reference():
    print(ec.case_net_odd_window_vs_golden(DEV))


def test_process_fn_bitwise_vs_reference():
    ec.case_process_fn_vs_golden(DEV)


def test_slabwise_equals_whole():
    print(ec.case_slabwise_equals_whole(DEV))


def test_patch_inferer_vs_reference():
    import patch_cases as pc

    print("cases", pc.case_patch_inferer_vs_reference(DEV))
    pc.case_patch_inferer_api(DEV)
    pc.case_gathered_split_and_batched_merge(DEV)


def test_bundle_shaped_pipeline_vs_reference():
    import pipeline_case as pl

    print(pl.case_pipeline_vs_reference(DEV))


def test_narrow_and_host_inputs():
    ec.case_narrow_and_host_inputs(DEV)


def test_basic_unet_2d_and_slice_inferer_vs_reference():
    """BasicUNet(spatial_dims=2) on the one-plane engine and SliceInferer over it, against the real reference"""
    print("max |dlogit|", ec.case_basic_unet_2d_vs_reference(DEV))


def test_upcat_fused_vs_two_layers_and_reference():
    """UpCat without its up-sampled intermediate (kernels/upconv_h2.h) inside BasicUNet: golden logits of the real reference + the engine's two-layer path"""
    print(ec.case_net_upcat_fused_vs_two_layers(DEV))


def test_basic_unet_pixelshuffle_vs_reference():
    """BasicUNet(upsample="pixelshuffle") on the HIP path (sub-pixel convolution + pixelshuffle_kernel) against the real reference's golden logits"""
    print(ec.case_basic_unet_pixelshuffle_vs_golden(DEV))


def test_conv_halves_vs_one_launch_and_reference():
    """UpCat's convolution over a 64-channel concatenation as two 32-channel launches of the Winograd split-precision kernel (BasicUNet._conv_halves): golden logits of the
    real reference + the engine's one-launch path"""
    print(ec.case_net_conv_halves_vs_one_launch(DEV))


def test_buffered_schedule_with_callbacks_bitwise_vs_reference():
    """SURVEY 8a row a7 with the rest of its call surface: process_fn / with_coord / tuple and dict outputs under buffer_steps"""
    assert ec.case_buffered_calls_vs_golden(DEV) == 6


def test_pooling_epilogue_leaves_the_logits_bitwise():
    assert ec.case_net_pool_fused_bitwise(DEV)


def test_swin_attention_from_table_and_regions_bitwise():
    """round 5: window attention with bias / mask evaluated in the kernel == the S x S table form through the whole SwinUNETR, bit for bit"""
    import swin_cases as sc

    print(sc.case_swin_rel_attention_bitwise(DEV))


def test_swin_block_moves_folded_into_kernels_bitwise():
    import swin_cases as sc

    assert sc.case_swin_fused_moves_bitwise(DEV)


def test_rccl_one_rank_forced_sharding_bitwise_192():
    """The N > 1 code on the hardware it was written for, with the one GPU a box has: backend "nccl" (= RCCL) initialised with ONE rank,
    `parallel.window_sharding(force=True)` sends the 192^3 headline case (27 windows of 96^3) through the round schedule (a main round and the short tail rounds: variable
    slot sizes), the padded window-major row buffer, the in-place all-gather probe, one asynchronous `all_gather_into_tensor` per round and `_Pending.wait` -- the result
    must equal the unsharded one bit for bit, also with 8 windows per launch (several main rounds + a tail) and with the out-of-place gather form."""
    import socket

    import torch.distributed as dist

    from monai_amd import parallel
    from monai_amd.inferers import SlidingWindowInferer
    from oracle import synthetic

    net, _ = ec.make_net(1, 1, 5, DEV)
    x = torch.from_numpy(synthetic.benchmark_volume(192))[None, None].to(DEV)
    inf = SlidingWindowInferer(roi_size=(96, 96, 96), sw_batch_size=4, overlap=0.5, mode="gaussian")
    single = inf(x, net).clone()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    issued = []
    real = parallel.WindowShard.gather_round

    def counting(self, full, q, nb):
        work = real(self, full, q, nb)
        issued.append((q, nb, work is not None and work.keep is None))
        return work

    parallel.WindowShard.gather_round = counting
    try:
        assert dist.get_backend() == "nccl"
        with parallel.window_sharding(force=True):
            forced = inf(x, net).clone()
            n_default = len(issued)
            os.environ["MONAI_AMD_SW_BATCH"] = "8"          # 27 windows: main rounds of 8, then tail rounds of 2
            try:
                assert [n for _, n in parallel.window_shard(27).schedule(8)] == [8, 8, 2, 2, 2, 2, 2, 2]
                forced8 = inf(x, net).clone()
            finally:
                del os.environ["MONAI_AMD_SW_BATCH"]
            verdicts = parallel.inplace_gather_verdicts()
            inplace_rounds = [i for i in issued if i[2]]
            os.environ["MONAI_AMD_GATHER_INPLACE"] = "0"
            try:
                staged = inf(x, net).clone()
            finally:
                del os.environ["MONAI_AMD_GATHER_INPLACE"]
        torch.cuda.synchronize()
        assert n_default >= 1 and len(issued) >= n_default + 8, issued
        assert len(verdicts) >= 1, "the in-place probe must have reached a verdict on RCCL"
        print({"inplace_gather_ok": list(verdicts.values()), "rounds_issued": len(issued), "in_place": len(inplace_rounds)})
        assert torch.equal(forced, single), "RCCL one-rank sharded result differs from the unsharded one"
        assert torch.equal(forced8, single) and torch.equal(staged, single)
        assert not parallel._ENABLED and not parallel._FORCE
    finally:
        parallel.WindowShard.gather_round = real
        dist.destroy_process_group()
