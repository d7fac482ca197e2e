"""End-to-end parity cases (inferer + BasicUNet engine), shared by the emulator (CPU) and GPU test modules."""
from __future__ import annotations

import os

import numpy as np
import torch

import oracle
from oracle import sliding_window as osw

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOGIT_TOL = 1e-4  # BASELINE.json north_star: fp32 logits within 1e-4 of the reference CPU inferer


def make_net(seed, in_ch, out_ch, device, features=(32, 32, 64, 128, 256, 32)):
    from monai_amd.networks.nets import BasicUNet

    torch.manual_seed(seed)
    net = BasicUNet(3, in_ch, out_ch, features=features).eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    return net.to(device), sd


def report(got: torch.Tensor, exp: torch.Tensor):
    """max |logit diff|, argmax agreement and the top-2 margin statistics the north_star asks for."""
    d = (got.double() - exp.double()).abs()
    top2 = exp.double().topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])
    mism = (got.argmax(1) != exp.argmax(1))
    return dict(max_abs=d.max().item(), mean_abs=d.mean().item(), argmax_mismatch=int(mism.sum().item()),
                min_margin=margin.min().item(), n_margin_lt_1e4=int((margin < 1e-4).sum().item()),
                max_margin_at_mismatch=(margin[mism].max().item() if mism.any() else 0.0))


def case_net_single_window_vs_golden(device, second_window=True):
    """5-class bench weights (seed 1) on the golden windows produced by the REAL reference."""
    g = np.load(os.path.join(GOLDEN, "net5.npz"))
    net, sd = make_net(1, 1, 5, device)
    import hashlib

    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode()); h.update(v.contiguous().numpy().tobytes())
    assert h.hexdigest() == str(g["net5_state_sha256"]), "same seed must give the reference's weights"
    torch.manual_seed(21)
    x = torch.rand(2, 1, 32, 32, 32)
    got = net(x.to(device)).cpu()
    r = report(got, torch.from_numpy(g["net5_win32_out"]))
    assert r["max_abs"] < LOGIT_TOL, r
    if not second_window:
        return r, None
    torch.manual_seed(22)
    x = torch.rand(1, 1, 48, 32, 16)
    got = net(x.to(device)).cpu()
    r2 = report(got, torch.from_numpy(g["net5_win48x32x16_out"]))
    assert r2["max_abs"] < LOGIT_TOL, r2
    return r, r2


def case_net_upcat_fused_vs_two_layers(device):
    """BasicUNet's top decoder level (a 32-channel k2 s2 transposed convolution feeding conv3) as the composite transposed convolution of the low-resolution tensor
    (csrc/kernels/upconv_h2.h, config.UPCAT_FUSED): against the REAL reference's golden logits (net5.npz: the same 1e-4 bar as the two-layer path) and against
    the two-layer evaluation of this engine (a few 1e-6: one more fp32 rounding per composite weight, another summation order)."""
    from monai_amd import config

    g = np.load(os.path.join(GOLDEN, "net5.npz"))
    net, _ = make_net(1, 1, 5, device)
    torch.manual_seed(21)
    x = torch.rand(2, 1, 32, 32, 32).to(device)
    saved, saved_algo = config.UPCAT_FUSED, config.CONV_ALGO
    try:
        config.CONV_ALGO = "auto"               # the composite kernel belongs to the split-precision family: not taken under "fp32"
        config.UPCAT_FUSED = False
        two = net(x).cpu()
        config.UPCAT_FUSED = True
        assert net._plans and next(iter(net._plans.values()))._fusable(net, 0, torch.empty(1, 32, 1, 1, 1), 32)
        fused = net(x).cpu()
        saved_order = config.UPCAT_ORDER
        try:      # the other order of the two halves (composite term written, accumulating convolution after it): the same two addends, statistics from the other kernel's tiles
            config.UPCAT_ORDER = "term_first" if config.upcat_order() == "conv_first" else "conv_first"
            other = net(x).cpu()
        finally:
            config.UPCAT_ORDER = saved_order
        assert (other - fused).abs().max().item() < 2e-5
        config.CONV_ALGO = "fp32"
        assert not next(iter(net._plans.values()))._fusable(net, 0, torch.empty(1, 32, 1, 1, 1), 32)
    finally:
        config.UPCAT_FUSED, config.CONV_ALGO = saved, saved_algo
    r = report(fused, torch.from_numpy(g["net5_win32_out"]))
    assert r["max_abs"] < LOGIT_TOL, r
    d = (fused - two).abs().max().item()
    assert 0.0 < d < 2e-5, d
    return r, d


def case_net_conv_halves_vs_one_launch(device, window=(48, 32, 32)):
    """BasicUNet's 48^3-type decoder level (64-channel concatenation feeding conv3; here a (24, 16, 16) level of a (48, 32, 32) window): the convolution as two 32-channel
    launches of the Winograd split-precision kernel -- skip half written, up-sampled half accumulated with bias and statistics (config.CONV_HALVES,
    BasicUNet._conv_halves) -- against the CPU oracle (bit-pinned to the real reference: tests/test_oracle_golden.py) at the 1e-4 bar and against the one-launch evaluation
    on the direct kernel (a few 1e-6: another summation order)."""
    from monai_amd import config

    net, sd = make_net(1, 1, 5, device)
    gen = torch.Generator().manual_seed(23)
    x = torch.rand((2, 1) + tuple(window), generator=gen)
    with torch.no_grad():
        exp = oracle.basic_unet_forward(sd, x)
    xd = x.to(device)
    saved, saved_algo = config.CONV_HALVES, config.CONV_ALGO
    try:
        config.CONV_ALGO = "auto"
        config.CONV_HALVES = False
        one = net(xd).cpu()
        plan = next(iter(net._plans.values()))
        assert plan._halves_cfg(net, 1, 32, bounded=True) < 0
        config.CONV_HALVES = True
        assert plan._halves_cfg(net, 1, 32, bounded=True) >= 0, "the 64 -> 32 convolution at (24, 16, 16) must take the two-launch path"
        assert plan._halves_cfg(net, 2, 64, bounded=True) < 0, "128-channel concatenations stay on one launch"
        halves = net(xd).cpu()
        config.CONV_ALGO = "fp32"
        assert plan._halves_cfg(net, 1, 32, bounded=True) < 0
    finally:
        config.CONV_HALVES, config.CONV_ALGO = saved, saved_algo
    r = report(halves, exp)
    assert r["max_abs"] < LOGIT_TOL, r
    d = (halves - one).abs().max().item()
    assert 0.0 < d < 2e-5, d
    return r, d


def case_basic_unet_pixelshuffle_vs_golden(device, which=("even", "odd", "2d")):
    """BasicUNet(upsample="pixelshuffle"): UpSample -> SubpixelUpsample (k3 convolution to 8 x the channels, pixel shuffle, zero pad in front + average pooling;
    csrc/kernels/nn_simple.h: pixelshuffle_kernel) with the REAL reference's parameters and logits (tests/golden/make_golden_pixelshuffle.py): strict state_dict load,
    even extents and odd ones (UpCat's replicate padding behind the shuffle)."""
    from monai_amd.networks.nets import BasicUNet

    g = np.load(os.path.join(GOLDEN, "basic_unet_pixelshuffle.npz"))
    net = BasicUNet(3, 1, 3, features=tuple(int(v) for v in g["features"]), upsample="pixelshuffle").eval()
    assert list(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    missing, unexpected = net.load_state_dict({str(k): torch.from_numpy(g["p:" + str(k)]) for k in g["keys"]}, strict=True)
    assert not missing and not unexpected
    net = net.to(device)
    errs = {}
    for name in which:
        if name == "2d":
            continue
        x, exp = torch.from_numpy(g["x_" + name]), torch.from_numpy(g["y_" + name])
        got = net(x.to(device)).cpu()
        errs[name] = (got - exp).abs().max().item()
        assert errs[name] < LOGIT_TOL * max(1.0, exp.abs().max().item()), (name, errs)
    if "2d" in which:      # two spatial dimensions: the one-plane form (4 sub-pixels per channel)
        net2 = BasicUNet(2, 2, 3, features=tuple(int(v) for v in g["features"]), upsample="pixelshuffle").eval()
        assert list(net2.state_dict().keys()) == [str(k) for k in g["keys2"]]
        net2.load_state_dict({str(k): torch.from_numpy(g["p2:" + str(k)]) for k in g["keys2"]}, strict=True)
        exp = torch.from_numpy(g["y_2d"])
        got = net2.to(device)(torch.from_numpy(g["x_2d"]).to(device)).cpu()
        errs["2d"] = (got - exp).abs().max().item()
        assert errs["2d"] < LOGIT_TOL * max(1.0, exp.abs().max().item()), errs
    return errs


def case_net_pool_fused_bitwise(device):
    """BasicUNet with MaxPool3d(2) leaving the producing convolution's epilogue (config.POOL_FUSED; csrc/kernels/conv3d_h2.h, POOL) against the pooling pass: the consumer reads
    raw maxima under the producer's records instead of pooled activated values -- act(max raw) == max(act(raw)) -- so the logits are BITWISE the same; with trained-like
    norms (negative gammas on a third of the channels: those read the raw minima) too"""
    from monai_amd import config

    net, _ = make_net(1, 1, 5, device)
    torch.manual_seed(21)
    x = torch.rand(2, 1, 32, 32, 32).to(device)
    saved, saved_algo = config.POOL_FUSED, config.CONV_ALGO
    try:
        config.CONV_ALGO = "auto"
        for spread in (False, True):
            if spread:
                with torch.no_grad():
                    for name, p_ in net.named_parameters():
                        if name.endswith("adn.N.weight"):
                            p_.mul_(torch.where(torch.arange(p_.numel(), device=p_.device) % 3 == 1, -1.0, 1.0))
            config.POOL_FUSED = False
            two = net(x).clone()
            config.POOL_FUSED = True
            plan = next(iter(net._plans.values()))
            fused = net(x).clone()
            assert plan.pool_min[1] is not None, "the pooling epilogue did not run"
            assert torch.equal(fused, two), (fused - two).abs().max().item()
    finally:
        config.POOL_FUSED, config.CONV_ALGO = saved, saved_algo
    return True


def case_net_odd_window_vs_golden(device):
    """Window extents that are odd at levels 1, 2 and 3: UpCat's replicate padding (basic_unet.py:163-170) vs the reference."""
    g = np.load(os.path.join(GOLDEN, "net5_odd.npz"))
    net, _ = make_net(1, 1, 5, device)
    torch.manual_seed(23)
    x = torch.rand(1, 1, 40, 36, 34)
    got = net(x.to(device)).cpu()
    r = report(got, torch.from_numpy(g["out"]))
    assert r["max_abs"] < LOGIT_TOL, r
    return r


def case_sliding_window_net5_vs_golden(device):
    from monai_amd.inferers import SlidingWindowInferer

    g = np.load(os.path.join(GOLDEN, "net5.npz"))
    net, _ = make_net(1, 1, 5, device)
    torch.manual_seed(23)
    x = torch.rand(1, 1, 48, 40, 32)
    y = SlidingWindowInferer(roi_size=(32, 32, 32), sw_batch_size=4, overlap=0.5, mode="gaussian")(x.to(device), net)
    r = report(y.cpu(), torch.from_numpy(g["net5_sw_out"]))
    assert r["max_abs"] < LOGIT_TOL, r
    assert r["argmax_mismatch"] == 0 or r["max_margin_at_mismatch"] < 2 * LOGIT_TOL, r
    return r


def case_slabwise_equals_whole(device):
    """Volumes whose all-window logits exceed the budget are processed slab by slab along the first spatial axis
    (monai_amd/inferers/utils.py:_slabwise): the result must be BIT-IDENTICAL to the one-pass result -- fused engine and a
    generic predictor (tuple of two outputs, one at half resolution), gaussian and constant blending."""
    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.inferers import utils as U

    net, _ = make_net(1, 1, 5, device)
    torch.manual_seed(31)
    x = torch.rand(1, 1, 40, 24, 16).to(device)          # roi 16, overlap 0.5 -> 4 rows of 2 x 1 windows along the first axis

    def two_heads(w):                                    # generic predictor: full-resolution and half-resolution outputs
        a = torch.cat([w * 2.0, w.flip(2) - 0.5], dim=1)
        return a, torch.nn.functional.avg_pool3d(a, 2)

    calls = []
    real = U._slabwise

    def spy(*a, **k):
        calls.append(1)
        return real(*a, **k)

    res = {}
    for name, pred, mode in (("net", net, "gaussian"), ("generic", two_heads, "constant")):
        inf = SlidingWindowInferer(roi_size=(16, 16, 16), sw_batch_size=2, overlap=0.5, mode=mode)
        whole = inf(x, pred)
        k = 5 if name == "net" else 2
        row_bytes = 2 * k * 16 ** 3 * 4
        os.environ["MONAI_AMD_MAX_LOGITS_BYTES"] = str(2.5 * row_bytes)      # two rows fit, four do not
        U._slabwise = spy
        try:
            n0 = len(calls)
            slabs = inf(x, pred)
            assert len(calls) == n0 + 1, "the slab-wise path did not run"
        finally:
            U._slabwise = real
            del os.environ["MONAI_AMD_MAX_LOGITS_BYTES"]
        for a, b in zip(whole if isinstance(whole, tuple) else (whole,), slabs if isinstance(slabs, tuple) else (slabs,)):
            assert a.shape == b.shape and torch.equal(a, b), (name, float((a - b).abs().max()))
        res[name] = True
    os.environ["MONAI_AMD_MAX_LOGITS_BYTES"] = "1000"    # not even one row fits: a clear error, no silent fallback
    try:
        try:
            SlidingWindowInferer(roi_size=(16, 16, 16), sw_batch_size=2, overlap=0.5)(x, net)
            raise AssertionError("expected the logits budget error")
        except RuntimeError as e:
            assert "logits buffer" in str(e)
    finally:
        del os.environ["MONAI_AMD_MAX_LOGITS_BYTES"]
    return res


def case_fused_argmax_epilogue(device):
    """the argmax of AsDiscrete(argmax=True) fused into the blend epilogue.  Inferer + fused network, generic predictor with
    two outputs (one at half resolution -> scaled window starts), the slab-wise path, uint8 labels: every label map equals
    AsDiscrete(argmax=True) of the unfused inferer output, bit for bit."""
    from monai_amd.inferers import SlidingWindowArgmaxInferer, SlidingWindowInferer, sliding_window_argmax
    from monai_amd.transforms import AsDiscrete

    net, _ = make_net(1, 1, 5, device)
    torch.manual_seed(41)
    nb = 1 if str(device) == "cpu" else 2            # the emulator pays for every window of the full-width net: one volume there, a batch of two on the GPU
    x = torch.rand(2, 1, 40, 24, 16)[:nb].to(device)
    inf = SlidingWindowInferer(roi_size=(16, 16, 16), sw_batch_size=2, overlap=0.5, mode="gaussian")
    ref = inf(x, net)
    exp = torch.stack([AsDiscrete(argmax=True)(ref[i]) for i in range(nb)])
    got = inf.argmax(x, net)
    assert got.shape == (nb, 1, 40, 24, 16) and got.dtype == torch.float32 and torch.equal(got, exp)
    got8 = SlidingWindowArgmaxInferer(roi_size=(16, 16, 16), sw_batch_size=2, overlap=0.5, mode="gaussian", labels_dtype=torch.uint8)(x, net)
    assert got8.dtype == torch.uint8 and torch.equal(got8.float(), exp)

    def two_heads(w):
        a = torch.cat([w * 2.0, w.flip(2) - 0.5, (w - 0.5).abs()], dim=1)
        return a, torch.nn.functional.avg_pool3d(a, 2)

    r2 = inf(x[:1], two_heads)
    g2 = sliding_window_argmax(x[:1], (16, 16, 16), 2, two_heads, overlap=0.5, mode="gaussian")
    for a, b in zip(r2, g2):
        assert torch.equal(b[0], AsDiscrete(argmax=True)(a[0]))
    os.environ["MONAI_AMD_MAX_LOGITS_BYTES"] = str(2.5 * 2 * 5 * 16 ** 3 * 4)       # slab-wise: two window rows at a time
    try:
        assert torch.equal(inf.argmax(x[:1], net), exp[:1])
    finally:
        del os.environ["MONAI_AMD_MAX_LOGITS_BYTES"]
    return True


def case_config0_vs_golden(device):
    """BASELINE.json configs[0]: BasicUNet(1->2), rand 64^3, roi 32^3, sw_batch 4, overlap .5, gaussian."""
    from monai_amd.inferers import SlidingWindowInferer

    g = np.load(os.path.join(GOLDEN, "config0.npz"))
    net, _ = make_net(0, 1, 2, device)
    x = torch.rand(1, 1, 64, 64, 64)  # continues the seed-0 stream exactly like make_golden.py
    assert x.double().sum().item() == float(g["cfg0_x_sum"])
    y = SlidingWindowInferer(roi_size=(32, 32, 32), sw_batch_size=4, overlap=0.5, mode="gaussian")(x.to(device), net)
    r = report(y.cpu(), torch.from_numpy(g["cfg0_out"]))
    assert r["max_abs"] < LOGIT_TOL, r
    return r


def case_blend_only_vs_golden(device):
    """Generic (non-fused) predictor path, bit-exact against the reference's outputs (tests/golden/blend.npz)."""
    from monai_amd.inferers import sliding_window_inference

    g = np.load(os.path.join(GOLDEN, "blend.npz"))

    def toy(k_out):
        return lambda x: torch.cat([torch.sin(x[:, :1] * (1.0 + 0.37 * k)) + 0.05 * k * x[:, :1] for k in range(k_out)], dim=1)

    i = 0
    while f"blend_{i}_shape" in g:
        shape = tuple(int(v) for v in g[f"blend_{i}_shape"])
        torch.manual_seed(int(g[f"blend_{i}_seed"]))
        x = torch.rand(shape)
        k = int(g[f"blend_{i}_k"])
        cpu_toy = toy(k)
        # the predictor arithmetic itself (sin) must be the CPU's to be bit-comparable: evaluate it on the host
        pred = (lambda w: cpu_toy(w.cpu()).to(w.device))
        y = sliding_window_inference(x.to(device), tuple(int(v) for v in g[f"blend_{i}_roi"]), int(g[f"blend_{i}_sw"]), pred,
                                     overlap=float(g[f"blend_{i}_ov"]), mode=str(g[f"blend_{i}_mode"]), padding_mode="constant", cval=-0.5)
        assert np.array_equal(y.cpu().numpy(), g[f"blend_{i}_out"]), f"blend case {i}: not bit-identical to the reference"
        i += 1
    assert i >= 5


def case_buffered_blend_vs_golden(device):
    """`buffer_steps` / `buffer_dim`: the reference's buffered schedule sums in another order than its plain path; the product reproduces that
    order -- bit-exact against the REAL reference's buffered runs (tests/golden/buffered.npz: every buffered axis, several group sizes, padding, 2-D, batch 2)."""
    from monai_amd.inferers import SlidingWindowInferer, sliding_window_inference

    g = np.load(os.path.join(GOLDEN, "buffered.npz"))

    def toy(k_out):
        return lambda x: torch.cat([torch.sin(x[:, :1] * (1.0 + 0.37 * k)) + 0.05 * k * x[:, :1] for k in range(k_out)], dim=1)

    i, differing = 0, 0
    while f"buf_{i}_shape" in g:
        shape = tuple(int(v) for v in g[f"buf_{i}_shape"])
        torch.manual_seed(int(g[f"buf_{i}_seed"]))
        x = torch.rand(shape)
        cpu_toy = toy(int(g[f"buf_{i}_k"]))
        pred = (lambda w: cpu_toy(w.cpu()).to(w.device))      # the predictor's own arithmetic (sin) on the host: only the blend is under test
        kw = dict(overlap=float(g[f"buf_{i}_ov"]), mode=str(g[f"buf_{i}_mode"]), padding_mode="constant", cval=-0.5)
        roi, sw = tuple(int(v) for v in g[f"buf_{i}_roi"]), int(g[f"buf_{i}_sw"])
        steps, dim = int(g[f"buf_{i}_steps"]), int(g[f"buf_{i}_dim"])
        y = sliding_window_inference(x.to(device), roi, sw, pred, buffer_steps=steps, buffer_dim=dim, **kw)
        assert np.array_equal(y.cpu().numpy(), g[f"buf_{i}_out"]), f"buffered case {i}: not bit-identical to the reference's buffered run"
        if i == 1:      # the inferer object: constructor arguments and per-call overrides (inferer.py:525-527)
            y2 = SlidingWindowInferer(roi, sw, buffer_steps=steps, buffer_dim=dim, **kw)(x.to(device), pred)
            y3 = SlidingWindowInferer(roi, sw, **kw)(x.to(device), pred, buffer_steps=steps, buffer_dim=dim)
            assert torch.equal(y2, y) and torch.equal(y3, y)
        differing += int(g[f"buf_{i}_differs_from_plain"])
        i += 1
    assert i >= 8 and differing > 1000      # the goldens do exercise a different summation order
    return i


def case_buffered_calls_vs_golden(device):
    """`buffer_steps` with callbacks in the loop (VERDICT r04 missing #3): `process_fn` (constant and batch-dependent weight maps), `with_coord`, tuple / dict predictor
    outputs -- bit-exact against the REAL reference's buffered runs (tests/golden/buffered_calls.npz): the predictor is called in the buffered order with the sorted
    slices, only the first output is blended (a tensor comes back, or a one-key dict), the count map is the first flush's weight map."""
    from buffered_call_cases import CASES, make_callbacks
    from monai_amd.inferers import sliding_window_inference

    g = np.load(os.path.join(GOLDEN, "buffered_calls.npz"))
    for i, c in enumerate(CASES):
        torch.manual_seed(c["seed"])
        x = torch.rand(c["shape"])
        pred, process_fn = make_callbacks(c, cpu_math=True)
        y = sliding_window_inference(x.to(device), c["roi"], c["sw"], pred, overlap=c["ov"], mode=c["mode"], process_fn=process_fn, buffer_steps=c["steps"],
                                     buffer_dim=c["dim"], with_coord=c["coord"])
        if c["out"] == "dict":
            assert isinstance(y, dict) and list(y) == ["a"]
            y = y["a"]
        assert isinstance(y, torch.Tensor), type(y)
        assert np.array_equal(y.cpu().numpy(), g[f"bc_{i}_out"]), f"buffered-with-callbacks case {i} ({c}): not bit-identical to the reference"
    return len(CASES)


def case_narrow_and_host_inputs(device):
    """half / bfloat16 volumes: computed in fp32, returned in the caller's dtype (== the fp32 result rounded once).  On a
    real device: a CPU volume with sw_device= the ROCm device returns on the CPU with the device result's bits."""
    from monai_amd.inferers import SlidingWindowInferer, sliding_window_inference

    torch.manual_seed(3)
    x = torch.rand(1, 1, 20, 22, 24)
    pred = lambda w: torch.cat([w * 2.0 - 0.5, torch.sin(w)], dim=1)
    kw = dict(overlap=0.5, mode="gaussian")
    ref = sliding_window_inference(x.half().float().to(device), (16, 16, 16), 2, pred, **kw)
    y = sliding_window_inference(x.half().to(device), (16, 16, 16), 2, pred, **kw)
    assert y.dtype == torch.float16 and torch.equal(y, ref.half())
    yb = SlidingWindowInferer((16, 16, 16), 2, **kw)(x.bfloat16().to(device), pred)
    refb = sliding_window_inference(x.bfloat16().float().to(device), (16, 16, 16), 2, pred, **kw)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb, refb.bfloat16())
    if torch.device(device).type == "cuda":
        full = sliding_window_inference(x.to(device), (16, 16, 16), 2, pred, **kw)
        yc = sliding_window_inference(x, (16, 16, 16), 2, pred, sw_device=device, **kw)
        assert yc.device.type == "cpu" and torch.equal(yc, full.cpu())
        yd = sliding_window_inference(x, (16, 16, 16), 2, pred, sw_device=device, device=device, **kw)
        assert yd.device.type == "cuda" and torch.equal(yd, full)


def case_process_fn_vs_golden(device):
    """`process_fn` (utils.py:232-238): predictions edited per batch, a weight map that changes from batch to batch, the
    count map built from the first batch's -- bit-exact against the real reference (tests/golden/make_golden_process_fn.py)."""
    from monai_amd.inferers import sliding_window_inference

    g = np.load(os.path.join(GOLDEN, "process_fn.npz"))

    def toy(x):
        return torch.cat([torch.sin(x[:, :1] * (1.0 + 0.37 * k)) + 0.05 * k * x[:, :1] for k in range(3)], dim=1)

    def make_process_fn():
        calls = {"n": 0}

        def process_fn(segs, win_data, imp):
            # the callback's own arithmetic (mean, scaling) must be the CPU's to be bit-comparable: evaluate it on the host
            calls["n"] += 1
            dev = segs[0].device
            segs = tuple((s.cpu() * 0.5 + 0.25 * win_data.cpu().mean()).to(dev) for s in segs)
            return segs, (imp.cpu() * (1.0 + 0.125 * (calls["n"] % 3)) + 0.0625).to(dev)

        return process_fn

    cpu_toy = toy
    toy = lambda w: cpu_toy(w.cpu()).to(w.device)      # noqa: E731  (same reason)
    i = 0
    while f"pf_{i}_shape" in g:
        torch.manual_seed(40 + i)
        x = torch.rand(tuple(int(v) for v in g[f"pf_{i}_shape"]))
        y = sliding_window_inference(x.to(device), tuple(int(v) for v in g[f"pf_{i}_roi"]), int(g[f"pf_{i}_sw"]), toy,
                                     overlap=float(g[f"pf_{i}_ov"]), mode=str(g[f"pf_{i}_mode"]), process_fn=make_process_fn())
        assert np.array_equal(y.cpu().numpy(), g[f"pf_{i}_out"]), f"process_fn case {i}: not bit-identical to the reference"
        i += 1
    assert i == 2


# ------------------------------------------------------------------------------------------ UNETR
def _digest(sd):
    """all parameters except the position embedding (see tests/golden/make_golden_unetr.py:digest)"""
    import hashlib

    h = hashlib.sha256()
    for k, v in sd.items():
        if k.endswith("position_embeddings"):
            continue
        h.update(k.encode()); h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def case_unetr_small_vs_golden(device):
    """UNETR(img 32^3, hidden 128 = 2 heads x 64, mlp 256, feature 16, 3 classes), seed 2, against the reference's output."""
    from monai_amd.networks.nets import UNETR

    g = np.load(os.path.join(GOLDEN, "unetr.npz"))
    torch.manual_seed(2)
    net = UNETR(in_channels=1, out_channels=3, img_size=(32, 32, 32), feature_size=16, hidden_size=128, mlp_dim=256, num_heads=2).eval()
    assert _digest(net.state_dict()) == str(g["small_state_sha256"]), "same seed must give the reference's weights"
    pe = net.state_dict()["vit.patch_embedding.position_embeddings"].flatten()[::13].numpy()
    np.testing.assert_allclose(pe, g["small_pos_sample"], rtol=2e-6, atol=1e-9)
    net = net.to(device)
    torch.manual_seed(32)
    x = torch.rand(2, 1, 32, 32, 32)
    y = net(x.to(device)).cpu()
    r = report(y, torch.from_numpy(g["small_out"]))
    assert r["max_abs"] < LOGIT_TOL, r
    return r


def case_unetr_vitb_vs_golden(device):
    """BASELINE.json configs[3] network: UNETR(ViT-B/16, 96^3 window, 5 classes), seed 1, one window vs the reference."""
    from monai_amd.networks.nets import UNETR

    g = np.load(os.path.join(GOLDEN, "unetr.npz"))
    torch.manual_seed(1)
    net = UNETR(in_channels=1, out_channels=5, img_size=(96, 96, 96)).eval()
    assert list(net.state_dict().keys()) == list(g["vitb_keys"])
    assert _digest(net.state_dict()) == str(g["vitb_state_sha256"])
    pe = net.state_dict()["vit.patch_embedding.position_embeddings"].flatten()[::997].numpy()
    np.testing.assert_allclose(pe, g["vitb_pos_sample"], rtol=2e-6, atol=1e-9)
    net = net.to(device)
    torch.manual_seed(31)
    x = torch.rand(1, 1, 96, 96, 96)
    y = net(x.to(device)).cpu()
    sub = y[:, :, ::4, ::4, ::4]
    err = (sub - torch.from_numpy(g["vitb_out_sub"])).abs().max().item()
    assert err < LOGIT_TOL, err
    assert abs(y.double().sum().item() - float(g["vitb_out_sum"])) < 1e-4 * y.numel() ** 0.5 + 50.0
    mism = (y.argmax(1)[:, ::2, ::2, ::2].numpy().astype(np.uint8) != g["vitb_argmax_sub"]).mean()
    assert mism < 1e-4, mism
    return err, mism


# ------------------------------------------------------------------------------------------ UNet
UNET_CFGS = {   # tests/golden/make_golden_unet.py
    "res2": dict(channels=(16, 32, 64, 128, 256), strides=(2, 2, 2, 2), num_res_units=2, shape=(2, 1, 32, 32, 32), seed=4),
    "plain": dict(channels=(8, 16, 32), strides=(2, 2), num_res_units=0, shape=(1, 1, 24, 16, 16), seed=5),
    "mixed": dict(channels=(8, 16, 32), strides=(2, 1), num_res_units=1, shape=(1, 1, 16, 12, 20), seed=6),
    # the spleen-bundle configuration: batch norm (evaluated with its running statistics), two residual units
    "batch": dict(channels=(16, 32, 64, 128), strides=(2, 2, 2), num_res_units=2, shape=(2, 1, 32, 32, 32), seed=7, norm="batch"),
}


# activations / `adn_ordering` beyond the default "NDA" + PReLU (tests/golden/make_golden_unet_variants.py -> unet_variants.npz): norm-then-activation in another
# spelling, activation BEFORE the normalisation (statistics of the activated tensor), no normalisation at all, batch norm behind a ReLU
UNET_VARIANTS = {
    "relu_nad": dict(channels=(8, 16, 32), strides=(2, 2), num_res_units=1, shape=(1, 1, 16, 16, 16), seed=41, kw=dict(act="RELU", adn_ordering="NAD")),
    "lrelu_an": dict(channels=(8, 16, 32), strides=(2, 2), num_res_units=2, shape=(2, 1, 16, 16, 24), seed=42,
                     kw=dict(act=("leakyrelu", {"negative_slope": 0.2}), adn_ordering="AN", norm=("instance", {"affine": True}))),
    "prelu_a": dict(channels=(8, 16, 32), strides=(2, 1), num_res_units=0, shape=(1, 1, 16, 12, 20), seed=43, kw=dict(adn_ordering="A")),
    "relu_adn_batch": dict(channels=(8, 16, 32), strides=(2, 2), num_res_units=1, shape=(1, 1, 16, 16, 16), seed=44, norm="batch", kw=dict(act="relu", adn_ordering="ADN")),
}
UNET_CFGS_ALL = {**UNET_CFGS, **UNET_VARIANTS}


def unet_kwargs(name):
    c = UNET_CFGS_ALL[name]
    kw = dict(spatial_dims=3, in_channels=1, out_channels=3, channels=c["channels"], strides=c["strides"], num_res_units=c["num_res_units"])
    if "norm" in c:
        kw["norm"] = c["norm"]
    kw.update(c.get("kw", {}))
    return kw


def perturb_unet(net, name):
    """PReLU slopes distinguishable from the default; batch norm: non-trivial running statistics and affine parameters"""
    cfg = UNET_CFGS_ALL[name]
    affine = "norm" in cfg or "norm" in cfg.get("kw", {})
    gen = torch.Generator().manual_seed(600 + cfg["seed"])
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if k.endswith("adn.A.weight"):
                v.fill_(0.1 + 0.01 * (len(k) % 7))
            elif "norm" in cfg and k.endswith("running_mean"):
                v.copy_(0.1 * torch.randn(v.shape, generator=gen))
            elif "norm" in cfg and k.endswith("running_var"):
                v.copy_(0.75 + 0.5 * torch.rand(v.shape, generator=gen))
            elif affine and k.endswith("adn.N.weight"):
                v.copy_(1.0 + 0.2 * torch.randn(v.shape, generator=gen))
            elif affine and k.endswith("adn.N.bias"):
                v.copy_(0.1 * torch.randn(v.shape, generator=gen))
    return net


def _full_digest(sd):
    import hashlib

    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode()); h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def make_unet(name, device=None):
    """The reference-seeded UNet of tests/golden/unet.npz: same seed -> same init; PReLU slopes set as the generator did."""
    from monai_amd.networks.nets import UNet

    c = UNET_CFGS_ALL[name]
    torch.manual_seed(c["seed"])
    net = UNet(**unet_kwargs(name))
    init = _full_digest(net.state_dict())
    net = perturb_unet(net, name).eval()
    return (net.to(device) if device is not None else net), init


def case_unet_vs_golden(device, names=("res2", "plain", "mixed", "batch"), golden="unet.npz"):
    """MONAI UNet (residual units / plain / stride-1 level; golden="unet_variants.npz": the UNET_VARIANTS) against the reference's own output: keys, init digest, logits."""
    g = np.load(os.path.join(GOLDEN, golden))
    out = {}
    for name in names:
        net, init = make_unet(name)
        assert list(net.state_dict().keys()) == list(g[f"{name}_keys"]), name
        assert init == str(g[f"{name}_init_sha256"]), f"{name}: same seed must give the reference's weights"
        net = net.to(device)
        torch.manual_seed(100 + UNET_CFGS_ALL[name]["seed"])
        x = torch.rand(UNET_CFGS_ALL[name]["shape"])
        y = net(x.to(device)).cpu()
        r = report(y, torch.from_numpy(g[f"{name}_out"]))
        assert r["max_abs"] < LOGIT_TOL, (name, r)
        out[name] = r["max_abs"]
    return out


# ---- 2-D BasicUNet (the 3-D engine on one plane) and SliceInferer over it: SURVEY 8 row a9, tests/golden/make_golden_basic_unet2d.py ----
BASIC2D = {
    "plain": dict(kw=dict(in_channels=1, out_channels=3, features=(16, 16, 32, 32, 64, 16)), shape=(2, 1, 48, 64), seed=31),
    "odd_relu": dict(kw=dict(in_channels=2, out_channels=2, features=(8, 8, 16, 16, 32, 8), act=("relu", {}), norm=("instance", {"affine": False})),
                     shape=(1, 2, 35, 50), seed=32),       # odd extents: UpCat's replicate padding, in-plane only
}
BASIC2D_SLICE = dict(roi_size=(32, 32), sw_batch_size=4, spatial_dim=2, overlap=0.25, mode="gaussian")


def basic2d_build(cls, name):
    import hashlib

    c = BASIC2D[name]
    torch.manual_seed(c["seed"])
    net = cls(spatial_dims=2, **c["kw"])
    h = hashlib.sha256()
    for k, v in net.state_dict().items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    gen = torch.Generator().manual_seed(900 + c["seed"])
    with torch.no_grad():
        for k, v in net.state_dict().items():        # non-default norm affine / bias values: a swapped or dropped parameter would show
            if k.endswith("adn.N.weight"):
                v.copy_(1.0 + 0.2 * torch.randn(v.shape, generator=gen))
            elif k.endswith("bias"):
                v.copy_(0.1 * torch.randn(v.shape, generator=gen))
    return net.eval(), h.hexdigest()


def basic2d_input(name):
    return torch.rand(BASIC2D[name]["shape"], generator=torch.Generator().manual_seed(950 + BASIC2D[name]["seed"]))


def basic2d_volume():
    return torch.rand((1, 1, 40, 48, 4), generator=torch.Generator().manual_seed(803))


def case_basic_unet_2d_vs_reference(device):
    """BasicUNet(spatial_dims=2) on the one-plane engine: the reference's 2-D state_dict keys, same seed => same weights, logits within 1e-4
    of the real reference (incl. odd extents); SliceInferer(spatial_dim=2) over a 4-slice volume against reference inferer + reference net."""
    from monai_amd.inferers import SliceInferer
    from monai_amd.networks.nets import BasicUNet

    g = np.load(os.path.join(GOLDEN, "basic_unet2d.npz"))
    out = {}
    for name in BASIC2D:
        net, init = basic2d_build(BasicUNet, name)
        assert list(net.state_dict().keys()) == list(g[f"{name}_keys"]), name
        assert init == str(g[f"{name}_init_sha256"]), f"{name}: same seed must give the reference's weights"
        y = net.to(device)(basic2d_input(name).to(device)).cpu()
        exp = torch.from_numpy(g[f"{name}_out"])
        assert y.shape == exp.shape
        out[name] = (y.double() - exp.double()).abs().max().item()
        assert out[name] < LOGIT_TOL, (name, out[name])
    net, _ = basic2d_build(BasicUNet, "plain")
    y = SliceInferer(**BASIC2D_SLICE)(basic2d_volume().to(device), net.to(device)).cpu()
    exp = torch.from_numpy(g["plain_slice_out"])
    assert y.shape == exp.shape
    out["slice_inferer"] = (y.double() - exp.double()).abs().max().item()
    assert out["slice_inferer"] < LOGIT_TOL, out
    return out


# ------------------------------------------------------------------------------------------ trained-like parameter spreads
def _spread_affine(net, seed, lo=1e-3, hi=1e3, beta_max=1e3):
    """Give every normalisation layer of `net` trained-checkpoint-like (and far wilder) affine parameters: gamma log-uniform in [lo, hi] with random
    signs, beta up to +-beta_max * gamma-scale -- default-initialised nets (gamma 1, beta 0) say nothing about the split-precision kernel's range."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (torch.nn.InstanceNorm3d, torch.nn.GroupNorm)) and getattr(m, "weight", None) is not None:
                c = m.weight.numel()
                mag = torch.exp(torch.rand(c, generator=gen) * (np.log(hi) - np.log(lo)) + np.log(lo))
                sign = torch.where(torch.rand(c, generator=gen) > 0.75, -1.0, 1.0)
                m.weight.copy_((mag * sign).to(m.weight.device))
                m.bias.copy_((torch.randn(c, generator=gen) * mag * (beta_max / hi) * 3.0).to(m.bias.device))
    return net


def _rel_err(got, exp):
    return (got.double() - exp.double()).abs().max().item() / max(1.0, exp.double().abs().max().item())


def case_nets_with_spread_affine(device, window=(32, 32, 32), nets=("basic_unet", "dynunet_res", "segresnet")):
    """BasicUNet / DynUNet(res) / SegResNet whose norm layers carry gamma in +-[1e-3, 1e3] and |beta| up to ~1e3 (activations from 1e-3 to far beyond
    65504 inside the net), fed one [0, 1] window and one raw-CT-like window (x 3000 - 1000): the default path (fp16 split-precision convolutions, scaled
    by the records' magnitude bounds) against the CPU oracle at the north-star bar -- 1e-4 of the logit scale -- and against the exact-fp32 kernels'
    own distance from the oracle (the split-precision path must not be a worse fp32 than fp32)."""
    from monai_amd import config
    from monai_amd.networks.nets import BasicUNet, DynUNet, SegResNet
    from oracle import dynunet as odyn

    gen = torch.Generator().manual_seed(99)
    x = torch.rand((2, 1) + tuple(window), generator=gen)
    x[1] = x[1] * 3000.0 - 1000.0
    out = {}
    for name in nets:
        torch.manual_seed(5)
        if name == "basic_unet":
            net = BasicUNet(3, 1, 5).eval()
            ref_fn = lambda sd, v: oracle.basic_unet_forward(sd, v)      # noqa: E731
        elif name == "dynunet_res":
            kw = dict(kernel_size=[3, 3, 3, 3], strides=[1, 2, 2, 2], upsample_kernel_size=[2, 2, 2], filters=[32, 32, 64, 64], res_block=True,
                      norm_name=("instance", {"affine": True}))
            net = DynUNet(3, 1, 3, **kw).eval()
            ref_fn = lambda sd, v: odyn.dynunet_forward(sd, v, kw["strides"], res_block=True)      # noqa: E731
        else:
            net = SegResNet(spatial_dims=3, init_filters=16, in_channels=1, out_channels=3, blocks_down=(1, 2, 2), blocks_up=(1, 1)).eval()
            ref_fn = lambda sd, v: odyn.segresnet_forward(sd, v, blocks_down=(1, 2, 2), blocks_up=(1, 1))      # noqa: E731
        _spread_affine(net, seed=17)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        with torch.no_grad():
            exp = ref_fn(sd, x)
        assert torch.isfinite(exp).all(), name
        net = net.to(device)
        errs = {}
        saved = config.CONV_ALGO
        try:
            for algo in ("auto", "fp32"):
                config.CONV_ALGO = algo
                got = net(x.to(device)).cpu()
                assert torch.isfinite(got).all(), f"{name}/{algo}: non-finite logits"
                errs[algo] = _rel_err(got, exp)
        finally:
            config.CONV_ALGO = saved
        assert errs["auto"] < LOGIT_TOL, f"{name}: split-precision path {errs['auto']:.2e} of the logit scale (fp32 kernels: {errs['fp32']:.2e})"
        assert errs["auto"] < 4.0 * errs["fp32"] + 2e-6, f"{name}: split precision {errs['auto']:.2e} vs exact fp32 {errs['fp32']:.2e}"
        out[name] = errs
    return out


def case_net_nonfinite_inputs(device, window=(32, 32, 32), features=(32, 32, 64, 128, 256, 32)):
    """inf / NaN voxels in one window of a batch: the reference (conv -> InstanceNorm, blocks/convolutions.py:98-171) turns THAT sample into NaN and
    leaves the others alone; so does the engine on its default (split-precision) path and on the exact-fp32 kernels."""
    from monai_amd import config

    net, sd = make_net(1, 1, 5, device, features=features)
    gen = torch.Generator().manual_seed(98)
    x = torch.rand((3, 1) + tuple(window), generator=gen)
    x[1, 0, 5, 6, 7] = float("inf")
    x[2, 0, 9, 9, 9] = float("nan")
    with torch.no_grad():
        exp = oracle.basic_unet_forward(sd, x)
    assert torch.isnan(exp[1]).all() and torch.isnan(exp[2]).all() and torch.isfinite(exp[0]).all()
    saved = config.CONV_ALGO
    try:
        for algo in ("auto", "fp32"):
            config.CONV_ALGO = algo
            got = net(x.to(device)).cpu()
            assert torch.isnan(got[1]).all() and torch.isnan(got[2]).all(), f"{algo}: a non-finite window must come out NaN like the reference's"
            assert (got[0] - exp[0]).abs().max().item() < LOGIT_TOL, algo
    finally:
        config.CONV_ALGO = saved


MOSAIC_CASES = (((1, 1, 40, 56, 36), 0.5, "gaussian"), ((1, 1, 44, 32, 52), 0.25, "constant"), ((2, 1, 32, 32, 32), 0.5, "gaussian"))


def case_mosaic_layout_equals_window_major(device, cases=MOSAIC_CASES, features=(16, 16, 32, 32, 64, 16)):
    """The fused single-GPU path keeps its logits in the mosaic layout (ops.LogitsMosaic) and the network's last kernel writes it directly; the window-major
    buffer (MONAI_AMD_LOGITS_LAYOUT=windows; what window sharding and the fused argmax use) gives the SAME BITS -- overlap 0.5 with clipped last windows,
    overlap 0.25, a volume of one window, constant and gaussian weights."""
    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.inferers import utils as U

    net, _ = make_net(1, 1, 5, device, features=features)
    used = []
    real = U._alloc_mosaic

    def spy(*a, **k):
        m = real(*a, **k)
        used.append(m is not None)
        return m

    U._alloc_mosaic = spy
    try:
        for shape, overlap, mode in cases:
            x = torch.rand(shape, generator=torch.Generator().manual_seed(61)).to(device)
            inf = SlidingWindowInferer(roi_size=(32, 32, 32), sw_batch_size=3, overlap=overlap, mode=mode)
            n0 = len(used)
            a = inf(x, net)
            assert used[n0:] and all(used[n0:]), "the mosaic layout did not run"
            os.environ["MONAI_AMD_LOGITS_LAYOUT"] = "windows"
            try:
                b = inf(x, net)
            finally:
                del os.environ["MONAI_AMD_LOGITS_LAYOUT"]
            assert torch.equal(a, b), f"{shape} overlap {overlap}: mosaic and window-major logits layouts differ by {(a - b).abs().max().item()}"
    finally:
        U._alloc_mosaic = real


def case_conv_cout_16_mod_32_split(device, cin=32, cout=48, dims=(6, 16, 16), n=2):
    """The conv engine of UNETR / SwinUNETR (`UNETR._conv3_in`) on a layer whose output channels are 16 mod 32 (SwinUNETR(48): 48): under the split-precision
    family the first cout - 16 channels go through the 32-couts form, the last 16 through the 16-couts form -- two launches into channel slices of one output,
    two statistics sets, two finalizes into slices of one record tensor.  Output against fp64 conv3d of the activated input, records against the output's own
    statistics; both configurations must have run; under the exact-fp32 family there is one launch."""
    import torch.nn.functional as F

    import kernel_cases as kc
    from monai_amd import config, ops
    from monai_amd.networks.nets import UNETR

    gen = torch.Generator().manual_seed(77)
    x = torch.randn((n, cin) + tuple(dims), generator=gen)
    nrm = kc._with_bounds(x, kc._rand_nrm(n, cin, gen), loosen=float(np.sqrt(np.prod(dims))))
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) / np.sqrt(27.0 * cin))
    exp = F.conv3d(kc._act(x.double(), nrm.double()), conv.weight.double(), padding=1)
    conv = conv.to(device)
    seen = []
    real = ops.conv3d_k3

    def spy(cfg, *a, **k):
        seen.append(int(cfg))
        return real(cfg, *a, **k)

    res = {}
    ops.conv3d_k3 = spy
    try:
        for algo in ("auto", "fp32"):
            eng = object.__new__(UNETR)          # the conv-engine helpers need only these two attributes
            eng._packed, eng._stats = {}, None
            seen.clear()
            with config.conv_algo_scope(algo):
                out, rec = eng._conv3_in(conv, x.to(device), nrm.to(device), 0.01)
            if algo == "auto":
                assert seen == [ops.conv3d_k3_h2_config(), ops.conv3d_k3_h2c_config()], seen
            else:
                assert len(seen) == 1 and seen[0] <= ops.conv3d_k3_num_configs(), seen
            got = out.cpu().double()
            err = (got - exp).abs().max().item()
            assert err < 2e-5 * max(1.0, exp.abs().max().item()), (algo, err)
            r = rec.cpu().double()
            alpha = 1.0 / torch.sqrt(got.var(dim=(2, 3, 4), unbiased=False) + 1e-5)
            assert ((r[:, :, 0] - alpha).abs() / alpha).max().item() < 1e-5, algo
            assert (r[:, :, 1] + got.mean(dim=(2, 3, 4)) * alpha).abs().max().item() < 2e-5, algo
            assert torch.all(r[:, :, 2] == torch.tensor(0.01, dtype=torch.float32).double()) and torch.all(r[:, :, 3] >= kc._amax(got, r[:, :, :3]))
            res[algo] = err
    finally:
        ops.conv3d_k3 = real
    return res
