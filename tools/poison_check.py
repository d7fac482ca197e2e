"""Uninitialised-read detector for the product path: run the same inference with every ``torch.empty`` buffer pre-filled with
different bit patterns (NaN, a huge finite value, zeros) and compare the results BITWISE.  A kernel that reads a cell it (or its
producer) never wrote shows up as a difference between the patterns -- the same defect that shows up on a GPU as a run-to-run
difference when the caching allocator hands out different garbage (DESIGN_HISTORY 6.0: the run-to-run mismatch recorded in round 3).

Runs on an MI355X (``--device cuda``) and on the build container through the SIMT emulator (``--device cpu``, small sizes).
Test infrastructure: nothing in monai_amd/ imports it."""
from __future__ import annotations

import argparse
import contextlib
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

PATTERNS = {"nan": 0x7FC00000, "huge": 0x7F7FFFFF, "neg": 0xFF7FFFFE, "zero": 0x00000000, "one": 0x3F800000}


@contextlib.contextmanager
def poisoned_empty(bits: int):
    """torch.empty / Tensor.new_empty / empty_like return memory filled with the 32-bit pattern `bits` (float32 / int32 tensors)."""
    real_empty, real_like = torch.empty, torch.empty_like
    val = struct.unpack("f", struct.pack("I", bits))[0]

    def fill(t):
        if t.dtype == torch.float32:
            t.view(torch.int32).fill_(bits if bits < 2**31 else bits - 2**32)
        elif t.is_floating_point():
            t.fill_(val)
        elif t.dtype in (torch.int32, torch.int64, torch.uint8, torch.int16, torch.int8):
            t.fill_(-1 if bits else 0)
        return t

    def empty(*a, **k):
        return fill(real_empty(*a, **k))

    def empty_like(*a, **k):
        return fill(real_like(*a, **k))

    torch.empty, torch.empty_like = empty, empty_like
    try:
        yield
    finally:
        torch.empty, torch.empty_like = real_empty, real_like


def run_case(device: str, net_name: str, size, roi, seed: int = 1):
    from monai_amd.inferers import SlidingWindowInferer
    from monai_amd.networks import nets

    torch.manual_seed(seed)
    if net_name == "basicunet":
        net = nets.BasicUNet(3, 1, 5).eval()
    elif net_name == "unet":
        net = nets.UNet(3, 1, 5, channels=(16, 32, 64), strides=(2, 2), num_res_units=2).eval()
    elif net_name == "unetr":
        net = nets.UNETR(1, 5, img_size=roi, hidden_size=96, mlp_dim=192, num_heads=4).eval()
    else:
        raise ValueError(net_name)
    net = net.to(device)
    torch.manual_seed(23)
    x = torch.rand((1, 1) + tuple(size)).to(device)
    y = SlidingWindowInferer(roi_size=roi, sw_batch_size=4, overlap=0.5, mode="gaussian")(x, net)
    if device != "cpu":
        torch.cuda.synchronize()
    return y.detach().cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--net", default="basicunet")
    ap.add_argument("--size", type=int, nargs=3, default=[48, 40, 32])
    ap.add_argument("--roi", type=int, nargs=3, default=[32, 32, 32])
    ap.add_argument("--algo", default=None, help="monai_amd.config.CONV_ALGO (h2 | fp32 ...)")
    ap.add_argument("--patterns", nargs="*", default=["zero", "nan", "huge"])
    a = ap.parse_args()
    if a.algo:
        os.environ["MONAI_AMD_CONV_ALGO"] = a.algo
    ctx = contextlib.nullcontext()
    if a.device == "cpu":
        from emu_backend import emu_backend

        ctx = emu_backend()
    with ctx:
        ref = None
        bad = 0
        for name in a.patterns:
            with poisoned_empty(PATTERNS[name]):
                y = run_case(a.device, a.net, a.size, tuple(a.roi))
            fin = bool(torch.isfinite(y).all())
            if ref is None:
                ref = y
                print(f"pattern {name:5s}: reference run, finite={fin}, checksum {y.double().sum().item()!r}")
                continue
            same = torch.equal(y.view(torch.int32), ref.view(torch.int32))
            nd = int((y.view(torch.int32) != ref.view(torch.int32)).sum())
            print(f"pattern {name:5s}: finite={fin}, bitwise equal to the first pattern: {same} ({nd} words differ)")
            bad += 0 if same else 1
    print("POISON CHECK", "CLEAN" if bad == 0 else f"FAILED ({bad} patterns differ)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
