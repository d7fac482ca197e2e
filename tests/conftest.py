"""pytest configuration: registers the ``gpu`` marker and puts the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fallthrough: the test exercises the fall-through to an importable reference MONAI (monai_amd/_fallback.py)")
    config.addinivalue_line("markers", "heavy_emu: minutes of SIMT emulation whose -m gpu twin runs the same case on the MI355X every round; "
                                       "skipped on the CPU unless MONAI_AMD_HEAVY_EMU=1 (keeps `pytest -m 'not gpu'` to a few minutes)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("MONAI_AMD_HEAVY_EMU") == "1":
        return
    skip = pytest.mark.skip(reason="heavy emulator case (its -m gpu twin runs on the MI355X); MONAI_AMD_HEAVY_EMU=1 runs it here")
    for item in items:
        if "heavy_emu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _explicit_errors_unless_asked(request, monkeypatch):
    """Tests pin the explicit `NotImplementedError` / `RuntimeError` of calls outside the HIP path; once another test of the same
    worker process has imported the reference MONAI those calls would fall through to it instead -- only the tests marked
    `fallthrough` want that."""
    if "fallthrough" not in request.keywords:
        monkeypatch.setenv("MONAI_AMD_NO_FALLTHROUGH", "1")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture()
def emu():
    """Kernel library compiled for the x86 SIMT emulator (tests/emu); CPU tensors allowed inside."""
    from emu_backend import emu_backend

    # the emulator runs the exact-fp32 convolution kernels unless a test asks otherwise: the fp16 split-precision kernel (the
    # product default, conv3d_h2.h) is an order of magnitude slower to EMULATE; its own kernel / network cases select it explicitly
    prev = os.environ.get("MONAI_AMD_CONV_ALGO")
    if prev is None:
        os.environ["MONAI_AMD_CONV_ALGO"] = "fp32"
    try:
        with emu_backend() as lib:
            yield lib
    finally:
        if prev is None:
            os.environ.pop("MONAI_AMD_CONV_ALGO", None)
