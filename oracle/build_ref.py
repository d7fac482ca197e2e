"""Oracle (test infrastructure): build the REFERENCE's own native resampler (CPU part of monai/csrc) as
``oracle/_ref/monai_ref_C.so``.

The sources are compiled where they lie under /root/reference (nothing is copied into this repository); only the
build directory ``oracle/_ref/`` is written (git-ignored; it travels to the GPU box with the snapshot, where
``load()`` just dlopens the prebuilt file).  Only the ``.cpp`` files are compiled -- the ``.cu`` files would need
hipify, which this project does not do.  The module exposes the reference's pybind11 surface
(``grid_pull``, ``BoundType``, ``InterpolationType`` ... -- monai/csrc/ext.cpp:20-80) and is used ONLY by tests as the
checker for ``monai_amd._C.grid_pull``.

    python oracle/build_ref.py        # ~3 minutes, g++ + torch headers
"""
from __future__ import annotations

import glob
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF_CSRC = "/root/reference/monai/csrc"
NAME = "monai_ref_C"


def so_path() -> str:
    return os.path.join(OUT, NAME + ".so")


def build(verbose: bool = False) -> str | None:
    """Compile when the reference is present and the .so is missing; return the path (None if unavailable)."""
    if os.path.exists(so_path()):
        return so_path()
    if not os.path.isdir(REF_CSRC):
        return None
    import torch
    from torch.utils import cpp_extension

    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(REF_CSRC, "**", "*.cpp"), recursive=True))
    ver = torch.__version__.split("+")[0].split(".")
    cpp_extension.load(
        name=NAME, sources=srcs, extra_include_paths=[REF_CSRC], build_directory=OUT, verbose=verbose,
        extra_cflags=["-O2", "-fopenmp", "-DAT_PARALLEL_OPENMP=1", f"-DMONAI_TORCH_VERSION={int(ver[0]) * 10000 + int(ver[1]) * 100}"],
        extra_ldflags=["-fopenmp"], is_python_module=True,
    )
    return so_path() if os.path.exists(so_path()) else None


def load():
    """Import the prebuilt module (None when it was never built, e.g. a checkout without oracle/_ref)."""
    path = so_path()
    if not os.path.exists(path):
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)

    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
