"""Build the HIP extension in-tree:  ``python -m monai_amd.build``  ->  monai_amd/csrc/libmonai_amd.so

One translation unit (csrc/capi.hip + kernels/*.h), compiled for gfx950 only.  hipcc cross-compiles without a
GPU, so this also runs in the GPU-less build container; the resulting .so travels to the GPU box in-tree."""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "capi.hip")
OUT = os.path.join(HERE, "csrc", "libmonai_amd.so")
# -ffp-contract=off: no implicit FMA formation anywhere.  The blend must round the multiply and the add
# separately to stay bit-identical to the reference (HIP's __fmul_rn/__fadd_rn are plain * and + and WOULD be
# fused under the default -ffp-contract=fast); every intended FMA in the kernels is an explicit fmaf().
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value"]


def _deps():
    kd = os.path.join(HERE, "csrc", "kernels")
    deps = [SRC, os.path.join(os.path.dirname(HERE), "include", "monai_amd.h")]
    deps += [os.path.join(kd, f) for f in sorted(os.listdir(kd)) if f.endswith(".h")]
    return deps


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("monai_amd.build: hipcc not found (set HIPCC or install ROCm)")


OUT_DEV = os.path.join(HERE, "csrc", "libmonai_amd_dev.so")


def build(force: bool = False, verbose: bool = False, dev: bool = False) -> str:
    """dev=True: the measurement flavour (-DMH_DEV_KNOBS: A/B knobs read from the environment, capi.hip) as libmonai_amd_dev.so --
    loaded only when MONAI_AMD_LIB names it (tools/); the product library has no knobs."""
    out = OUT_DEV if dev else OUT
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in _deps()):
        return out
    cmd = [hipcc()] + FLAGS + (["-DMH_DEV_KNOBS"] + [f for f in os.environ.get("MONAI_AMD_DEV_FLAGS", "").split() if f] if dev else []) + [SRC, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv))
