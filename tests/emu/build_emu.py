"""TEST INFRASTRUCTURE: build the x86 SIMT-emulator flavour of the kernel library.

Compiles the UNMODIFIED product translation unit ``monai_amd/csrc/capi.hip`` as host C++ with
``tests/emu/stub`` first on the include path, so ``<hip/hip_runtime.h>`` resolves to the fiber-based
emulator.  The result (``tests/emu/_build/libmonai_amd_emu.so``) exports the same C ABI but takes HOST
pointers; CPU tests load it explicitly to check kernel logic against the oracle.  The product package
never loads it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libmonai_amd_emu.so")
SRC = os.path.join(ROOT, "monai_amd", "csrc", "capi.hip")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def sources():
    deps = [SRC, os.path.abspath(__file__), os.path.join(HERE, "stub", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "monai_amd.h")]      # this file: the compiler flags
    kd = os.path.join(ROOT, "monai_amd", "csrc", "kernels")
    deps += [os.path.join(kd, f) for f in sorted(os.listdir(kd)) if f.endswith(".h")]
    return deps


def _fresh() -> bool:
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in sources())


def build(force: bool = False) -> str:
    """One build at a time across processes (pytest-xdist starts several workers that all need the library): the others wait on the
    lock and find it fresh; the compiler writes a temporary file that is renamed into place, so nobody ever loads a half-written one."""
    import fcntl

    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and _fresh():
        return OUT
    with open(os.path.join(OUT_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _fresh():
                return OUT
            cxx = CLANG if os.path.exists(CLANG) else "clang++"
            tmp = f"{OUT}.{os.getpid()}.tmp"
            cmd = [cxx, "-x", "c++", "-std=c++17", "-O2", "-g0", "-fPIC", "-shared", "-pthread", "-mfma", "-mavx2",
                   "-ffp-contract=off", "-fno-strict-aliasing", "-Wno-unused-value", "-Wno-psabi", "-DMH_BLEND_IDX32_LIMIT=50000", "-I", os.path.join(HERE, "stub"), SRC, "-o", tmp]
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, OUT)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
