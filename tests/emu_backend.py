"""TEST INFRASTRUCTURE: route monai_amd's ctypes binding to the x86 SIMT-emulator build of the kernels.

Usage in a CPU test:   ``with emu_backend(): ...``  (or the ``emu`` pytest fixture in conftest.py).
Inside the context, `monai_amd._lib.lib()` returns the emulator library and the "must be a ROCm tensor"
check is lifted so CPU tensors (host pointers) can be passed (the dtype check stays).  Nothing in the product package knows this
exists."""
import contextlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))


@contextlib.contextmanager
def emu_backend():
    import build_emu
    import torch

    from monai_amd import _lib

    path = build_emu.build()
    saved = (_lib._LIB, _lib.require_device)
    _lib._LIB = _lib.Library(path)

    def host_ok(*tensors, dtypes=(torch.float32,)):
        # only the "must be a ROCm tensor" half is lifted: the dtype half stays, so calls the product would hand to the reference
        # (double volumes, ...) still do
        for t in tensors:
            if t is not None and t.dtype not in dtypes:
                raise _lib.UnsupportedOnDevice(f"monai_amd: dtype {t.dtype} is not accepted on this path (expected one of {dtypes})")

    _lib.require_device = host_ok
    try:
        yield _lib._LIB
    finally:
        _lib._LIB, _lib.require_device = saved
