"""TEST INFRASTRUCTURE: route monai_amd's ctypes binding to the x86 SIMT-emulator build of the kernels.

Usage in a CPU test:   ``with emu_backend(): ...``  (or the ``emu`` pytest fixture in conftest.py).
Inside the context, `monai_amd._lib.lib()` returns the emulator library and the "must be a ROCm tensor"
check is lifted so CPU tensors (host pointers) can be passed.  Nothing in the product package knows this
exists."""
import contextlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))


@contextlib.contextmanager
def emu_backend():
    import build_emu

    from monai_amd import _lib

    path = build_emu.build()
    saved = (_lib._LIB, _lib.require_device)
    _lib._LIB = _lib.Library(path)
    _lib.require_device = lambda *a, **k: None
    try:
        yield _lib._LIB
    finally:
        _lib._LIB, _lib.require_device = saved
