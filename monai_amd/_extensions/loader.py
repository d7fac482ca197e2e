"""``load_module`` -- the JIT extension loader of monai/_extensions/loader.py:49-93, for gfx950.

Same signature and naming rules as the reference: ``module_name`` names a source directory, ``defines`` become ``-D`` flags and
part of the cached artefact's name (so two configurations never share a binary), the build runs under ``build_timeout``.  The
backend is different by design: sources are HIP (``*.hip``, plus host ``*.cpp``) compiled by ``hipcc --offload-arch=gfx950`` into
a C-ABI shared library (hipcc cross-compiles without a GPU), and the returned module exposes the library's ``extern "C"`` symbols
through ``ctypes`` -- there is no pybind11 layer on this path.  ``load_module("monai_amd")`` builds the
library of this package (csrc/capi.hip) under the given defines.  The reference's own JIT extension (``gmm``) is out of scope
(SURVEY.md 8b, B2): asking for it raises the reference's ``ValueError("No extension module named gmm")``.
"""

from __future__ import annotations

import ctypes
import os
import platform
import signal
import subprocess
import sys
from glob import glob
from types import ModuleType

import torch

dir_path = os.path.dirname(os.path.realpath(__file__))
# directories searched for ``<module_name>/`` source trees (the package's own first); tests / users may append
EXTENSION_DIRS = [dir_path]
BUILD_DIR = os.path.join(dir_path, "_build")
ARCH = "gfx950"

__all__ = ["load_module"]


def _rm(path: str) -> None:
    if os.path.exists(path):
        os.remove(path)


def _package_sources():
    csrc = os.path.join(os.path.dirname(dir_path), "csrc")
    return [os.path.join(csrc, "capi.hip")], sorted(glob(os.path.join(csrc, "kernels", "*.h"))) + [os.path.join(os.path.dirname(os.path.dirname(dir_path)), "include", "monai_amd.h")]


def load_module(module_name: str, defines: dict | None = None, verbose_build: bool = False, build_timeout: int = 300) -> ModuleType:
    """Build (or reuse) the extension ``module_name`` and return it as a module whose attributes are the library's C symbols
    (``module.<symbol>`` is a ``ctypes`` function; set ``argtypes`` / ``restype`` as for any C-ABI entry point);
    ``module.__file__`` is the shared library, ``module.cdll`` the ``ctypes.CDLL``."""
    from ..build import FLAGS, hipcc

    if os.path.basename(module_name) != module_name or module_name in ("", ".", ".."):     # never resolve outside EXTENSION_DIRS
        raise ValueError(f"No extension module named {module_name}")
    if module_name == "monai_amd":
        sources, deps = _package_sources()
    else:
        module_dir = next((os.path.join(d, module_name) for d in EXTENSION_DIRS if os.path.isdir(os.path.join(d, module_name))), None)
        if module_dir is None or module_name.startswith("_"):
            raise ValueError(f"No extension module named {module_name}")
        sources = sorted(glob(os.path.join(module_dir, "**", "*.hip"), recursive=True) + glob(os.path.join(module_dir, "**", "*.cpp"), recursive=True))
        deps = sorted(glob(os.path.join(module_dir, "**", "*.h"), recursive=True))
        if not sources:
            raise ValueError(f"No extension module named {module_name}")

    # loader.py:69-80: platform, interpreter, torch version, device toolchain and the define VALUES are part of the name
    tv = torch.__version__.split("+")[0].split(".")[:2]
    platform_str = f"_{platform.system()}_{platform.python_version()}_" + "".join(tv) + f"_{ARCH}_{torch.version.hip or 'hip'}"
    name = module_name if defines is None else "_".join([module_name] + [f"{v}" for v in defines.values()])
    name = (name + platform_str).replace(".", "_").replace("/", "_").replace(" ", "_")
    define_args = [] if not defines else [f"-D{key}={defines[key]}" for key in defines]

    os.makedirs(BUILD_DIR, exist_ok=True)
    out = os.path.join(BUILD_DIR, f"{name}.so")
    stale = not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(s) for s in sources + deps)
    if stale:
        tmp = f"{out}.{os.getpid()}.tmp"
        # `-x hip` applies to the files that follow it: device sources first, then `-x none` so host *.cpp files compile as plain C++
        hip_src = [f for f in sources if not f.endswith(".cpp")]
        cpp_src = [f for f in sources if f.endswith(".cpp")]
        cmd = [hipcc()] + FLAGS + define_args + (["-x", "hip"] + hip_src if hip_src else []) + (["-x", "none"] + cpp_src if cpp_src else []) + ["-o", tmp]
        if verbose_build:
            print(" ".join(cmd), file=sys.stderr)
        pipe = None if verbose_build else subprocess.PIPE
        proc = subprocess.Popen(cmd, stdout=pipe, stderr=pipe, text=True, start_new_session=True)     # own process group: hipcc forks clang / lld
        try:
            _, err = proc.communicate(timeout=build_timeout)
        except BaseException as e:          # timeout, Ctrl-C, anything: the detached compiler group must not outlive the call
            try:
                os.killpg(proc.pid, signal.SIGKILL)       # exactly the group started above, compiler children included
            except ProcessLookupError:
                pass
            proc.wait()
            _rm(tmp)
            if isinstance(e, subprocess.TimeoutExpired):
                raise TimeoutError("Build appears to be blocked. Is there a stopped process building the same extension?") from e
            raise
        if proc.returncode != 0:
            _rm(tmp)
            raise RuntimeError(f"Error building extension '{name}'" + ("" if verbose_build else f":\n{err}"))
        os.replace(tmp, out)        # atomic: a concurrent builder of the same configuration never sees a partial file

    cdll = ctypes.CDLL(out)
    mod = ModuleType(name)
    mod.__file__ = out
    mod.cdll = cdll
    mod.__getattr__ = lambda sym: getattr(cdll, sym)      # PEP 562: unknown attributes resolve to exported symbols
    return mod
