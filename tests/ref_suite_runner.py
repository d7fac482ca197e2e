"""TEST INFRASTRUCTURE: run ONE of the reference's own unittest modules (/root/reference/tests/...) over the product classes.

    python tests/ref_suite_runner.py /root/reference/tests/inferers/test_sliding_window_inference.py
    python tests/ref_suite_runner.py /root/reference/tests/networks/layers/test_grid_pull.py --compiled      # BUILD_MONAI=1, monai._C = monai_amd._C

`monai_amd.patch.install()` rebinds the reference's names to the MI355X classes, the SIMT-emulator build of the kernels stands in for
the GPU (tests/emu_backend.py: host pointers accepted, dtype rules kept), and the module's tests run unmodified.  Calls the HIP path does
not cover fall through to the reference (boundary B3) -- the last line reports how many kernel launches the module caused and which
components fell through, so a green run cannot be a run that never touched the product.  `parameterized` is not part of this image: a
minimal stand-in (tests/ref_shims) provides `parameterized.expand`.  Prints  RESULT {json}  as its last line."""
import importlib.util
import json
import os
import sys
import unittest
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "ref_shims"), os.path.dirname(HERE), HERE, "/root/reference"]
os.environ.setdefault("MONAI_AMD_CONV_ALGO", "fp32")      # exact-fp32 convolutions: the split-precision default is 10x slower to emulate


def main(path: str, compiled: bool = False, skip=()) -> int:
    warnings.filterwarnings("ignore")
    import torch  # noqa: F401
    from emu_backend import emu_backend

    if compiled:
        # the reference's BUILD_MONAI=1 state (monai/config/deviceconfig.py: USE_COMPILED = HAS_EXT and BUILD_MONAI == "1"), with
        # monai_amd._C standing in for the compiled extension `monai._C` -- it has to be importable before `import monai`
        os.environ["BUILD_MONAI"] = "1"
        from monai_amd import _C

        sys.modules["monai._C"] = _C
    import monai  # noqa: F401
    from monai_amd import _fallback, _lib, patch

    launches = {"n": 0}
    with emu_backend() as lib:
        real_call = type(lib).call

        def counting_call(self, name, *a):
            launches["n"] += 1
            return real_call(self, name, *a)

        type(lib).call = counting_call
        try:
            patch.install()
            spec = importlib.util.spec_from_file_location("ref_suite_module", path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules["ref_suite_module"] = mod
            spec.loader.exec_module(mod)
            suite = unittest.defaultTestLoader.loadTestsFromModule(mod)
            if skip:      # tests that cannot mean on the emulator what they mean on a real system (each one explained where it is listed)
                def strip(su):
                    out = unittest.TestSuite()
                    for t in su:
                        if isinstance(t, unittest.TestSuite):
                            out.addTest(strip(t))
                        elif t._testMethodName not in skip:
                            out.addTest(t)
                    return out

                suite = strip(suite)
            res = unittest.TextTestRunner(verbosity=1, stream=sys.stderr).run(suite)
        finally:
            type(lib).call = real_call
    # tests that need a package this image does not have (torchvision, nibabel, ...) are the environment's, not the product's
    env = [(t, tb) for t, tb in res.errors if "ModuleNotFoundError" in tb.splitlines()[-1] or "OptionalImportError" in tb.splitlines()[-1]]
    res.errors[:] = [e for e in res.errors if e not in env]
    fell = {}
    for comp, _ in _fallback.fell_through():
        fell[comp] = fell.get(comp, 0) + 1
    out = {"module": os.path.relpath(path, "/root/reference/tests"), "run": res.testsRun, "failures": len(res.failures), "errors": len(res.errors),
           "skipped": len(res.skipped) + len(env), "missing_packages": len(env), "kernel_launches": launches["n"], "fell_through": fell,
           "failed": [str(t) for t, _ in res.failures + res.errors][:40]}
    print("RESULT " + json.dumps(out))
    return 0 if not res.failures and not res.errors else 1


if __name__ == "__main__":
    _skip = tuple(n for a in sys.argv[2:] if a.startswith("--skip=") for n in a[len("--skip="):].split(",") if n)
    sys.exit(main(sys.argv[1], compiled="--compiled" in sys.argv[2:], skip=_skip))
