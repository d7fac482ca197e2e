"""Golden vectors for oracle/synthetic.py from the REAL reference (build container only):
    PYTHONPATH=/root/reference python tests/golden/make_golden_synthetic.py
Stores small volumes in full and the 512^3 benchmark volume as a sha256 digest + a few probes."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from monai.data import create_test_image_3d  # noqa: E402

out = {}
cases = {
    "a": dict(height=48, width=40, depth=44, num_objs=6, rad_max=12, rad_min=3, noise_max=0.2, num_seg_classes=4),
    "b": dict(height=33, width=35, depth=31, num_objs=3, rad_max=9, rad_min=2, noise_max=0.0, num_seg_classes=1),
    "c": dict(height=64, width=64, depth=64, num_objs=40, rad_max=30, rad_min=10, noise_max=0.2, num_seg_classes=4),
}
for name, kw in cases.items():
    img, lab = create_test_image_3d(random_state=np.random.RandomState(7 if name != "c" else 0), **kw)
    out[f"{name}_img"], out[f"{name}_lab"] = img, lab.astype(np.int8)
    out[f"{name}_args"] = np.array([kw[k] for k in ("height", "width", "depth", "num_objs", "rad_max", "rad_min", "num_seg_classes")], dtype=np.int64)
    out[f"{name}_noise"] = np.array(kw["noise_max"])
img, lab = create_test_image_3d(512, 512, 512, num_objs=40, rad_max=60, rad_min=10, noise_max=0.2, num_seg_classes=4, random_state=np.random.RandomState(0))
assert img.dtype == np.float32 and img.shape == (512, 512, 512)
out["bench_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(img).tobytes()).digest(), dtype=np.uint8)
out["bench_probe_idx"] = np.array([[0, 0, 0], [255, 255, 255], [100, 200, 300], [511, 511, 511], [300, 17, 450]], dtype=np.int64)
out["bench_probe_val"] = np.array([img[tuple(i)] for i in out["bench_probe_idx"]], dtype=np.float32)
out["bench_sum"] = np.array(img.astype(np.float64).sum())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "synthetic.npz"), **out)
print({k: (v.shape, v.dtype) for k, v in out.items()})
