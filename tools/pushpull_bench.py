"""Development aid: time the monai._C resampling entry points (monai_amd._C) on one 256^3 volume with a smooth random
deformation, per interpolation order.  Algorithmic bytes per target voxel: grid 3 x 4 B + one value read + one written
(pull: 20 B; push: 20 B + the scattered read-modify-writes; grad: + 3 x 4 B written)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monai_amd import _C  # noqa: E402

dev = torch.device("cuda")
E = int(os.environ.get("PB_EDGE", "256"))


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


torch.manual_seed(0)
vol = torch.rand(1, 1, E, E, E, device=dev)
ax = torch.arange(E, device=dev, dtype=torch.float32)
ident = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1)[None]
coarse = torch.randn(1, 3, 8, 8, 8, device=dev) * 3.0
disp = torch.nn.functional.interpolate(coarse, size=(E, E, E), mode="trilinear", align_corners=True).permute(0, 2, 3, 4, 1)
grid = (ident + disp).contiguous()
gridg = grid.clone().requires_grad_(True)
volg = vol.clone().requires_grad_(True)
cot = torch.rand(1, 1, E, E, E, device=dev)
bd = [_C.BoundType.dct2]
rows = []
n = E ** 3
for order in (0, 1, 2, 3):
    it = [_C.InterpolationType(order)]
    r = {"order": order}
    ms = timeit(lambda: _C.grid_pull(vol, grid, bd, it, True))
    r["pull_ms"], r["pull_GBps"] = round(ms, 3), round(20.0 * n / ms / 1e6, 1)
    ms = timeit(lambda: _C.grid_push(cot, grid, [E, E, E], bd, it, True))
    r["push_ms"], r["push_GBps"] = round(ms, 3), round(20.0 * n / ms / 1e6, 1)
    ms = timeit(lambda: _C.grid_count(grid, [E, E, E], bd, it, True))
    r["count_ms"] = round(ms, 3)
    ms = timeit(lambda: _C.grid_grad(vol, grid, bd, it, True))
    r["grad_ms"], r["grad_GBps"] = round(ms, 3), round(28.0 * n / ms / 1e6, 1)
    ms = timeit(lambda: _C.grid_pull_backward(cot, volg, gridg, bd, it, True))
    r["pull_backward_ms"] = round(ms, 3)
    rows.append(r)
print(json.dumps({"edge": E, "dtype": "f32", "bound": "dct2", "rows": rows}))
