import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F, numpy as np
from monai_amd import ops
dev = "cuda"
cfg = ops.conv3d_k3_h2_config()
for cin, cout, dims in ((16, 32, (4, 16, 16)), (32, 64, (6, 8, 24)), (32, 32, (8, 32, 32))):
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, cin) + dims, generator=g)
    w = torch.randn((cout, cin, 3, 3, 3), generator=g) / np.sqrt(27 * cin)
    packed = ops.conv3d_k3_pack(cfg, w.to(dev))
    exp = F.conv3d(x.double(), w.double(), None, padding=1)
    for rep in range(3):
        out = torch.full((1, cout) + dims, float('nan'), device=dev)
        ops.conv3d_k3(cfg, x.to(dev), None, packed, None, out, None)
        torch.cuda.synchronize()
        d = (out.cpu().double() - exp).abs()
        print(cin, cout, dims, "rep", rep, "max err", d.max().item(), "nan", torch.isnan(out).sum().item(), "frac>1e-4", (d > 1e-4).double().mean().item())
    print(" tail", packed[-4:].cpu())
    print(" err by z", d.amax(dim=(0, 1, 3, 4)).numpy().round(6))
    print(" err by y", d.amax(dim=(0, 1, 2, 4)).numpy().round(6))
    print(" err by x", d.amax(dim=(0, 1, 2, 3)).numpy().round(6))
    print(" err by co", d.amax(dim=(0, 2, 3, 4)).numpy().round(6))
    # packed weights vs host split
    s = float(packed[-3])
    hw = (w * s).half()
    lw = ((w * s) - hw.float()).half()
    pk = packed[:-4].view(torch.float16).cpu().reshape(cout // 32, cin // 16, 2, 27, 2, 32, 8)
    ref = torch.stack([hw, lw])  # [piece][co][ci][27]
    ref = ref.reshape(2, cout // 32, 32, cin // 16, 2, 8, 27).permute(1, 3, 0, 6, 4, 2, 5)
    print(" packed == host split:", torch.equal(pk, ref.contiguous()))
