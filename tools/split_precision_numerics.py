"""Feasibility study for DESIGN_HISTORY.md section 8 item 1 (no GPU needed): run the oracle BasicUNet with every 3x3x3
convolution evaluated in bf16 split-precision arithmetic -- both operands split into hi + mid + lo bf16 pieces, the
six largest piece products accumulated in fp32 -- and compare logits / argmax with the plain fp32 forward.
Products of two bf16 values are exact in fp32, so F.conv3d on the pieces (fp32 accumulation) is the arithmetic a
v_mfma_f32_16x16x32_bf16 pipeline would perform, up to summation order."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.basic_unet as ob  # noqa: E402


def split3(t):
    hi = t.to(torch.bfloat16).to(torch.float32)
    r = t - hi
    mid = r.to(torch.bfloat16).to(torch.float32)
    lo = (r - mid).to(torch.bfloat16).to(torch.float32)
    return hi, mid, lo


def make_conv(n_terms):
    real = F.conv3d

    def conv(x, w, b=None, stride=1, padding=0, *a, **k):
        if w.shape[-1] != 3:
            return real(x, w, b, stride, padding, *a, **k)
        xs, ws = split3(x), split3(w)
        pairs = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1), (1, 2), (2, 1), (2, 2)][:n_terms]
        acc = None
        for i, j in reversed(pairs):          # small terms first
            y = real(xs[i], ws[j], None, stride, padding)
            acc = y if acc is None else acc + y
        return acc if b is None else acc + b.reshape(1, -1, 1, 1, 1)

    return conv


def main():
    torch.manual_seed(1)
    sd = ob.make_basic_unet_state(1, 5)
    torch.manual_seed(23)
    x = torch.rand(2, 1, 64, 64, 64)
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // 2))
    with torch.no_grad():
        ref = ob.basic_unet_forward(sd, x)
        rows = []
        for n_terms in (1, 3, 6, 9):
            real = F.conv3d
            ob.F.conv3d = make_conv(n_terms)
            try:
                y = ob.basic_unet_forward(sd, x)
            finally:
                ob.F.conv3d = real
            d = (y - ref).abs()
            mism = int((y.argmax(1) != ref.argmax(1)).sum())
            top2 = ref.topk(2, dim=1).values
            margin = (top2[:, 0] - top2[:, 1])
            mm = float(margin[(y.argmax(1) != ref.argmax(1))].max()) if mism else 0.0
            rows.append({"piece_products": n_terms, "max_abs_logit_diff": float(d.max()), "mean_abs_logit_diff": float(d.mean()),
                         "argmax_mismatch_voxels": mism, "voxels": int(ref[:, 0].numel()), "max_top2_margin_at_mismatch": mm})
            print(rows[-1], flush=True)
    print(json.dumps({"net": "BasicUNet 1->5, seed 1, 2 x 64^3 windows", "logit_scale": float(ref.abs().max()), "rows": rows}))


if __name__ == "__main__":
    main()
