"""Build profiles/rNN_pmc_hbm_traffic.{json,txt} from the two per-kernel PMC tables tools/gpu_runs/pmc.sh leaves
(`pmc_fetch_stats.txt`, `pmc_write_stats.txt`: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over
tools/pmc_probe.py).  Counter unit: KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
tallies the 128-B requests of a wide (16 B/lane) coalesced read as 64 B, so the kernels listed in WIDE_READS have their
FETCH doubled; nothing else is corrected.

usage: python tools/pmc_traffic_json.py gpurun_out/pmc_fetch_stats.txt gpurun_out/pmc_write_stats.txt profiles/r02_pmc_hbm_traffic"""
import json
import re
import sys

VOL = 512 ** 3 * 4.0
RES_OUT = 410 * 410 * 819 * 4.0
CONV = 2 * 64 * 32 * 96 ** 3 * 4.0                  # input + output of 64 windows, 32 channels
BLEND = 1000 * 5 * 96 ** 3 * 4.0 + 5 * VOL          # every window's logits once + the blended volume once (tools/pmc_probe.py)
ALGORITHMIC = {                                     # key (substring of the kernel name) -> algorithmic bytes per launch
    "affine_resample_kernel<double": VOL + RES_OUT, "affine_resample_kernel<float": VOL + RES_OUT,
    "separable_resample_stream_kernel<double": VOL + RES_OUT, "separable_resample_stream_kernel<float": VOL + RES_OUT,
    "gauss3d_stream_kernel": 2 * VOL, "gauss3d_rowvec_kernel": 2 * VOL, "gauss3d_rowdpp_kernel": 2 * VOL,
    "conv3d_k3_mfma_kernel": CONV, "conv3d_k3_wino2d_kernel": CONV, "conv3d_k3_wino2p_kernel": CONV, "conv3d_k3_wino2s_kernel": CONV, "conv3d_k3_h2_kernel": CONV,
    "sw_blend_kernel": BLEND, "sw_blend_reg_kernel": BLEND, "sw_blend_mosaic_kernel": BLEND,
}
WIDE_READS = ("sw_blend_kernel", "sw_blend_reg_kernel", "sw_blend_mosaic_kernel", "gauss3d_rowvec_kernel", "gauss3d_rowdpp_kernel")


def table(path, counter):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+dispatches\s+(\d+)\s+avg\s+([\d.]+)", line)
            if m and m.group(1) == counter and name:
                out[name] = float(m.group(3)) * 1024.0
    return out


def main(fetch_txt, write_txt, stem):
    fetch, write = table(fetch_txt, "FETCH_SIZE"), table(write_txt, "WRITE_SIZE")
    kernels, rows = {}, []
    for name in sorted(set(fetch) & set(write)):
        key = next((k for k in sorted(ALGORITHMIC, key=len, reverse=True) if k in name), None)
        if key is None:
            continue
        wide = any(w + "<" in name or w + "(" in name for w in WIDE_READS)
        f = fetch[name] * (2.0 if wide else 1.0)
        short = key.split("<")[0] if key.split("<")[0] not in kernels else key
        kernels[short] = {"kernel": name[:130], "fetch_bytes": fetch[name], "write_bytes": write[name], "fetch_corrected_x2": wide,
                          "hbm_bytes_per_launch": f + write[name], "algorithmic_bytes": ALGORITHMIC[key]}
        rows.append(f"{name[:100]:100s} {fetch[name] / 1e6:10.1f} {write[name] / 1e6:10.1f} {ALGORITHMIC[key] / 1e6:12.1f} "
                    f"{(f + write[name]) / ALGORITHMIC[key]:8.3f}{'   (FETCH x2)' if wide else ''}")
    src = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python tools/pmc_probe.py; KiB per dispatch averaged "
           "over 3 dispatches; FETCH doubled for the 16 B/lane streaming readers (gfx950 counts their 128-B requests as 64 B)")
    with open(stem + ".json", "w") as f:
        json.dump({"source": src, "kernels": kernels}, f, indent=1)
    with open(stem + ".txt", "w") as f:
        f.write(src + "\n\n" + f"{'kernel':100s} {'FETCH MB':>10s} {'WRITE MB':>10s} {'algorithm MB':>12s} {'HBM/alg':>8s}\n" + "\n".join(rows) + "\n")
    print("\n".join(rows))


if __name__ == "__main__":
    main(*sys.argv[1:4])
