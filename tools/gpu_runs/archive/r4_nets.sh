# round 4: the other networks through the same inferer and volume (record for DESIGN_HISTORY 6.0), and the bench line with the extras (no CPU leg) after the last bench.py change
export TMPDIR=/tmp
O=gpurun_out/r4nets; mkdir -p $O
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-windows 0 > $O/bench_extras.json 2> $O/bench_extras.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4nets/bench_extras.json").read().strip().splitlines()[-1])
    print("ms", d["ms_per_step"], "hbm", d["roofline_hbm"]["frac"], d["roofline_hbm"].get("device_copy_GBps"), d["roofline_hbm"].get("frac_of_copy_ceiling"))
    e = d["extra"]
    print("config4 copy", e["config4"].get("device_copy_GBps"), [(round(r["frac"], 3), round(r["frac_of_copy_ceiling"], 3)) for r in e["config4"]["runs"]], e["config4"].get("cpu_baseline"), e["config4"].get("error"))
    print("config3", e["config3"].get("ms_per_step"), e["config3"].get("error"), "fp32", e["fp32_exact"].get("ms_per_step"), e["fp32_exact"].get("error"))
except Exception as ex:
    print("bench failed", ex); print(open("gpurun_out/r4nets/bench_extras.err").read()[-2500:])
PY
for net in unet segresnet dynunet swinunetr; do
  timeout 400 python bench.py --net $net --steps 2 --warmup 1 --cpu-windows 0 --no-extra > $O/bench_$net.json 2> $O/bench_$net.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r4nets/bench_$net.json").read().strip().splitlines()[-1])
    print("$net", round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvoxel/s")
except Exception as e:
    print("$net failed", e, open("gpurun_out/r4nets/bench_$net.err").read()[-600:])
PY
done
