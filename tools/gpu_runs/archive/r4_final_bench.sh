# round 4, last call (7 GPU-minutes left): the bench on the final tree with the CPU leg on a 125-window corner instead of the whole volume (the 1000-window leg alone is 176 s;
# the whole-volume line of this round is profiles/r04_bench_line_final.json, measured before the blend's index rewrite -- bit-identical by test -- and the 16-couts kernel,
# which the headline network does not use): final-tree speed, parity, extras (config 3 now on the 16-couts kernel, 27-window parity), hardware-counter traffic in-run
export TMPDIR=/tmp
O=gpurun_out/r4final2; mkdir -p $O
( time timeout 400 python bench.py --cpu-windows 125 > $O/bench_line.json 2> $O/bench_line.err ) 2> $O/bench_line.time; tail -3 $O/bench_line.time
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r4final2/bench_line.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "value", d["value"], "roofline.frac", d["roofline"]["frac"], "hbm", d["roofline_hbm"]["frac"], d["roofline_hbm"].get("frac_of_copy_ceiling"))
    print("traffic", d["roofline"].get("traffic"), (d["roofline"].get("traffic_detail") or {}).get("measured", "")[:60], "|", d["roofline_hbm"].get("traffic"), (d["roofline_hbm"].get("traffic_detail") or {}).get("measured", "")[:60])
    c = d["cpu_baseline"]; print("cpu", c["value"], c["cores"], {k: v for k, v in c["parity_vs_gpu"].items() if k != "compared"}); print(c["sample"][:300])
    for k, v in d.get("extra", {}).items():
        print(k, v.get("ms_per_step"), v.get("parity_vs_cpu_oracle", v.get("parity_vs_cpu_restatement")), json.dumps(v.get("cpu_baseline"))[:300], v.get("error"))
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r4final2/bench_line.err").read()[-3000:])
PY
