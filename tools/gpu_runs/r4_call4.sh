# round 4, call 4: (1) which of the two instruction-count changes of the z-Winograd kernel faults on the GPU (each variant under its own timeout);
# (2) the driver's bench command with the whole-volume parity leg
export TMPDIR=/tmp
O=gpurun_out/h2zv; mkdir -p $O; : > $O/bisect.txt
for v in "" "-DHZX_OLD_ACT" "-DHZX_OLD_SPLIT" "-DHZX_OLD_ACT -DHZX_OLD_SPLIT"; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc $v tools/ubench/h2z_variants.hip -o /tmp/h2zv 2>/dev/null || { echo "build failed: $v" >> $O/bisect.txt; continue; }
  for sh in "32 96 8 32" "64 96 64 32"; do
    set -- $sh
    timeout 90 /tmp/h2zv $1 "${v:-default}" $2 $3 $4 >> $O/bisect.txt 2>&1 || echo "FAILED rc=$? : ${v:-default} $sh" >> $O/bisect.txt
  done
done
cat $O/bisect.txt
O=gpurun_out/bench; mkdir -p $O
( time timeout 1700 python bench.py > $O/line_r4_v1.json 2> $O/line_r4_v1.err ) 2> $O/line_r4_v1.time
tail -3 $O/line_r4_v1.time; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench/line_r4_v1.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"])
    print("cpu_baseline", {k: v for k, v in d["cpu_baseline"].items() if k != "sample"})
    print("sample", d["cpu_baseline"]["sample"])
    for k, v in d.get("extra", {}).items():
        print(k, json.dumps(v)[:1500])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/bench/line_r4_v1.err").read()[-3000:])
PY
