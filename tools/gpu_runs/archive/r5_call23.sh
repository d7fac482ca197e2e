# round 5, call 23: kernel trace of the headline step on the round's last tree (rocprofv3 --kernel-trace --stats, summarised by tools/rocpd_stats.py)
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c23}; mkdir -p $O
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $GRAFT_REPO_ROOT/$O/bench_line_traced.json; cd $GRAFT_REPO_ROOT
find $O/trace -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/bench_kernel_trace_stats.txt 2>&1; head -26 $O/bench_kernel_trace_stats.txt | cut -c1-160
python - <<PY
import json
d = json.load(open("$O/bench_line_traced.json"))
print("traced:", round(d["ms_per_step"], 1), "ms", d["roofline"]["ms_avg"], d["roofline"]["launches"], d["roofline"]["frac"])
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
