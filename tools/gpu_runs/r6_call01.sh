# round 6, call 1: (a) the RCCL path on one rank: the -m gpu test and bench.py --force-shard next to the plain line on the same box;
# (b) is rocprofv3 --att usable on this pool (no decoder library in the image)?
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c01}; mkdir -p $O
timeout 600 python -m pytest tests/test_e2e_gpu.py -q -m gpu -x -n 0 -s -k "rccl_one_rank" 2>&1 | tail -8 | tee $O/rccl_test.txt
timeout 600 python bench.py --steps 5 --warmup 2 --cpu-windows 0 --no-extra --no-pmc 2>$O/bench_plain.err | grep "^{" > $O/bench_plain.json
timeout 600 python bench.py --force-shard --steps 5 --warmup 2 2>$O/bench_forced.err | grep "^{" > $O/bench_forced.json
MONAI_AMD_TAIL_ROUNDS=0 timeout 600 python bench.py --force-shard --steps 5 --warmup 2 2>/dev/null | grep "^{" > $O/bench_forced_notail.json
python - <<PY
import json
for n in ("plain", "forced", "forced_notail"):
    try:
        d = json.load(open("$O/bench_%s.json" % n))
        print(n, round(d["ms_per_step"], 2), "ms", d.get("per_rank_ms_per_step"), d.get("exposed_comm_share"), d.get("forced_shard"), d["checksum"])
    except Exception as e:
        print(n, "failed", e)
PY
( cd /tmp && timeout 300 rocprofv3 --att --kernel-trace -d $OLDPWD/$O/att -- python $OLDPWD/tools/pmc_probe.py --only resample ) > $O/att_probe.log 2>&1; echo "att rc=$?" | tee -a $O/att_probe.log
tail -5 $O/att_probe.log; find $O/att -type f | head -20; find $O/att -size +2M -delete
