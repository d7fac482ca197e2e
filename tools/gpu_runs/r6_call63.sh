# round 6, call 63: composite kernel, operand reads in consumption order (product build) and all eight of a group in the first two gaps (-DUC_DSG=4), against the session's first build
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c63}; mkdir -p $O
for v in prev product dsg4 product dsg4; do
  L=$PWD/monai_amd/csrc/libmonai_amd_$v.so; [ $v = product ] && L=$PWD/monai_amd/csrc/libmonai_amd.so
  MONAI_AMD_LIB=$L timeout 300 python tools/upconv_bench.py > $O/upconv_$v.json 2> $O/upconv_$v.err
  python - <<PY
import json
d=json.load(open("$O/upconv_$v.json")); print("$v", "write", d["write"]["median_ms"], "rmw", d["rmw"]["median_ms"], "rmw_stats", d["rmw_stats"]["median_ms"], d["write_sum"][0], d["rmw_stats_sum"][0])
PY
done | tee $O/upconv_fetch_ab.txt
