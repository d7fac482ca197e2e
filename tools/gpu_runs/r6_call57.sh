# round 6, call 57: what the pooling form of the Winograd split kernel pays for (8.1 against 7.07 ms): the product build, a build whose pooled stores go beyond the buffer
# (-DHWX_OFF=64), a build without the pooling arithmetic and stores (-DHWX_OFF=32); tools/h2w_bench.py at 96^3, one box
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c57}; mkdir -p $O
for v in product x64 x32 product; do
  L=$PWD/monai_amd/csrc/libmonai_amd_$v.so; [ $v = product ] && L=$PWD/monai_amd/csrc/libmonai_amd.so
  MONAI_AMD_LIB=$L KB_EDGES=96 KB_ITERS=12 timeout 600 python tools/h2w_bench.py > $O/h2w_$v.json 2> $O/h2w_$v.err
  python - <<PY
import json
d=json.load(open("$O/h2w_$v.json"))["cases"][0]["h2w"]
print("$v", "plain", d["plain"]["median_ms"], "acc", d["acc"]["median_ms"], "pool", d["pool"]["median_ms"], d["pool"].get("pool_bitwise"), d.get("pool_bitwise"))
PY
done | tee $O/pool_ablation.txt
