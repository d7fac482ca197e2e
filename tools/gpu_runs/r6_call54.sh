# round 6, call 54: UpCat order A/B (convolution first + composite term with its statistics in place | composite term first + accumulating convolution) with the pipelined composite kernel, one box
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c54}; mkdir -p $O
for i in 1 2; do
for ord in term_first conv_first; do
MONAI_AMD_UPCAT_ORDER=$ord timeout 600 python bench.py --steps 8 --warmup 3 --cpu-windows 0 --no-extra --no-pmc --no-spread > $O/bench_${ord}_$i.json 2> $O/bench_${ord}_$i.err
python - <<PY
import json
d=json.load(open("$O/bench_${ord}_$i.json")); print("$ord", $i, d["ms_per_step"], d["value"])
PY
done; done | tee $O/order_ab.txt
timeout 1200 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee $O/gpu_tests.txt
