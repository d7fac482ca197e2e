# round 6, call 58: DynUNet's 64-channel concatenation at the top level as two launches of the Winograd split kernel: the 96^3 window case against the oracle, bench --net dynunet A/B
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c58}; mkdir -p $O
timeout 900 python -m pytest tests/test_widen_gpu.py -x -q -m gpu -k "dynunet" 2>&1 | tail -4 | tee $O/gpu_tests.txt
for hv in 0 1 0 1; do
MONAI_AMD_CONV_HALVES=$hv timeout 600 python bench.py --net dynunet --steps 4 --warmup 2 --cpu-windows 27 --no-extra --no-pmc --no-spread > $O/bench_dyn_halves${hv}.json 2> $O/bench_dyn_halves${hv}.err
grep -o '"ms_per_step": [0-9.]*' $O/bench_dyn_halves${hv}.json | head -1 | sed "s/^/dynunet halves=$hv /"
grep -o '"max_abs_logit_diff": [0-9.e-]*' $O/bench_dyn_halves${hv}.json | head -1
done | tee $O/dyn_halves_ab.txt
