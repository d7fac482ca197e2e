# round 6, call 60: the accumulating form of the 16-cout split-precision kernel; SegResNet's residual joins as that form (x += conv2, statistics of the sum): cases, nets vs reference / oracle, bench A/B
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c60}; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_widen_gpu.py tests/test_e2e_gpu.py -x -q -m gpu -k "16_couts or segresnet or accumulating or spread" 2>&1 | tail -5 | tee $O/gpu_tests.txt
for ra in 0 1 0 1; do
MONAI_AMD_RESIDUAL_ACC=$ra timeout 600 python bench.py --net segresnet --steps 4 --warmup 2 --cpu-windows 0 --no-extra --no-pmc --no-spread > $O/bench_seg_acc${ra}.json 2> $O/bench_seg_acc${ra}.err
grep -o '"ms_per_step": [0-9.]*' $O/bench_seg_acc${ra}.json | head -1 | sed "s/^/segresnet residual_acc=$ra /"
done | tee $O/seg_acc_ab.txt
