# round 6, call 61: UNet (row a11): the down path's joins leave magnitude bounds, its stride-2 convolutions (unit + shortcut) run on the split-precision stride-2 kernel with fused statistics: tests, bench A/B
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c61}; mkdir -p $O
timeout 1200 python -m pytest tests/test_e2e_gpu.py tests/test_widen_gpu.py -x -q -m gpu -k "unet and not unetr and not dynunet and not swin" 2>&1 | tail -5 | tee $O/gpu_tests.txt
for sh in 0 1 0 1; do
MONAI_AMD_STRIDED_H2=$sh timeout 600 python bench.py --net unet --steps 6 --warmup 2 --cpu-windows 27 --no-extra --no-pmc --no-spread > $O/bench_unet_s2h2_${sh}.json 2> $O/bench_unet_s2h2_${sh}.err
grep -o '"ms_per_step": [0-9.]*' $O/bench_unet_s2h2_${sh}.json | head -1 | sed "s/^/unet strided_h2=$sh /"
grep -o '"max_abs_logit_diff": [0-9.e-]*' $O/bench_unet_s2h2_${sh}.json | head -1
done | tee $O/unet_ab.txt
