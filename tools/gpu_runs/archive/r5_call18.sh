# round 5, call 18: the transposed convolution k2 s2 on the matrix cores (kernels/deconv_h2.h) -- its cases, per-shape timing against the vector-ALU kernel, the headline and
# DynUNet with and without it on one box; DynUNet's deferred skip tensor; the 384-channel concat convolution of SwinUNETR as two accumulating halves
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c18}; mkdir -p $O
timeout 500 python -m pytest tests/test_kernels_gpu.py tests/test_widen_gpu.py tests/test_e2e_gpu.py -q -m gpu -x -k "transposed_convolution_on_matrix or strided or dynunet or segresnet or swin_unetr_vs or net_single_window or sliding_window_net5 or config0 or upcat or odd_window or test_deconv" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
timeout 200 python tools/deconv_h2_bench.py 2>&1 | grep "^{" | tee $O/deconv_h2_layers.txt
line() { python - "$1" "$2" <<PY
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "checksum", d["checksum"])
PY
}
for f in 0 1 0 1; do
  MONAI_AMD_DECONV_H2=$f timeout 200 python bench.py --steps 3 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_basicunet_dc_$f.json
  line $O/bench_basicunet_dc_$f.json "basicunet DECONV_H2=$f"
done
for f in 0 1; do
  MONAI_AMD_DECONV_H2=$f timeout 200 python bench.py --net dynunet --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_dynunet_dc_$f.json
  line $O/bench_dynunet_dc_$f.json "dynunet DECONV_H2=$f"
done
timeout 300 python bench.py --net swinunetr --steps 1 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_swinunetr.json
line $O/bench_swinunetr.json "swinunetr"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --net dynunet --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find $O/trace -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/dynunet_kernel_trace_stats.txt 2>&1; head -16 $O/dynunet_kernel_trace_stats.txt | cut -c1-150
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
