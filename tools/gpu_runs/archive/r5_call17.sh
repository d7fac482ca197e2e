# round 5, call 17: the stride-2 convolution on the matrix cores (kernels/conv3d_s2_h2.h) -- its cases, per-layer timing against the vector-ALU kernel, DynUNet / SegResNet
# with and without it on one box, DynUNet's kernel trace (also: the 512-channel concat convolution as two accumulating halves)
export TMPDIR=/tmp
O=${O:-gpurun_out/r5c17}; mkdir -p $O
timeout 400 python -m pytest tests/test_kernels_gpu.py tests/test_widen_gpu.py -q -m gpu -x -k "strided or dynunet or segresnet" 2>&1 | tail -4 | tee $O/gpu_tests_subset.txt
timeout 200 python tools/s2_bench.py 2>&1 | tee $O/s2_layers.txt
line() { python - "$1" "$2" <<PY
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], round(d["ms_per_step"], 1), "ms", round(d["value"] / 1e6, 1), "Mvox/s", "checksum", d["checksum"], "parity", d.get("parity"))
PY
}
for net in dynunet segresnet; do
for f in 0 1; do
  MONAI_AMD_STRIDED_H2=$f timeout 200 python bench.py --net $net --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_${net}_s2_$f.json
  line $O/bench_${net}_s2_$f.json "$net STRIDED_H2=$f"
done
done
timeout 300 python bench.py --net dynunet --steps 2 --warmup 1 --cpu-windows 27 --no-extra --no-pmc 2>/dev/null | grep "^{" > $O/bench_dynunet_parity.json
line $O/bench_dynunet_parity.json "dynunet (27-window parity)"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --net dynunet --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find $O/trace -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/dynunet_kernel_trace_stats.txt 2>&1; head -24 $O/dynunet_kernel_trace_stats.txt | cut -c1-150
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
