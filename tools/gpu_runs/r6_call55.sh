# round 6, call 55: composite kernel with its M block's rows eight staged rows apart (conflict-free operand reads); UpCat's 64-channel convolution as two Winograd launches: test + bench A/B
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c55}; mkdir -p $O
MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_prev.so timeout 300 python tools/upconv_bench.py > $O/upconv_prev.json 2> $O/upconv_prev.err; cat $O/upconv_prev.json
timeout 300 python tools/upconv_bench.py > $O/upconv_new.json 2> $O/upconv_new.err; cat $O/upconv_new.json; tail -3 $O/upconv_new.err
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -x -q -m gpu -k "upcat or upconv or halves" 2>&1 | tail -4 | tee $O/gpu_tests.txt
for i in 1 2; do
for hv in 0 1; do
MONAI_AMD_CONV_HALVES=$hv timeout 600 python bench.py --steps 8 --warmup 3 --cpu-windows 0 --no-extra --no-pmc --no-spread > $O/bench_halves${hv}_$i.json 2> $O/bench_halves${hv}_$i.err
grep -o '"ms_per_step": [0-9.]*' $O/bench_halves${hv}_$i.json | sed "s/^/halves=$hv run $i /"
done; done | tee $O/halves_ab.txt
