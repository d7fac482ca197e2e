# round 6, call 53: upconv_k4s2_h2_kernel with the previous plane's epilogue and the conversion inside the matrix phases: A/B against the previous build on one box, its gpu cases
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c53}; mkdir -p $O
MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_prev.so timeout 300 python tools/upconv_bench.py > $O/upconv_prev.json 2> $O/upconv_prev.err; cat $O/upconv_prev.json
timeout 300 python tools/upconv_bench.py > $O/upconv_new.json 2> $O/upconv_new.err; cat $O/upconv_new.json; tail -3 $O/upconv_new.err
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "upcat or upconv" 2>&1 | tail -4 | tee $O/gpu_tests.txt
