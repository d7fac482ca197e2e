# Gaussian smoothing kernels A/B with the development library (tools/gauss_ab.py); writes gpurun_out/gauss/gauss_ab.json
O=gpurun_out/gauss; mkdir -p $O
MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so timeout 300 python tools/gauss_ab.py > $O/gauss_ab.txt 2> $O/err.txt; tail -1 $O/gauss_ab.txt > $O/gauss_ab.json; grep -v '^{"volume' $O/gauss_ab.txt; tail -3 $O/err.txt
