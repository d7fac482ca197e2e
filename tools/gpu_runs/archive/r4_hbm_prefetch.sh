# round 4: two source planes ahead in the streaming resample / Gaussian kernels: previous build vs this tree, alternating processes on one box
export TMPDIR=/tmp
O=gpurun_out/r4hbm; mkdir -p $O
: > $O/ab.jsonl
for i in 1 2; do
  MONAI_AMD_LIB=$PWD/tools/ubench/_old/libmonai_amd_prev.so timeout 200 python tools/hbm_prefetch_ab.py >> $O/ab.jsonl 2>> $O/err.txt
  timeout 200 python tools/hbm_prefetch_ab.py >> $O/ab.jsonl 2>> $O/err.txt
done
cat $O/ab.jsonl | cut -c1-900
