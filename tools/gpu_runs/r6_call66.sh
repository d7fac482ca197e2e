# round 6, call 66: the last session's host-level changes of the headline together (UpCat order + the 64-channel convolution as two Winograd launches) against the configuration the session started from, one box
export TMPDIR=/tmp
O=gpurun_out/r6c66; mkdir -p $O
for i in 1 2; do
for cfgname in start final; do
  if [ $cfgname = start ]; then export MONAI_AMD_UPCAT_ORDER=term_first MONAI_AMD_CONV_HALVES=0; else export MONAI_AMD_UPCAT_ORDER=conv_first MONAI_AMD_CONV_HALVES=1; fi
  timeout 600 python bench.py --steps 8 --warmup 3 --cpu-windows 0 --no-extra --no-pmc --no-spread 2>/dev/null | grep '^{' > $O/bench_${cfgname}_$i.json
  echo "$cfgname $i $(grep -o '"ms_per_step": [0-9.]*' $O/bench_${cfgname}_$i.json | head -1)"
done; done | tee $O/ab.txt
