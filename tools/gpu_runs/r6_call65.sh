# round 6, call 65: conv1x1_windows_kernel with 8 instead of 4 channel loads in flight per thread (-DMH_C1W_CB=8): the headline's kernel trace with either build, one box
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6c65; mkdir -p $O
for v in product cb8 product cb8; do
  L=$PWD/monai_amd/csrc/libmonai_amd_$v.so; [ $v = product ] && L=$PWD/monai_amd/csrc/libmonai_amd.so
  ( cd /tmp && MONAI_AMD_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o bench -- python $OLDPWD/bench.py --steps 4 --warmup 2 --cpu-windows 0 --no-extra --no-pmc --no-spread ) > $O/bench_$v.json 2> $O/trace_$v.err
  find $O/prof_$v -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/stats_$v.txt 2>&1; rm -rf $O/prof_$v
  echo "$v $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$v.json | head -1) $(grep conv1x1_windows $O/stats_$v.txt | awk '{print "conv1x1_windows avg_us", $(NF-9), "calls", $(NF-11)}')"
  grep conv1x1_windows $O/stats_$v.txt | cut -c100-200
done | tee $O/cb_ab.txt
