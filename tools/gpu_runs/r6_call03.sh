# round 6, call 3: do the kernels of two streams overlap at all?  kernel trace of two headline steps at SW_STREAMS = 2 and 1 -> tools/trace_overlap.py
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c03}; mkdir -p $O
for k in 2 1; do
  ( cd /tmp && MONAI_AMD_SW_STREAMS=$k timeout 400 rocprofv3 --kernel-trace -d $OLDPWD/$O/prof$k -o t -- python $OLDPWD/bench.py --steps 2 --warmup 1 --cpu-windows 0 --no-extra --no-pmc ) > $O/traced_$k.json 2> $O/trace_$k.err
  find $O/prof$k -name "*.db" | head -1 | xargs -I{} python tools/trace_overlap.py {} 2>&1 | cut -c1-300 | tee $O/overlap_$k.txt
  rm -rf $O/prof$k
done
