# kernel traces of SwinUNETR and UNETR through the bench (one step each)
export TMPDIR=/tmp
O=gpurun_out/r4prof; mkdir -p $O
for net in swinunetr unetr; do
  timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_$net -o b -- python bench.py --net $net --steps 1 --warmup 1 --cpu-windows 0 --no-extra > $O/$net.json 2> $O/$net.err
  find $O/prof_$net -name "*.db" | head -1 | xargs -I{} python tools/rocpd_stats.py {} > $O/${net}_kernel_trace_stats.txt 2>&1; rm -rf $O/prof_$net
  head -30 $O/${net}_kernel_trace_stats.txt | cut -c1-170
done
