# What bounds the k2 s2 transposed convolution (3.0 TB/s written, arithmetic is not it -- profiles/r02_deconv_bench_v1.json)?  HBM bytes fetched and written per
# launch of both forms (separate --pmc passes, kernel trace only): fetched >> 0.9 GB algorithmic means the stores pull their lines in first (write-allocate of
# partial lines), written >> 7.25 GB means write amplification; neither means the limit is in front of the memory system (issue, TLB, stream count).
O=gpurun_out/pmc_deconv; mkdir -p $O; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/$c
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p -- python tools/deconv_bench.py > $O/$c.log 2>&1
  find $O/$c -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%deconv%" > $O/${c}_stats.txt 2>&1
  find $O/$c -name "*.db" -delete
  cut -c1-150 $O/${c}_stats.txt
done
