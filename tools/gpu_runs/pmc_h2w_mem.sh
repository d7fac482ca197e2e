# memory-side counters of the Winograd split-precision kernel (standalone harness): L2 hits / misses, HBM read / write requests, FETCH_SIZE / WRITE_SIZE in separate passes
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc_h2w_mem; rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -Iinclude -Imonai_amd/csrc tools/ubench/h2w_variants.hip -o /tmp/h2wv
R=$PWD
cd /tmp
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  rm -rf $O/p; rocprofv3 --kernel-trace --pmc $C -d $O/p -o w -- /tmp/h2wv pmc > $O/run.log 2>&1
  find $O/p -name "*.db" | head -1 | xargs -I{} python $R/tools/pmc_stats.py {} "%h2w_kernel%" >> $O/stats.txt 2>&1
done
rm -rf $O/p; cat $O/stats.txt; tail -2 $O/run.log
