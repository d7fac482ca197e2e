# PMC counters of the Winograd split-precision kernel (standalone harness, 32 -> 32 ch, 96^3, 64 windows): two passes of 8 SQ counters
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc_h2w; rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -Iinclude -Imonai_amd/csrc tools/ubench/h2w_variants.hip -o /tmp/h2wv
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $O/p1 -o w -- /tmp/h2wv pmc > $O/run1.log 2>&1
find $O/p1 -name "*.db" | head -1 | xargs -I{} python $R/tools/pmc_stats.py {} "%h2w_kernel%" > $O/stats.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d $O/p2 -o w -- /tmp/h2wv pmc > $O/run2.log 2>&1
find $O/p2 -name "*.db" | head -1 | xargs -I{} python $R/tools/pmc_stats.py {} "%h2w_kernel%" >> $O/stats.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/p1 $O/p2
cat $O/stats.txt; tail -2 $O/run1.log
