# round 5: SQ counters of the stride-2 split-precision GEMM (conv3d_k3s2_h2_kernel<2, true, false>, split form) on DynUNet's 32 -> 64 @ 96^3 -> 48^3 layer, 64 windows:
# is it the LDS that bounds it (DESIGN_HISTORY 4.1b)?  Counters in their own passes with --kernel-trace only.
export TMPDIR=/tmp
O=gpurun_out/r5pmc_s2; rm -rf $O; mkdir -p $O
pass() { n=$1; shift; timeout -k 5 150 rocprofv3 --kernel-trace --pmc "$@" -d $O/p$n -o w -- python tools/s2_bench.py --layers "32,64,96" --reps 2 > $O/p$n.log 2>&1; echo "== pass $n: $*" >> $O/stats.txt; find $O/p$n -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%conv3d_k3s2_h2_kernel%" >> $O/stats.txt 2>&1; }
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
pass 2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES
pass 3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
cat $O/stats.txt | cut -c1-150
