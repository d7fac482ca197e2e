# where do the wave cycles of the composite transposed convolution go (upconv_k4s2_h2_kernel<STATS, RMW>, the headline's form since round 6: 64 windows of 32 ch @ 48^3 -> 96^3)
export TMPDIR=/tmp PMC_UPCONV_FORM=rmw
O=gpurun_out/pmc_uc; rm -rf $O; mkdir -p $O
pass() { n=$1; shift; timeout -k 5 180 rocprofv3 --kernel-trace --pmc "$@" -d $O/p$n -o w -- python tools/pmc_probe.py --only upconv > $O/p$n.log 2>&1; echo "== pass $n: $*" >> $O/stats.txt; find $O/p$n -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%upconv_k4s2_h2%" >> $O/stats.txt 2>&1; }
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
pass 2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass 3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
pass 4 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
grep -v "dispatches     [0-9]  avg *0.0$" $O/stats.txt | cut -c1-150
