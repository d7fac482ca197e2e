# SQ counters of the z-Winograd split-precision kernel next to the direct one (standalone harness: both kernels on the same tensors, 4 launches each):
# three passes of <= 8 counters, kernel trace only; SHAPE="Cin edge windows Cout"
export TMPDIR=/tmp
O=gpurun_out/pmc_h2z; rm -rf $O; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc ${XFLAGS:-} tools/ubench/h2z_variants.hip -o /tmp/h2zv 2>/dev/null || { echo build failed; exit 1; }
set -- ${SHAPE:-32 96 64 32}
timeout 120 /tmp/h2zv $1 plain $2 $3 $4 | tee $O/plain.txt
pass() { n=$1; shift; timeout -k 5 240 rocprofv3 --kernel-trace --pmc "$@" -d $O/p$n -o w -- /tmp/h2zv $SHAPE1 > $O/p$n.log 2>&1; find $O/p$n -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%conv3d_k3_h2%" >> $O/stats.txt 2>&1; }
SHAPE1="$1 pmc $2 $3 $4"
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
pass 2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass 3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM
pass 4 TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
cat $O/stats.txt
