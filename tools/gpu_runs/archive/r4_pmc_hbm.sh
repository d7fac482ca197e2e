# round 4: SQ / TA counters of the z-marching HBM kernels (fused Gaussian, separable resample) and of the blend, per kernel: where do the wave cycles go
export TMPDIR=/tmp
O=gpurun_out/r4pmc_hbm; rm -rf $O; mkdir -p $O
pass() { n=$1; shift; timeout -k 5 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/p$n -o w -- python tools/pmc_probe.py --only mosaic,resample,gaussian > $O/p$n.log 2>&1; find $O/p$n -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%mh::%" >> $O/stats.txt 2>&1; }
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
pass 2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass 3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
pass 4 TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
grep -v "^    .*dispatches     [0-9]  avg *0.0$" $O/stats.txt | grep -A9 "gauss3d_rowdpp\|separable_resample_stream_kernel<float\|sw_blend_mosaic" | cut -c1-150
