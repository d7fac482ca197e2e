# round 5: where do the wave cycles of the separable resample go -- the barrier-free wave kernel (product library) against the workgroup-staged stream kernel
# (development library built from 7a72e7e), same probe (tools/pmc_probe.py --only resample), SQ + TCC counters in separate passes
export TMPDIR=/tmp
O=gpurun_out/r5pmc_rs; rm -rf $O; mkdir -p $O
pass() { tag=$1; n=$2; shift 2; timeout -k 5 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/$tag$n -o w -- python tools/pmc_probe.py --only resample > $O/$tag$n.log 2>&1; echo "== $tag pass $n: $*" >> $O/stats.txt; find $O/$tag$n -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%separable_resample%" >> $O/stats.txt 2>&1; }
for tag in new old; do
  if [ $tag = old ]; then export MONAI_AMD_LIB=$PWD/monai_amd/csrc/libmonai_amd_dev.so; else unset MONAI_AMD_LIB; fi
  pass $tag 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
  pass $tag 2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
  pass $tag 3 SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
  pass $tag 4 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  pass $tag 5 FETCH_SIZE
  pass $tag 6 WRITE_SIZE
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
grep -v "dispatches     [0-9]  avg *0.0$" $O/stats.txt | cut -c1-160
