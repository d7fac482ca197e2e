# PMC counters of the fp16 split-precision kernel (32 -> 32 ch, 96^3, 64 windows): two passes of 8 SQ counters
mkdir -p gpurun_out; export TMPDIR=/tmp WB_SKIP_DIRECT=1 WB_SKIP_SPLIT=1 WB_FIRST=1 KB_BATCH=64
rm -rf gpurun_out/pmc_h gpurun_out/pmc_h2 gpurun_out/pmc_h3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pmc_h -o w -- python tools/wino_bench.py > gpurun_out/pmc_w.log 2>&1
find gpurun_out/pmc_h -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%h2_kernel%" > gpurun_out/pmc_h2_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES -d gpurun_out/pmc_h2 -o w -- python tools/wino_bench.py > gpurun_out/pmc_w2.log 2>&1
find gpurun_out/pmc_h2 -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%h2_kernel%" >> gpurun_out/pmc_h2_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum -d gpurun_out/pmc_h3 -o w -- python tools/wino_bench.py > gpurun_out/pmc_w3.log 2>&1
find gpurun_out/pmc_h3 -name "*.db" | head -1 | xargs -I{} python tools/pmc_stats.py {} "%h2_kernel%" >> gpurun_out/pmc_h2_stats.txt 2>&1
find gpurun_out -name "*.db" -delete
cat gpurun_out/pmc_h2_stats.txt; tail -3 gpurun_out/pmc_w.log
