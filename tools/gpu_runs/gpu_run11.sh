mkdir -p gpurun_out
export WB_SKIP_DIRECT=1 WB_FIRST=1
for v in variants/*.so; do MH_LIB=$v python tools/wino_bench.py 2>&1 | tail -1; done | tee gpurun_out/wino_bench.log
