export WB_SKIP_DIRECT=1 KB_BATCH=64 WB_FIRST=1
python tools/wino_bench.py 2>&1 | tail -1 | cut -c1-260
for v in variants/*.so; do MH_LIB=$v python tools/wino_bench.py 2>&1 | tail -1 | cut -c1-260; done
