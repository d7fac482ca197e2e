mkdir -p gpurun_out
export TMPDIR=/tmp WB_SKIP_DIRECT=1 MH_LIB=variants/lib_v11.so








