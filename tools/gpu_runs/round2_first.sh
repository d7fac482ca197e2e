# First GPU call of the next round: everything that was added after round 1's GPU budget was spent.
#   gpurun --timeout 1500 -- 'bash tools/gpu_runs/round2_first.sh'
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/widen_gpu_tests.txt      # the whole GPU suite; tests/test_widen_gpu.py (never run on hardware in round 1) sorts last
python tools/preproc_bench.py > gpurun_out/preproc_bench.json 2> gpurun_out/preproc_bench.err
python bench.py --net dynunet --steps 2 --warmup 1 --cpu-windows 0 > gpurun_out/bench_dynunet.json 2> gpurun_out/bench_dynunet.err
python bench.py --net segresnet --steps 2 --warmup 1 --cpu-windows 0 > gpurun_out/bench_segresnet.json 2> gpurun_out/bench_segresnet.err
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
# kernel trace of the pre-processing bench (copy the *_kernel_stats.csv summary into profiles/ afterwards)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_preproc -- python $GRAFT_REPO_ROOT/tools/preproc_bench.py > /dev/null 2>&1 )
tail -5 gpurun_out/widen_gpu_tests.txt; cut -c1-400 gpurun_out/preproc_bench.json gpurun_out/bench_dynunet.json gpurun_out/bench_segresnet.json gpurun_out/bench_line.json
