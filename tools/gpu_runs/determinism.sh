# run-to-run determinism stress of the headline path: concurrent processes on ONE GPU, every launch digested, first differing kernel named;
# then the same without the digest hooks but with poisoned allocator blocks and fresh activation buffers per run; then the poison check of tools/poison_check.py
O=gpurun_out/determinism; mkdir -p $O; export TMPDIR=/tmp
RUNS=${RUNS:-12}; PROCS=${PROCS:-4}
timeout 900 python tools/determinism_stress.py --procs $PROCS --runs $RUNS --size 512 2>&1 | grep "^STRESS " > $O/stress_hooks.json; cut -c1-600 $O/stress_hooks.json
timeout 900 python tools/determinism_stress.py --procs $PROCS --runs $RUNS --size 512 --no-hooks --poison --fresh-plans 2>&1 | grep "^STRESS " > $O/stress_poison.json; cut -c1-600 $O/stress_poison.json
for algo in h2 fp32; do
  MONAI_AMD_CONV_ALGO=$algo timeout 600 python tools/poison_check.py --device cuda --size 192 192 192 --roi 96 96 96 --patterns zero nan huge neg 2>&1 | grep -v "^BasicUNet" > $O/poison_$algo.txt; tail -2 $O/poison_$algo.txt
done
