# round 4, call 6: host probe v2 (thread x process layouts under the CPU quota), schedule-invariance of the 512^3 inference (one process: the forced schedules need the memory),
# SQ / TA counters of the z-Winograd kernel next to the direct one
export TMPDIR=/tmp
O=gpurun_out/r4c6; mkdir -p $O
timeout 420 python tools/cpu_probe.py > $O/cpu_probe_v2.txt 2>&1; cat $O/cpu_probe_v2.txt
timeout 600 python tools/determinism_stress.py --procs 1 --runs 8 --size 512 --vary 2>&1 | grep "^STRESS " > $O/stress_vary.json; cut -c1-900 $O/stress_vary.json
SHAPE="64 96 64 32" timeout 1100 bash tools/gpu_runs/pmc_h2z.sh > $O/pmc_h2z.txt 2>&1; tail -80 $O/pmc_h2z.txt
