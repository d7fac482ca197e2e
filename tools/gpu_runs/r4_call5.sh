# round 4, call 5: host probe for the CPU leg; the z-Winograd kernel's GPU tests and the shape that faulted in call 3; same-box A/B of the direct kernel's new split
export TMPDIR=/tmp
O=gpurun_out/r4c5; mkdir -p $O
timeout 300 python tools/cpu_probe.py > $O/cpu_probe.txt 2>&1; cat $O/cpu_probe.txt
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc tools/ubench/h2z_variants.hip -o /tmp/h2zv 2>/dev/null
for sh in "32 96 64 32" "32 96 16 32" "32 48 64 32"; do set -- $sh; timeout 60 /tmp/h2zv $1 again $2 $3 $4 >> $O/h2z_again.txt 2>&1 || echo "FAILED rc=$? $sh" >> $O/h2z_again.txt; done; cat $O/h2z_again.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -n 0 -k "z_winograd or fp16_split" 2>&1 | tail -4 > $O/gpu_tests_h2z.txt; cat $O/gpu_tests_h2z.txt
timeout 300 python -m pytest tests/test_e2e_gpu.py -q -x -n 0 -k "buffered" 2>&1 | tail -3 > $O/gpu_tests_buffered.txt; cat $O/gpu_tests_buffered.txt
timeout 600 bash tools/gpu_runs/h2_epilogue_ab.sh > $O/h2_split_ab.txt 2>&1; cat $O/h2_split_ab.txt
