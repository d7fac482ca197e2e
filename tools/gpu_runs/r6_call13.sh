# round 6, call 13: linear_h2_big_kernel (two-chunk interval form of tools/experiments/r06_linear_variants.patch, rebuilt in tools/experiments/lin_scratch): requests ONE interval ahead
# against TWO intervals ahead (-DDB_DEPTH2) -- is the kernel waiting on memory latency?
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6c13; mkdir -p $O
cd tools/experiments/lin_scratch
for f in "" "-DDB_DEPTH2" ${EXTRA_VARIANTS}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value $f tools/ubench/linear_ks.hip -o /tmp/lks_${f#-D} 2>>$O/compile.err
done
for rep in 1 2; do for f in "" "-DDB_DEPTH2" ${EXTRA_VARIANTS}; do echo "== variant '${f}' rep $rep"; /tmp/lks_${f#-D} | grep "variant 2"; done; done 2>&1 | tee $O/depth.txt
