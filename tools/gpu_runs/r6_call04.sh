# round 6, call 4: linear_h2_big_kernel with KS chunks per barrier (tools/ubench/linear_ks.hip (tools/experiments/r06_linear_variants.patch)) at UNETR's ViT-B shapes
export TMPDIR=/tmp
O=${O:-gpurun_out/r6c04}; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value tools/ubench/linear_ks.hip (tools/experiments/r06_linear_variants.patch) -o /tmp/linear_ks 2>$O/compile.err && timeout 120 /tmp/linear_ks 2>&1 | tee $O/linear_ks.txt
