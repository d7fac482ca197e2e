# round 4, call 8: the four-wave / 512-register form of the z-Winograd kernel against the eight-wave form and the direct kernel
export TMPDIR=/tmp
O=gpurun_out/r4c8; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc ${XFLAGS:-} tools/ubench/h2z_variants.hip -o /tmp/h2zv 2>$O/build.err || { echo build failed; tail -5 $O/build.err; exit 1; }
for sh in "32 96 64 32" "64 96 64 32" "32 48 64 32" "64 24 64 64"; do set -- $sh; timeout 90 /tmp/h2zv $1 h2zw $2 $3 $4 >> $O/h2zw.txt 2>&1 || echo "FAILED rc=$? $sh" >> $O/h2zw.txt; done; cat $O/h2zw.txt
