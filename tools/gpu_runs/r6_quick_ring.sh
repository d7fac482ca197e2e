export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Imonai_amd/csrc -DRSV_RING=3 tools/ubench/resample_variants.hip -o /tmp/rsv3 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Imonai_amd/csrc tools/ubench/resample_variants.hip -o /tmp/rsv0 2>/dev/null
for c in 26 26; do /tmp/rsv3 "fp64 ring3 (offset from the generic pointer) chunks=$c" $c; /tmp/rsv0 "fp64 regs chunks=11" 11; done
