# round 6, call 10: the LDS-DMA ring form of the separable resample (RSV_RING=3) against the register-prefetch form, standalone harness, with z-chunk sweeps
export TMPDIR=/tmp
O=gpurun_out/r6c10; mkdir -p $O; : > $O/ring.txt
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Imonai_amd/csrc $1 tools/ubench/resample_variants.hip -o $2 2>>$O/compile.err; }
build "" /tmp/rsv_d0; build "-DRSV_RING=3" /tmp/rsv_d3; build "-DRSV_T=float -DRSV_NL=4 -DRSV_NT=512" /tmp/rsv_f0; build "-DRSV_T=float -DRSV_NL=4 -DRSV_NT=512 -DRSV_RING=3" /tmp/rsv_f3
build "-DRSV_NL=4 -DRSV_NT=512 -DRSV_RING=3" /tmp/rsv_d3w
for rep in 1 2; do
  for c in 0 11 21 26 41 52; do
    /tmp/rsv_d0 "fp64 256 regs  chunks=$c" $c >> $O/ring.txt; /tmp/rsv_d3 "fp64 256 ring3 chunks=$c" $c >> $O/ring.txt; /tmp/rsv_d3w "fp64 512 ring3 chunks=$c" $c >> $O/ring.txt
    /tmp/rsv_f0 "fp32 512 regs  chunks=$c" $c >> $O/ring.txt; /tmp/rsv_f3 "fp32 512 ring3 chunks=$c" $c >> $O/ring.txt
  done
done
sort -s -k1,4 $O/ring.txt | cut -c1-170; tail -3 $O/compile.err
