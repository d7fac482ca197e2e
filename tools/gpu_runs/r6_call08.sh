# round 6, call 8: z-chunk sweep of the separable resample stream kernel (standalone harness): fp64 default form and the fp32 512-thread form
export TMPDIR=/tmp
O=gpurun_out/r6c08; mkdir -p $O; : > $O/chunks.txt
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Imonai_amd/csrc $1 tools/ubench/resample_variants.hip -o $2 2>/dev/null; }
build "" /tmp/rsv_d; build "-DRSV_T=float -DRSV_NL=4 -DRSV_NT=512" /tmp/rsv_f; build "-DRSV_T=float -DRSV_NL=4 -DRSV_NT=512 -DMH_RS_MINW=8" /tmp/rsv_f8
for rep in 1 2; do
  for c in 0 6 8 11 14 17 21 26 32 41 52; do
    /tmp/rsv_d "fp64 256 chunks=$c" $c >> $O/chunks.txt; /tmp/rsv_f "fp32 512 chunks=$c" $c >> $O/chunks.txt; /tmp/rsv_f8 "fp32 512 minw8 chunks=$c" $c >> $O/chunks.txt
  done
done
sort -s -k1,4 $O/chunks.txt | cut -c1-150
