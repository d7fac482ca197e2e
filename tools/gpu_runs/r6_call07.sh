# round 6, call 7: variants of the separable resample stream kernel on the standalone harness (tools/ubench/resample_variants.hip), config 4 shape, interleaved twice
export TMPDIR=/tmp
O=gpurun_out/r6c07; mkdir -p $O; : > $O/variants.txt
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Imonai_amd/csrc $1 tools/ubench/resample_variants.hip -o $2 2>/dev/null; }
IFS=';' read -ra VS <<< "${VARIANTS:-;-DRSV_VEC=false;-DMH_RS_MINW=5;-DRSV_NL=4 -DRSV_NT=512;-DRSV_NL=4 -DRSV_NT=512 -DMH_RS_MINW=6;-DRSV_T=float -DRSV_NL=4 -DRSV_NT=512;-DRSV_T=float -DRSV_NL=4 -DRSV_NT=512 -DMH_RS_MINW=8;-DRSV_T=float -DRSV_NL=8 -DRSV_NT=256;-DRSV_T=float -DRSV_NL=8 -DRSV_NT=256 -DMH_RS_MINW=6}"
i=0
for v in "${VS[@]}"; do build "$v" /tmp/rsv_$i; i=$((i+1)); done
for rep in 1 2; do
  i=0
  for v in "${VS[@]}"; do /tmp/rsv_$i "${v:-default (double, 8, 256, VEC)}" >> $O/variants.txt 2>&1; i=$((i+1)); done
done
cat $O/variants.txt
