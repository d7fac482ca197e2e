O=gpurun_out/h2v; mkdir -p $O; : > $O/variants.txt
build() { hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc $1 tools/ubench/h2_variants.hip -o /tmp/h2v 2>/dev/null; }
for v in "" "-DH2V_RES=true"; do
  build "$v" && /tmp/h2v 32 "${v:-full}" >> $O/variants.txt 2>&1
done
cat $O/variants.txt
