# the z-Winograd split-precision convolution against the direct one, standalone (tools/ubench/h2z_variants.hip): VARIANTS="<flags>;<flags>" build variants,
# SHAPES="Cin edge windows Cout;..." the layer shapes; writes gpurun_out/h2zv/variants.txt
O=gpurun_out/h2zv; mkdir -p $O; : > $O/variants.txt
build() { hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc $1 tools/ubench/h2z_variants.hip -o /tmp/h2zv 2>/dev/null; }
IFS=';' read -ra VS <<< "${VARIANTS:-}"
[ ${#VS[@]} -eq 0 ] && VS=("")
IFS=';' read -ra SH <<< "${SHAPES:-32 96 64 32;64 96 64 32;32 48 64 32;64 24 64 64;128 24 64 64}"
for v in "${VS[@]}"; do
  build "$v" || { echo "build failed: $v" >> $O/variants.txt; continue; }
  for sh in "${SH[@]}"; do
    set -- $sh
    timeout 120 /tmp/h2zv $1 "${v:-default}" $2 $3 $4 >> $O/variants.txt 2>&1
  done
done
cat $O/variants.txt
