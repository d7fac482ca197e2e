# A/B of the split-precision convolution's two region shapes (16 x 16 vs 8 x 32 outputs per workgroup) on the levels where they differ or might:
# 24^3 (4 vs 3 regions per plane), 48^3 and 96^3 (control: the launcher keeps 16 x 16 there); writes gpurun_out/h2v/wide.txt
O=gpurun_out/h2v; mkdir -p $O; : > $O/wide.txt
for wide in false true; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc -DH2V_WIDE=$wide $XFLAGS tools/ubench/h2_variants.hip -o /tmp/h2v_$wide 2>/dev/null
done
IFS=';' read -ra CF <<< "${CFGS:-64 24 256 128;128 24 256 128;256 24 256 128;64 48 64 64;32 96 32 32}"
for cfg in "${CF[@]}"; do
  set -- $cfg
  for wide in false true; do /tmp/h2v_$wide $1 "8x32=$wide" 0 $2 $3 $4 >> $O/wide.txt 2>&1; done
done
cat $O/wide.txt
