# builds tools/ubench/h2_variants.hip (the split-precision convolution standalone, seconds per build) with each listed set of -D switches and times it;
# VARIANTS="<flags>;<flags>;..." overrides the list; writes gpurun_out/h2v/variants.txt
O=gpurun_out/h2v; mkdir -p $O; : > $O/variants.txt
build() { hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc $1 tools/ubench/h2_variants.hip -o /tmp/h2v 2>/dev/null; }
IFS=';' read -ra VS <<< "${VARIANTS:-;-DH2V_RES=true;-DH2X_XTRA=1;-DH2X_XTRA=3;-DH2V_RES=true -DH2X_XTRA=3}"
for v in "${VS[@]}"; do
  for cin in ${CINS:-32 64}; do
    case "$v" in *H2V_RES=true*) [ "$cin" -gt 32 ] && continue;; esac
    build "$v" && /tmp/h2v $cin "${v:-default}" >> $O/variants.txt 2>&1
  done
done
cat $O/variants.txt
