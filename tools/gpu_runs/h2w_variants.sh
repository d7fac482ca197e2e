# builds tools/ubench/h2w_variants.hip (the Winograd split-precision convolution standalone, seconds per build) with each listed set of -D switches and times it;
# HWINC=<dir with kernels/conv3d_wino_h2.h> builds an experimental copy of the header (tools/experiments/...) instead of the product one;
# VARIANTS="<flags>;<flags>;..." overrides the list; writes gpurun_out/h2wv/variants.txt
O=gpurun_out/h2wv; mkdir -p $O; : > $O/variants.txt
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -Iinclude ${HWINC:+-I$HWINC} -Imonai_amd/csrc $1 tools/ubench/h2w_variants.hip -o /tmp/h2wv 2>>$O/compile.err; }
IFS=';' read -ra VS <<< "${VARIANTS:-;-DHWX_PROF;-DHWX_OFF=1;-DHWX_OFF=2;-DHWX_OFF=4;-DHWX_OFF=8;-DHWX_OFF=16;-DHWX_OFF=7;-DHWX_OFF=23;-DHWX_OFF=31}"
for v in "${VS[@]}"; do
  build "$v" && /tmp/h2wv "${v:-default}" ${EDGE:-96} >> $O/variants.txt 2>&1
done
cat $O/variants.txt
