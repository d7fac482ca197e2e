# what the fp16 matrix pipe sustains over ~0.5 s per configuration on zero / smooth / random / split-piece operands (tools/ubench/mfma_sustained.hip):
# the practical ceiling of roofline.frac for the split-precision convolution; writes gpurun_out/ubench/mfma_sustained.txt
O=gpurun_out/ubench; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_sustained.hip -o /tmp/mfma_sustained 2>/dev/null && timeout 120 /tmp/mfma_sustained ${ITERS:-60000} ${LAUNCHES:-30} > $O/mfma_sustained.txt 2>&1
cat $O/mfma_sustained.txt
