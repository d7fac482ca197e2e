# same-box A/B of the split-precision convolution before / after a kernel-header change: tools/ubench/_old/kernels/ holds the previous header
# (git show <rev>:monai_amd/csrc/kernels/conv3d_h2.h, common.h -- not committed); writes gpurun_out/h2v/epilogue_ab.txt
O=gpurun_out/h2v; mkdir -p $O; : > $O/epilogue_ab.txt
for res in true false; do
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Itools/ubench/_old -Imonai_amd/csrc -DH2V_RES=$res tools/ubench/h2_variants.hip -o /tmp/h2v_old_$res 2>/dev/null
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc -DH2V_RES=$res tools/ubench/h2_variants.hip -o /tmp/h2v_new_$res 2>/dev/null
done
for rep in 1 2; do
  for v in old new; do
    /tmp/h2v_${v}_true 32 "$v resident" 0 96 64 32 >> $O/epilogue_ab.txt 2>&1
    /tmp/h2v_${v}_false 64 "$v streamed" 0 96 64 32 >> $O/epilogue_ab.txt 2>&1
    /tmp/h2v_${v}_true 32 "$v resident" 0 48 64 32 >> $O/epilogue_ab.txt 2>&1
    /tmp/h2v_${v}_false 128 "$v streamed" 0 12 512 128 >> $O/epilogue_ab.txt 2>&1
  done
done
cat $O/epilogue_ab.txt
