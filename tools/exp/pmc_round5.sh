mkdir -p gpurun_out/r05g
for layer in conv3 conv4 conv5; do
  ODW_CONV_P2=1 bash tools/pmc_conv.sh $layer > gpurun_out/r05g/pmc_conv_planes2_$layer.txt 2>&1
done
ODW_CONV_P2=1 ODW_CONV_P2_OUT=planes bash tools/pmc_conv.sh conv3 > gpurun_out/r05g/pmc_conv_planes2_conv3_planesout.txt 2>&1
ODW_ONE_SHAPE=2000,25088,4096 bash tools/pmc_gemm.sh big > gpurun_out/r05g/pmc_gemm_big7_fc6dgrad.txt 2>&1
cat gpurun_out/r05g/pmc_conv_planes2_conv4.txt | head -40
