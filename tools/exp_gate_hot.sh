# EXPERIMENT (results WRONG on purpose): the ceiling of a 1-bit ReLU mask instead of the saved FFN activation in the hidden-gradient epilogue --
# the encoder's 4480 x 3072 x 768 input-gradient GEMMs gate by ONE cache-resident row (tools/experiments/ffn_gate_hot.patch, -DFFN_GATE_HOT).
#   git apply tools/experiments/ffn_gate_hot.patch && bash tools/build_variant.sh gatehot -DFFN_GATE_HOT && git checkout -- vqacl_amd/csrc/engine.hip
export VLT5_ALLOW_EXPERIMENT=1
bash tools/ab_libs.sh 4 vqacl_amd/libvlt5_hip.so vqacl_amd/libvlt5_gatehot.so
