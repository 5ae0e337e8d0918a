for L in vqacl_amd/libvlt5_dma0.so vqacl_amd/libvlt5_dma2.so vqacl_amd/libvlt5_hip.so; do
echo "== $L"
VLT5_LIB=$PWD/$L python tools/gemm_probe2.py "4096 4096 4480 0 0 256 256" "4096 4096 4480 0 1 256 256" "4096 4096 4480 1 1 256 256" "4096 4096 4480 1 1 128 128" "4480 3072 768 0 0 224 256" "4480 3072 768 0 1 224 256" "4480 768 3072 0 0 64 128" "4480 768 3072 0 1 128 64" "4640 18432 768 0 0 224 256" 2>&1 | grep -v amdgpu
done
