# EXPERIMENT (tools/experiments/gemm_fat128.patch, -DGEMM_FAT128=1: the 128x128 tile with four k-tile slots and one barrier per two k-tiles): bash tools/build_variant.sh fat -DGEMM_FAT128=1 first
SH="4480 768 3072 0 0 128 128|4480 768 3072 0 1 128 128|4480 768 3072 0 0 64 128|4480 768 3072 0 1 128 64|4480 768 2304 0 1 128 128|4480 768 2304 0 1 128 64|4480 768 768 0 0 128 128|4480 768 768 0 0 64 128|2304 768 4480 1 1 128 128 6|4480 3072 768 0 0 128 128|4480 2304 768 0 0 128 128"
IFS='|' read -ra ARR <<< "$SH"
for lib in libvlt5_hip.so libvlt5_fat.so; do echo "== $lib"; VLT5_LIB=$PWD/vqacl_amd/$lib python3 tools/gemm_probe2.py "${ARR[@]}" 2>&1 | grep "M="; done
VLT5_LIB=$PWD/vqacl_amd/libvlt5_fat.so timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" 2>&1 | tail -3
