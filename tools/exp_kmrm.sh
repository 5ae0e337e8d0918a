# EXPERIMENT: the interleaved k-step (KM_STEP: DMA pieces spread between the MFMAs, fragments of both halves up front) for ROW-MAJOR operands too
#   measured in round 6 and not kept (profiles/r06_g_kmrm_probe.txt).  To repeat: in vqacl_amd/csrc/gemm_kernel.h make the line that defines
#   KM_STEP read `(AKM || BKM || GEMM_KM_STEP_RM)` with `#define GEMM_KM_STEP_RM 0` as the default above it, then
#   bash tools/build_variant.sh kmrm -DGEMM_KM_STEP_RM=1
SH="4480 768 3072 0 0 64 128|4480 768 3072 0 0 128 64|4480 768 768 0 0 64 128|4480 768 768 0 0 128 64|4480 768 2304 0 0 64 128|400 768 768 0 0 64 64|400 2304 768 0 0 64 64|400 3072 768 0 0 64 64|2880 768 2048 0 0 64 128|400 768 3072 0 0 64 64 1 4"
IFS='|' read -ra ARR <<< "$SH"
for lib in libvlt5_hip.so libvlt5_kmrm.so; do echo "== $lib"; VLT5_LIB=$PWD/vqacl_amd/$lib python3 tools/gemm_probe2.py "${ARR[@]}" 2>&1 | grep "M="; done
VLT5_LIB=$PWD/vqacl_amd/libvlt5_kmrm.so timeout 900 python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" 2>&1 | tail -2
bash tools/ab_libs.sh 3 vqacl_amd/libvlt5_hip.so vqacl_amd/libvlt5_kmrm.so 2>&1 | tail -8
