// 64 x 128 tile instantiations of the GEMM kernel template (all four operand layouts); see gemm_kernel.h / gemm.hip.
#include "gemm_kernel.h"
namespace vlt5gemm {
int launch_64x128(const GemmArgs& a, int akm, int bkm, int splits, int batch, hipStream_t st) {
    return launch_tile<64, 128>(a, akm, bkm, splits, batch, st);
}
}  // namespace vlt5gemm
