// 224 x 256 tile instantiations (8 waves, row-major A) of the GEMM kernel template; see gemm_kernel.h / gemm.hip.
#include "gemm_kernel.h"
namespace vlt5gemm {
int launch_224x256(const GemmArgs& a, int akm, int bkm, int splits, int batch, hipStream_t st) {
    return launch_tile<224, 256>(a, akm, bkm, splits, batch, st);
}
}  // namespace vlt5gemm
