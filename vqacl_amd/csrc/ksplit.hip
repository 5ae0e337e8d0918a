// K-split GEMM for few-row activations (the decoder stack: M = B*T = 400 rows), round 6.
//
// C[M,N] = epi(alpha * A[M,K] B[N,K]^T), A and B row-major bf16 -- nn.Linear of the decoder's T5Block sublayers (HF T5Attention q/k/v/o,
// T5DenseReluDense wi; called from VL-T5/src/modeling_t5_our.py:641-655).
//
// Why another kernel: at M = 400 the tiled kernel (gemm_kernel.h) runs 84-252 workgroups whose k-loop is 12 SERIAL k-steps of ~840
// cycles each (barrier, fragment reads, 8 MFMAs per wave, the next LDS-DMA request): 4-5 us of dependent latency for 0.5 us of matrix
// work, on the decoder's launch chain.  Here the REDUCTION is split over the four waves of a workgroup instead: every wave
// owns a quarter of K for the whole 64 x 64 (or 32 x 64) output tile, requests all of its operand fragments straight into registers in
// MFMA layout -- a wave's operands are private, nothing to share through LDS, no ring, no barrier in the loop; one wave per SIMD has the
// 512-register file to itself -- pays the memory latency ONCE, runs its 96 (48) MFMAs back to back, and the four partial tiles are
// summed through LDS in a fixed order.  Epilogue as the tiled kernel's (alpha, ReLU, inverted dropout on element index m*N+n with the
// same counters, f32 residual), bit-compatible masks.
#include "common.h"
#include "vlt5_hip.h"

namespace {

struct KsArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K, lda, ldb, ldc, ldr;
    float alpha, dscale;
    const float* resid;
    uint32_t drop_thr, drop_seed;
    int relu;
};

// FM: 16-row fragments per tile (4: 64 x 64, 2: 32 x 64); STEPS: 32-deep k-slices per wave (K = 4 * 32 * STEPS)
template <int FM, int STEPS, bool F32OUT>
__global__ __launch_bounds__(256, 1) void ksplit_kernel(KsArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* red = reinterpret_cast<float4*>(smem);                  // [4 waves][FM*4 fragments][64 lanes]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * (FM * 16);
    const int k0 = wave * (STEPS * 32) + lg * 8;
    const bf16_t* ap[FM];
    const bf16_t* bp[4];
#pragma unroll
    for (int i = 0; i < FM; ++i) ap[i] = p.A + (size_t)min(m0 + i * 16 + lrow, p.M - 1) * p.lda + k0;
#pragma unroll
    for (int j = 0; j < 4; ++j) bp[j] = p.B + (size_t)min(n0 + j * 16 + lrow, p.N - 1) * p.ldb + k0;
    bf16x8_t a[STEPS][FM], b[STEPS][4];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {                              // every request of the wave before anything waits
#pragma unroll
        for (int j = 0; j < 4; ++j) b[s][j] = *reinterpret_cast<const bf16x8_t*>(bp[j] + s * 32);
#pragma unroll
        for (int i = 0; i < FM; ++i) a[s][i] = *reinterpret_cast<const bf16x8_t*>(ap[i] + s * 32);
    }
    // (pinned: left to itself the compiler interleaves the requests with the MFMAs to save registers -- 184 instead of ~330 -- and the
    // wave pays the memory latency several times; one wave per SIMD has the whole register file, that is the point of this kernel)
    __builtin_amdgcn_sched_barrier(0);
    f32x4_t acc[FM][4];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)                              // (operands swapped: a lane then owns 4 consecutive n of one row)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[s][j], a[s][i], acc[i][j], 0, 0, 0);
    // the four k-quarters through LDS, summed in wave order by the wave that owns the fragment (f = i*4 + j; wave w owns [w*FM, (w+1)*FM))
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = i * 4 + j;
            if (f / FM != wave)                                       // (the owner keeps its own partial in registers)
                red[(wave * FM * 4 + f) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = i * 4 + j;
            if (f / FM != wave) continue;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {                             // fixed order 0, 1, 2, 3 whoever the owner is
                if (w == wave) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += acc[i][j][r];
                } else {
                    const float4 t = red[(w * FM * 4 + f) * 64 + lane];
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                }
            }
            const int m = m0 + i * 16 + lrow, n = n0 + j * 16 + lg * 4;
            if (m >= p.M || n >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
            if (p.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (p.drop_thr) {
                bool kp[4];
                drop_keep4(p.drop_seed, (uint32_t)m * (uint32_t)p.N + (uint32_t)n, p.drop_thr, kp);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = kp[r] ? v[r] * p.dscale : 0.f;
            }
            if constexpr (F32OUT) {
                if (p.resid) {
                    const float4 q = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n);
                    v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
                }
                store_wt16f(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
            } else {
                store_wt8(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n, make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])));
            }
        }
}

template <int FM, int STEPS>
int launch(const KsArgs& a, bool f32, hipStream_t st) {
    const dim3 grid((a.N + 63) / 64, (a.M + FM * 16 - 1) / (FM * 16));
    const size_t lds = (size_t)4 * FM * 4 * 64 * sizeof(float4);
    if (f32) hipLaunchKernelGGL((ksplit_kernel<FM, STEPS, true>), grid, dim3(256), lds, st, a);
    else     hipLaunchKernelGGL((ksplit_kernel<FM, STEPS, false>), grid, dim3(256), lds, st, a);
    LAUNCH_CHECK();
    return VLT5_OK;
}

}  // namespace

// shapes the kernel takes (the dispatcher and the engine ask the same question): few rows, row-major operands, K = 768 or 1024
bool vlt5_gemm_ksplit_shape_ok(int M, int N, int K) {
    return M >= 1 && M <= 512 && (K == 768 || K == 1024) && N >= 64 && (N % 64) == 0;
}

// VLT5_OK when launched, -1 when the description asks for something this kernel does not do (the caller falls through to the tiles)
int vlt5_gemm_ksplit_try(const vlt5_gemm_desc* g, hipStream_t st) {
    if (!vlt5_gemm_ksplit_shape_ok(g->M, g->N, g->K)) return -1;
    if (g->a_kmajor || g->b_kmajor || g->bias || g->gate || g->accum || g->split_k > 1 || g->batch > 1 || g->grouped_with ||
        g->c_bf16_copy || g->emit_xw_bf16 || g->norm_partials || g->sumsq || g->tile_m || g->tile_n)
        return -1;
    if (g->resid && !g->out_f32) return -1;
    if ((g->lda & 7) || (g->ldb & 7) || (g->ldc & 3) || (g->resid && (g->ldr & 3))) return -1;
    KsArgs a;
    a.A = (const bf16_t*)g->A; a.B = (const bf16_t*)g->B; a.C = g->C;
    a.M = g->M; a.N = g->N; a.K = g->K; a.lda = g->lda; a.ldb = g->ldb; a.ldc = g->ldc; a.ldr = g->ldr;
    a.alpha = g->alpha; a.resid = g->resid; a.relu = g->relu;
    a.drop_thr = g->drop_p > 0.f ? drop_thr16(g->drop_p) : 0u;
    a.drop_seed = g->drop_seed;
    a.dscale = drop_scale(a.drop_thr);
    // 64 x 64 tiles once they give every CU work; 32 x 64 below that (N = 768: 84 -> 156 workgroups)
    const bool tall = ((g->M + 63) / 64) * (g->N / 64) >= 192;
    const bool f32 = g->out_f32 != 0;
    if (g->K == 768) return tall ? launch<4, 6>(a, f32, st) : launch<2, 6>(a, f32, st);
    return tall ? launch<4, 8>(a, f32, st) : launch<2, 8>(a, f32, st);
}
