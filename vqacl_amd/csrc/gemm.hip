// bf16 MFMA GEMM for the VL-T5 projections (gfx950, wave64).
//
//   C[M,N] = epilogue( alpha * sum_k A[m,k] * B[n,k] )
//
// A is the activation-side operand, B the weight-side operand (nn.Linear keeps W as [N,K]).
// Either operand may be stored "k-major" (element (r,k) at base + k*ld + r) so the three GEMMs of a
// linear layer run on the same kernel without transposed copies:
//   forward  y  = x  W^T : A = x  [M,K] row-major,  B = W  [N,K] row-major
//   dgrad    dx = dy W   : A = dy [M,N] row-major,  B = W  read k-major (k runs over W's rows)
//   wgrad    dW = dy^T x : A = dy read k-major,     B = x  read k-major (k runs over the M rows)
//
// Structure: 256 threads = 4 waves in a 2x2 grid over a BM x BN tile, BK = 64 per step, two LDS
// stages, global -> registers -> LDS staging (the load of step t+1 is issued before the MFMAs of
// step t).  LDS tiles are always [row][64 k] with k contiguous, 16-byte slots XOR-swizzled by
// (row & 7) so the ds_read_b128 fragment reads are bank-conflict free; k-major operands are
// transposed in registers (8x4 blocks) on their way into LDS.
// MFMA: v_mfma_f32_16x16x32_bf16 with the operands swapped (weight fragment as A, activation
// fragment as B) so each lane ends up with 4 consecutive n of one row m -> 8/16-byte stores.
//
// Fused epilogue (all optional): bias[n], ReLU, gate by the sign of a saved bf16 activation (ReLU and
// dropout backward in one), counter-based dropout, fp32 residual add, accumulate into C, bf16 or
// fp32 output.  Split-K (grid.z) writes fp32 slabs that vlt5_reduce_slabs sums in a fixed order.
#include "common.h"
#include "vlt5_hip.h"

namespace {

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    int M, N, K, lda, ldb, ldc;
    float alpha;
    const float* bias;
    const float* resid; int ldr;
    const bf16_t* gate; int ldg; float gate_scale;
    uint32_t drop_thr, drop_seed;
    int relu, out_f32, accum;
    int ktiles_per_split; long long c_split_stride;
};

constexpr int BK = 64;

__device__ __forceinline__ uint32_t lds_off(int row, int kchunk) {            // byte offset in a [R][64] bf16 tile
    return (uint32_t)(row * 128 + ((kchunk ^ (row & 7)) << 4));
}

// ---- row-major operand: tile [R rows][64 k], 16-byte chunks along k ------------------------------
template <int R>
__device__ __forceinline__ void gload_rm(const bf16_t* __restrict__ base, int ld, int row0, int k0, int rmax, int K,
                                         uint4 (&v)[4], int tid) {
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
        int c = tid + i * 256;
        int row = c >> 3, kc = c & 7;
        int gr = row0 + row, gk = k0 + kc * 8;
        uint4 z = make_uint4(0, 0, 0, 0);
        if (gr < rmax && gk < K) z = *reinterpret_cast<const uint4*>(base + (size_t)gr * ld + gk);
        v[i] = z;
    }
}
template <int R>
__device__ __forceinline__ void lstore_rm(char* tile, const uint4 (&v)[4], int tid) {
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
        int c = tid + i * 256;
        int row = c >> 3, kc = c & 7;
        *reinterpret_cast<uint4*>(tile + lds_off(row, kc)) = v[i];
    }
}

// ---- k-major operand: storage [K][R'] (r contiguous).  A thread owns an 8(r) x 4(k) block ---------
template <int R>
__device__ __forceinline__ void gload_km(const bf16_t* __restrict__ base, int ld, int row0, int k0, int rmax, int K,
                                         uint4 (&v)[4], int tid) {
    if (tid < 2 * R) {
        int kq = (tid & 7) | (((tid >> 6) & 1) << 3);
        int rb = ((tid >> 3) & 7) | ((tid >> 7) << 3);
        int gr = row0 + rb * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int gk = k0 + kq * 4 + c;
            uint4 z = make_uint4(0, 0, 0, 0);
            if (gr < rmax && gk < K) z = *reinterpret_cast<const uint4*>(base + (size_t)gk * ld + gr);
            v[c] = z;
        }
    }
}
__device__ __forceinline__ uint32_t word_of(const uint4& q, int i) {
    return i == 0 ? q.x : (i == 1 ? q.y : (i == 2 ? q.z : q.w));
}
template <int R>
__device__ __forceinline__ void lstore_km(char* tile, const uint4 (&v)[4], int tid) {
    if (tid < 2 * R) {
        int kq = (tid & 7) | (((tid >> 6) & 1) << 3);
        int rb = ((tid >> 3) & 7) | ((tid >> 7) << 3);
        int kc = kq >> 1, within = (kq & 1) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint32_t w0 = word_of(v[0], e >> 1), w1 = word_of(v[1], e >> 1);
            uint32_t w2 = word_of(v[2], e >> 1), w3 = word_of(v[3], e >> 1);
            uint2 o;
            if (e & 1) { o.x = (w0 >> 16) | (w1 & 0xffff0000u); o.y = (w2 >> 16) | (w3 & 0xffff0000u); }
            else       { o.x = (w0 & 0xffffu) | (w1 << 16);     o.y = (w2 & 0xffffu) | (w3 << 16); }
            int row = rb * 8 + e;
            *reinterpret_cast<uint2*>(tile + lds_off(row, kc) + within) = o;
        }
    }
}

template <int BM, int BN, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int FM = BM / 32, FN = BN / 32;                 // 16x16 fragments per wave
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;        // stage s: A tile at s*STAGE_BYTES, B tile right after it

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nk_total = (p.K + BK - 1) / BK;
    int kt0 = 0, kt1 = nk_total;
    char* Cbase = reinterpret_cast<char*>(p.C);
    if (p.ktiles_per_split > 0) {
        kt0 = blockIdx.z * p.ktiles_per_split;
        kt1 = min(nk_total, kt0 + p.ktiles_per_split);
        Cbase += (size_t)blockIdx.z * (size_t)p.c_split_stride * 4;
    }

    f32x4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[4];
    auto gload = [&](int kt) {
        if (AKM) gload_km<BM>(p.A, p.lda, m0, kt * BK, p.M, p.K, ra, tid);
        else     gload_rm<BM>(p.A, p.lda, m0, kt * BK, p.M, p.K, ra, tid);
        if (BKM) gload_km<BN>(p.B, p.ldb, n0, kt * BK, p.N, p.K, rb, tid);
        else     gload_rm<BN>(p.B, p.ldb, n0, kt * BK, p.N, p.K, rb, tid);
    };
    auto lstore = [&](int s) {
        char* at = smem + s * STAGE_BYTES;
        char* bt = at + A_BYTES;
        if (AKM) lstore_km<BM>(at, ra, tid); else lstore_rm<BM>(at, ra, tid);
        if (BKM) lstore_km<BN>(bt, rb, tid); else lstore_rm<BN>(bt, rb, tid);
    };

    if (kt0 < kt1) {
        gload(kt0);
        lstore(0);
    }
    __syncthreads();

    int cur = 0;
    const int lrow = lane & 15, lg = lane >> 4;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = (kt + 1) < kt1;
        if (more) gload(kt + 1);
        const char* at = smem + cur * STAGE_BYTES;
        const char* bt = at + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                int row = wm * (BM / 2) + i * 16 + lrow;
                fa[i] = *reinterpret_cast<const bf16x8_t*>(at + lds_off(row, ks * 4 + lg));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                int row = wn * (BN / 2) + j * 16 + lrow;
                fb[j] = *reinterpret_cast<const bf16x8_t*>(bt + lds_off(row, ks * 4 + lg));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + (lane>>4)*4 -----------
    const float dscale = drop_scale(p.drop_thr);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * (BM / 2) + i * 16 + lrow;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 16 + lg * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha;
            if (p.bias) {
                float4 b = *reinterpret_cast<const float4*>(p.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (p.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (p.gate) {
                uint2 g = *reinterpret_cast<const uint2*>(p.gate + (size_t)m * p.ldg + n);
                uint32_t gw[2] = {g.x, g.y};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bf16_t h = (bf16_t)((gw[r >> 1] >> ((r & 1) * 16)) & 0xffffu);
                    v[r] = (bf16_to_f32(h) > 0.f) ? v[r] * p.gate_scale : 0.f;
                }
            }
            if (p.drop_thr) {
                uint32_t idx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = drop_keep(p.drop_seed, idx + r, p.drop_thr) ? v[r] * dscale : 0.f;
            }
            if (p.resid) {
                float4 q = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ldr + n);
                v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
            }
            if (p.out_f32) {
                float* c = reinterpret_cast<float*>(Cbase) + (size_t)m * p.ldc + n;
                if (p.accum) {
                    float4 q = *reinterpret_cast<const float4*>(c);
                    v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
                }
                *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                bf16_t* c = reinterpret_cast<bf16_t*>(Cbase) + (size_t)m * p.ldc + n;
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(c) = o;
            }
        }
    }
}

__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n,
                                    int nslabs, long long stride, int accum) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 s = accum ? *reinterpret_cast<const float4*>(out + i) : make_float4(0, 0, 0, 0);
    for (int k = 0; k < nslabs; ++k) {
        float4 q = *reinterpret_cast<const float4*>(slabs + (size_t)k * stride + i);
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    *reinterpret_cast<float4*>(out + i) = s;
}

template <int BM, int BN>
int launch_tile(const GemmArgs& a, int akm, int bkm, int splits, hipStream_t st) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, splits > 1 ? splits : 1);
    size_t lds = 2 * (size_t)(BM + BN) * BK * 2;
    if (!akm && !bkm)      hipLaunchKernelGGL((gemm_kernel<BM, BN, false, false>), grid, dim3(256), lds, st, a);
    else if (!akm && bkm)  hipLaunchKernelGGL((gemm_kernel<BM, BN, false, true>), grid, dim3(256), lds, st, a);
    else if (akm && bkm)   hipLaunchKernelGGL((gemm_kernel<BM, BN, true, true>), grid, dim3(256), lds, st, a);
    else                   hipLaunchKernelGGL((gemm_kernel<BM, BN, true, false>), grid, dim3(256), lds, st, a);
    LAUNCH_CHECK();
    return VLT5_OK;
}

}  // namespace

extern "C" int vlt5_gemm_bf16(const vlt5_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->B || !d->C) return VLT5_ERR_ARG;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return VLT5_ERR_ARG;
    // contiguous dimensions are read/written as 8-element (16-byte) vectors
    int a_contig = d->a_kmajor ? d->M : d->K, b_contig = d->b_kmajor ? d->N : d->K;
    if ((a_contig & 7) || (b_contig & 7) || (d->N & 7) || (d->lda & 7) || (d->ldb & 7) || (d->ldc & 3)) return VLT5_ERR_ALIGN;
    if (d->resid && (d->ldr & 3)) return VLT5_ERR_ALIGN;
    if (d->gate && (d->ldg & 3)) return VLT5_ERR_ALIGN;
    if (d->accum && !d->out_f32) return VLT5_ERR_ARG;
    if (d->split_k > 1 && (!d->out_f32 || !d->workspace || d->ldc != d->N || d->bias || d->relu || d->gate || d->drop_p > 0.f || d->resid))
        return VLT5_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GemmArgs a;
    a.A = (const bf16_t*)d->A; a.B = (const bf16_t*)d->B; a.C = d->C;
    a.M = d->M; a.N = d->N; a.K = d->K; a.lda = d->lda; a.ldb = d->ldb; a.ldc = d->ldc;
    a.alpha = d->alpha; a.bias = d->bias; a.resid = d->resid; a.ldr = d->ldr;
    a.gate = (const bf16_t*)d->gate; a.ldg = d->ldg; a.gate_scale = d->gate_scale;
    a.drop_thr = d->drop_p > 0.f ? drop_thr16(d->drop_p) : 0u; a.drop_seed = d->drop_seed;
    a.relu = d->relu; a.out_f32 = d->out_f32; a.accum = d->accum;
    a.ktiles_per_split = 0; a.c_split_stride = 0;

    int bm = d->tile_m, bn = d->tile_n;
    if (bm == 0 || bn == 0) {                       // heuristic: largest tile that still gives >= ~2 blocks per CU
        long t128 = (long)((d->M + 127) / 128) * ((d->N + 127) / 128);
        long t12864 = (long)((d->M + 127) / 128) * ((d->N + 63) / 64);
        int sk = d->split_k > 1 ? d->split_k : 1;
        if (t128 * sk >= 448) { bm = 128; bn = 128; }
        else if (t12864 * sk >= 384) { bm = 128; bn = 64; }
        else { bm = 64; bn = 64; }
    }
    if (!((bm == 128 || bm == 64) && (bn == 128 || bn == 64))) return VLT5_ERR_ARG;

    int splits = d->split_k > 1 ? d->split_k : 1;
    const int nk = (d->K + BK - 1) / BK;
    if (splits > nk) splits = nk;
    if (splits > 1) {
        a.ktiles_per_split = (nk + splits - 1) / splits;
        splits = (nk + a.ktiles_per_split - 1) / a.ktiles_per_split;
        a.c_split_stride = (long long)d->M * d->ldc;
        a.C = d->workspace;
        a.accum = 0;
    }
    int rc;
    if (bm == 128 && bn == 128) rc = launch_tile<128, 128>(a, d->a_kmajor, d->b_kmajor, splits, st);
    else if (bm == 128 && bn == 64) rc = launch_tile<128, 64>(a, d->a_kmajor, d->b_kmajor, splits, st);
    else if (bm == 64 && bn == 128) rc = launch_tile<64, 128>(a, d->a_kmajor, d->b_kmajor, splits, st);
    else rc = launch_tile<64, 64>(a, d->a_kmajor, d->b_kmajor, splits, st);
    if (rc) return rc;
    if (splits > 1) {
        long long n = (long long)d->M * d->ldc;
        int blocks = (int)((n / 4 + 255) / 256);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, st, (const float*)d->workspace,
                           (float*)d->C, n, splits, n, d->accum);
        LAUNCH_CHECK();
    }
    return VLT5_OK;
}

extern "C" long long vlt5_gemm_workspace_bytes(int M, int ldc, int split_k) {
    return split_k > 1 ? (long long)M * ldc * 4 * split_k : 0;
}
