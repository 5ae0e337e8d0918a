// T5LayerNorm (RMS norm) forward / backward, one wave64 per row, fp32 statistics.
// Bandwidth-bound: algorithmic bytes fwd = rows*d*(4 in + 2 out bf16 [+4 f32 out]) ; bwd = rows*d*(4 dy + 4 x + 4..8 dx).
#include "common.h"
#include "vlt5_hip.h"

namespace {

__device__ __forceinline__ int remap_row(int r, int group, int gstride) {
    return group > 0 ? (r / group) * gstride + (r % group) : r;
}

// SLABS: the row is not read from `x` but assembled first -- x_new = resid + dropout(sum of the `nslabs` split-K partial sums the
// producing GEMM left at x + s*slab_stride (fixed slab order)) -- and written to `xsum` (the residual-stream snapshot backward
// reads); i.e. the kernel also plays the residual/dropout epilogue and the slab reduction of that GEMM.
template <bool SLABS>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     bf16_t* __restrict__ yb, float* __restrict__ yf,
                                                     float* __restrict__ rstd_out, int rows, int d, float eps,
                                                     uint32_t thr, uint32_t seed, int group, int gstride, int nslabs,
                                                     long long slab_stride, const float* __restrict__ resid,
                                                     float* __restrict__ xsum, uint32_t rthr, uint32_t rseed) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // one wave per row; 1 or 4 rows per workgroup
    if (row >= rows) return;
    const float* xr = x + (size_t)row * d;
    float ss = 0.f;
    float4 xv[8];                                   // this lane's slice of the row (d <= 2048), read from HBM once
    // SLABS (the decoder's 400-row launches, one row per wave): every load of the row -- slab 0, slabs 1..3, the residual, the norm weight --
    // is requested before the first wait; a slab loop with a wait per iteration and column chunk, the residual behind it and the weight
    // behind the row reduction were up to a dozen dependent round trips in a kernel that moves 12 KB per wave
    float4 s1[SLABS ? 8 : 1], s2[SLABS ? 8 : 1], s3[SLABS ? 8 : 1], rv[SLABS ? 8 : 1], wv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = lane * 4 + k * 256;
        if (c < d) {
            xv[k] = *reinterpret_cast<const float4*>(xr + c);
            if (SLABS) {
                if (nslabs > 1) s1[k] = *reinterpret_cast<const float4*>(xr + (size_t)slab_stride + c);
                if (nslabs > 2) s2[k] = *reinterpret_cast<const float4*>(xr + 2 * (size_t)slab_stride + c);
                if (nslabs > 3) s3[k] = *reinterpret_cast<const float4*>(xr + 3 * (size_t)slab_stride + c);
                rv[k] = *reinterpret_cast<const float4*>(resid + (size_t)row * d + c);
            }
            wv[k] = *reinterpret_cast<const float4*>(w + c);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int c = lane * 4 + k * 256;
        if (c < d) {
            if (SLABS) {
                if (nslabs > 1) { xv[k].x += s1[k].x; xv[k].y += s1[k].y; xv[k].z += s1[k].z; xv[k].w += s1[k].w; }
                if (nslabs > 2) { xv[k].x += s2[k].x; xv[k].y += s2[k].y; xv[k].z += s2[k].z; xv[k].w += s2[k].w; }
                if (nslabs > 3) { xv[k].x += s3[k].x; xv[k].y += s3[k].y; xv[k].z += s3[k].z; xv[k].w += s3[k].w; }
                int sl = 4;
                for (; sl + 3 < nslabs; sl += 4) {            // (more than four slabs) four slab loads in flight, summed in slab order
                    const float* q = xr + (size_t)sl * slab_stride + c;
                    const float4 u0 = *reinterpret_cast<const float4*>(q);
                    const float4 u1 = *reinterpret_cast<const float4*>(q + slab_stride);
                    const float4 u2 = *reinterpret_cast<const float4*>(q + 2 * slab_stride);
                    const float4 u3 = *reinterpret_cast<const float4*>(q + 3 * slab_stride);
                    xv[k].x = (((xv[k].x + u0.x) + u1.x) + u2.x) + u3.x; xv[k].y = (((xv[k].y + u0.y) + u1.y) + u2.y) + u3.y;
                    xv[k].z = (((xv[k].z + u0.z) + u1.z) + u2.z) + u3.z; xv[k].w = (((xv[k].w + u0.w) + u1.w) + u2.w) + u3.w;
                }
                for (; sl < nslabs; ++sl) {
                    const float4 u = *reinterpret_cast<const float4*>(xr + (size_t)sl * slab_stride + c);
                    xv[k].x += u.x; xv[k].y += u.y; xv[k].z += u.z; xv[k].w += u.w;
                }
                if (rthr) {
                    const float rsc = drop_scale(rthr);
                    bool kp[4];
                    drop_keep4(rseed, (uint32_t)row * (uint32_t)d + (uint32_t)c, rthr, kp);
                    xv[k].x = kp[0] ? xv[k].x * rsc : 0.f;
                    xv[k].y = kp[1] ? xv[k].y * rsc : 0.f;
                    xv[k].z = kp[2] ? xv[k].z * rsc : 0.f;
                    xv[k].w = kp[3] ? xv[k].w * rsc : 0.f;
                }
                const float4 r = rv[k];
                xv[k].x += r.x; xv[k].y += r.y; xv[k].z += r.z; xv[k].w += r.w;
                *reinterpret_cast<float4*>(xsum + (size_t)row * d + c) = xv[k];
            }
            ss += xv[k].x * xv[k].x + xv[k].y * xv[k].y + xv[k].z * xv[k].z + xv[k].w * xv[k].w;
        }
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)d + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rs;
    const int orow = remap_row(row, group, gstride);
    const float dsc = drop_scale(thr);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = lane * 4 + k * 256;
        if (c >= d) continue;
        float4 v = xv[k];
        float4 g = wv[k];
        float o[4] = {g.x * (v.x * rs), g.y * (v.y * rs), g.z * (v.z * rs), g.w * (v.w * rs)};
        if (thr) {
            bool kp[4];
            drop_keep4(seed, (uint32_t)row * (uint32_t)d + (uint32_t)c, thr, kp);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = kp[k] ? o[k] * dsc : 0.f;
        }
        if (yf) *reinterpret_cast<float4*>(yf + (size_t)orow * d + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (yb) {
            uint2 pk;
            pk.x = pack_bf16x2(o[0], o[1]);
            pk.y = pack_bf16x2(o[2], o[3]);
            *reinterpret_cast<uint2*>(yb + (size_t)orow * d + c) = pk;
        }
    }
}

constexpr int LN_MAXCH_MAX = 8;   // d <= 2048 (4 chunks of 256 columns per lane suffice for d <= 1024: fewer registers)

// grid-stride over rows; each wave keeps dw partials for its columns, block-reduced through LDS
template <int LN_MAXCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ w, const float* __restrict__ rstd,
                                                     float* __restrict__ dx, float* __restrict__ dwp, int rows, int d,
                                                     int accum_dx, uint32_t thr, uint32_t seed, int group, int gstride,
                                                     bf16_t* __restrict__ dxb, uint32_t thr2, uint32_t seed2, int nslabs,
                                                     long long slab_stride, bf16_t* __restrict__ xn_out) {
    extern __shared__ float red[];                       // [4][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 dwacc[LN_MAXCH];
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k) dwacc[k] = make_float4(0, 0, 0, 0);
    const float dsc = drop_scale(thr);
    const float inv_d = 1.0f / (float)d;
    // the norm weight is the same for every row: keep this lane's columns in registers
    float4 wreg[LN_MAXCH];
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k) {
        int c = lane * 4 + k * 256;
        wreg[k] = (c < d) ? *reinterpret_cast<const float4*>(w + c) : make_float4(0, 0, 0, 0);
    }
    const float dsc2 = drop_scale(thr2);
    // Rows are independent and a wave walks several of them: the loads of the NEXT row (x, dy, and dx when it is accumulated into)
    // are issued before the current row's reduction, so one memory round trip per row instead of two stays exposed.
    auto load_row = [&](int row, float4 (&xv)[LN_MAXCH], float4 (&gv)[LN_MAXCH], float4 (&qv)[LN_MAXCH]) {
        const float* xr = x + (size_t)row * d;
        const float* gr = dy + (size_t)remap_row(row, group, gstride) * d;
        const float* dp = dx + (size_t)row * d;
#pragma unroll
        for (int k = 0; k < LN_MAXCH; ++k) {
            const int c = lane * 4 + k * 256;
            if (c < d) {
                xv[k] = *reinterpret_cast<const float4*>(xr + c);
                gv[k] = *reinterpret_cast<const float4*>(gr + c);
                if (accum_dx) qv[k] = *reinterpret_cast<const float4*>(dp + c);
            }
        }
    };
    // dy handed over as split-K slabs of the producing GEMM (the decoder's 400-row launches): slabs 1..3 are requested together with the
    // row -- into the registers of the row read-ahead, which such launches do not use -- instead of one dependent round trip per slab and
    // column chunk behind the first wait (nslabs = 4: nine in a row, half of the kernel's 9 us)
    auto load_slabs = [&](int row, float4 (&s1)[LN_MAXCH], float4 (&s2)[LN_MAXCH], float4 (&s3)[LN_MAXCH]) {
        const float* gr = dy + (size_t)remap_row(row, group, gstride) * d;
#pragma unroll
        for (int k = 0; k < LN_MAXCH; ++k) {
            const int c = lane * 4 + k * 256;
            if (c < d) {
                s1[k] = *reinterpret_cast<const float4*>(gr + (size_t)slab_stride + c);
                if (nslabs > 2) s2[k] = *reinterpret_cast<const float4*>(gr + 2 * (size_t)slab_stride + c);
                if (nslabs > 3) s3[k] = *reinterpret_cast<const float4*>(gr + 3 * (size_t)slab_stride + c);
            }
        }
    };
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    float4 xv[LN_MAXCH], gv[LN_MAXCH], qv[LN_MAXCH], xn[LN_MAXCH], gn[LN_MAXCH], qn[LN_MAXCH];
    if (row < rows) load_row(row, xv, gv, qv);
    for (; row < rows; row += stride) {
        const int nxt = row + stride;
        const bool ahead = nslabs == 1 && nxt < rows;           // (with slabs a wave has one row: few-row launches)
        if (ahead) load_row(nxt, xn, gn, qn);
        if (nslabs > 1) load_slabs(row, xn, gn, qn);
        const float rs = rstd[row];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_MAXCH; ++k) {
            int c = lane * 4 + k * 256;
            if (c < d) {
                float4 t = gv[k];
                if (nslabs > 1) {          // fixed-order sum over the slabs: 1..3 are in registers, the rest 4 loads at a time
                    const float* gr = dy + (size_t)remap_row(row, group, gstride) * d + c;
                    t.x += xn[k].x; t.y += xn[k].y; t.z += xn[k].z; t.w += xn[k].w;
                    if (nslabs > 2) { t.x += gn[k].x; t.y += gn[k].y; t.z += gn[k].z; t.w += gn[k].w; }
                    if (nslabs > 3) { t.x += qn[k].x; t.y += qn[k].y; t.z += qn[k].z; t.w += qn[k].w; }
                    int sl = 4;
                    for (; sl + 3 < nslabs; sl += 4) {
                        const float* q = gr + (size_t)sl * slab_stride;
                        const float4 u0 = *reinterpret_cast<const float4*>(q);
                        const float4 u1 = *reinterpret_cast<const float4*>(q + slab_stride);
                        const float4 u2 = *reinterpret_cast<const float4*>(q + 2 * slab_stride);
                        const float4 u3 = *reinterpret_cast<const float4*>(q + 3 * slab_stride);
                        t.x = (((t.x + u0.x) + u1.x) + u2.x) + u3.x; t.y = (((t.y + u0.y) + u1.y) + u2.y) + u3.y;
                        t.z = (((t.z + u0.z) + u1.z) + u2.z) + u3.z; t.w = (((t.w + u0.w) + u1.w) + u2.w) + u3.w;
                    }
                    for (; sl < nslabs; ++sl) {
                        const float4 u = *reinterpret_cast<const float4*>(gr + (size_t)sl * slab_stride);
                        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                    }
                }
                if (thr) {
                    bool kp[4];
                    drop_keep4(seed, (uint32_t)row * (uint32_t)d + (uint32_t)c, thr, kp);
                    t.x = kp[0] ? t.x * dsc : 0.f;
                    t.y = kp[1] ? t.y * dsc : 0.f;
                    t.z = kp[2] ? t.z * dsc : 0.f;
                    t.w = kp[3] ? t.w * dsc : 0.f;
                }
                gv[k] = t;
                s += t.x * wreg[k].x * xv[k].x + t.y * wreg[k].y * xv[k].y + t.z * wreg[k].z * xv[k].z + t.w * wreg[k].w * xv[k].w;
                dwacc[k].x += t.x * xv[k].x * rs; dwacc[k].y += t.y * xv[k].y * rs;
                dwacc[k].z += t.z * xv[k].z * rs; dwacc[k].w += t.w * xv[k].w * rs;
            }
        }
        s = wave_sum(s);
        const float coef = rs * rs * rs * s * inv_d;
#pragma unroll
        for (int k = 0; k < LN_MAXCH; ++k) {
            int c = lane * 4 + k * 256;
            if (c < d) {
                float4 o;
                o.x = rs * gv[k].x * wreg[k].x - xv[k].x * coef; o.y = rs * gv[k].y * wreg[k].y - xv[k].y * coef;
                o.z = rs * gv[k].z * wreg[k].z - xv[k].z * coef; o.w = rs * gv[k].w * wreg[k].w - xv[k].w * coef;
                float* dp = dx + (size_t)row * d + c;
                if (accum_dx) { o.x += qv[k].x; o.y += qv[k].y; o.z += qv[k].z; o.w += qv[k].w; }
                *reinterpret_cast<float4*>(dp) = o;
                if (xn_out) {                   // the norm's forward output bf16(x * rstd * w): the operand of the weight-gradient GEMM of the
                    uint2 pn;                   // projection behind it, when the forward never materialised it (norm folded around the GEMM)
                    pn.x = pack_bf16x2(wreg[k].x * (xv[k].x * rs), wreg[k].y * (xv[k].y * rs));
                    pn.y = pack_bf16x2(wreg[k].z * (xv[k].z * rs), wreg[k].w * (xv[k].w * rs));
                    *reinterpret_cast<uint2*>(xn_out + (size_t)row * d + c) = pn;
                }
                if (dxb) {                      // bf16(dropout(dx)) = the A operand of the next sublayer's backward GEMMs
                    float q[4] = {o.x, o.y, o.z, o.w};
                    if (thr2) {
                        bool kp[4];
                        drop_keep4(seed2, (uint32_t)row * (uint32_t)d + (uint32_t)c, thr2, kp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) q[e] = kp[e] ? q[e] * dsc2 : 0.f;
                    }
                    uint2 pk;
                    pk.x = pack_bf16x2(q[0], q[1]);
                    pk.y = pack_bf16x2(q[2], q[3]);
                    *reinterpret_cast<uint2*>(dxb + (size_t)row * d + c) = pk;
                }
            }
        }
        if (ahead) {
#pragma unroll
            for (int k = 0; k < LN_MAXCH; ++k) { xv[k] = xn[k]; gv[k] = gn[k]; qv[k] = qn[k]; }
        } else if (nxt < rows) {
            load_row(nxt, xv, gv, qv);
        }
    }
#pragma unroll
    for (int k = 0; k < LN_MAXCH; ++k) {
        int c = lane * 4 + k * 256;
        if (c < d) *reinterpret_cast<float4*>(red + wave * d + c) = dwacc[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256)
        dwp[(size_t)blockIdx.x * d + c] = red[c] + red[d + c] + red[2 * d + c] + red[3 * d + c];
}

// sum of rows grp, grp+4, grp+8, ... of one column: four independent accumulators so four loads are always in flight (a single
// dependent chain of ~80 loads is latency-bound); fixed order
__device__ __forceinline__ float strided_rows_sum(const float* __restrict__ col, size_t row_stride, int grp, int nblk) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = grp;
    for (; b + 60 < nblk; b += 64) {            // sixteen loads in flight; the four sums receive their terms in the order of the loop below
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = col[(size_t)(b + 4 * u) * row_stride];
#pragma unroll
        for (int u = 0; u < 16; u += 4) { s0 += t[u]; s1 += t[u + 1]; s2 += t[u + 2]; s3 += t[u + 3]; }
    }
    for (; b + 12 < nblk; b += 16) {
        s0 += col[(size_t)b * row_stride];
        s1 += col[(size_t)(b + 4) * row_stride];
        s2 += col[(size_t)(b + 8) * row_stride];
        s3 += col[(size_t)(b + 12) * row_stride];
    }
    for (; b < nblk; b += 4) s0 += col[(size_t)b * row_stride];
    return (s0 + s1) + (s2 + s3);
}

// out[c] (+)= sum_blk partial[blk*row_stride + c]: 64 columns per block, 4 row groups reduced through LDS (fixed order)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int width,
                                                     int row_stride, int accum) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < width) s = strided_rows_sum(partial + c, row_stride, grp, nblk);
    sh[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && c < width) {
        float t = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
        out[c] = accum ? out[c] + t : t;
    }
}

struct MultiColsum { long long out_off[64]; int nblk[64]; };
// job j (blockIdx.y): out[out_off[j] + c] = sum_blk partial[(j*slot_rows + blk)*width + c]   -- one launch for all norms
__global__ __launch_bounds__(256) void colsum_multi_kernel(const float* __restrict__ partial, float* __restrict__ out_base,
                                                           MultiColsum t, int slot_rows, int width) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6, j = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;
    const float* pj = partial + (size_t)j * slot_rows * width;
    float s = 0.f;
    if (c < width) s = strided_rows_sum(pj + c, width, grp, t.nblk[j]);
    sh[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && c < width) out_base[t.out_off[j] + c] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}

}  // namespace

extern "C" int vlt5_layernorm_bwd_blocks(int rows) {
    int b = (rows + 3) / 4;
    return b < LNB_MAXBLK ? (b < 1 ? 1 : b) : LNB_MAXBLK;
}

extern "C" int vlt5_layernorm_fwd(const float* x, const float* w, void* y_bf16, float* y_f32, float* rstd, int rows, int d,
                                  float eps, float drop_p, uint32_t drop_seed, int out_group, int out_group_stride,
                                  void* stream) {
    if (!x || !w || (!y_bf16 && !y_f32) || rows <= 0 || d <= 0) return VLT5_ERR_ARG;
    if ((d & 3) || d > 2048) return VLT5_ERR_ALIGN;
    uint32_t thr = drop_p > 0.f ? drop_thr16(drop_p) : 0u;
    const int rpw = rows < 2048 ? 1 : 4;
    hipLaunchKernelGGL(ln_fwd_kernel<false>, dim3((rows + rpw - 1) / rpw), dim3(64 * rpw), 0, (hipStream_t)stream, x, w, (bf16_t*)y_bf16, y_f32,
                       rstd, rows, d, eps, thr, drop_seed, out_group, out_group_stride, 1, 0ll, nullptr, nullptr, 0u, 0u);
    LAUNCH_CHECK();
    return VLT5_OK;
}

extern "C" int vlt5_layernorm_fwd_slabs(const float* slabs, int nslabs, long long slab_stride, const float* resid, float* x_out,
                                        float resid_drop_p, uint32_t resid_drop_seed, const float* w, void* y_bf16, float* y_f32,
                                        float* rstd, int rows, int d, float eps, float drop_p, uint32_t drop_seed, int out_group,
                                        int out_group_stride, void* stream) {
    if (!slabs || !resid || !x_out || !w || (!y_bf16 && !y_f32) || rows <= 0 || d <= 0) return VLT5_ERR_ARG;
    if (nslabs < 1 || (nslabs > 1 && (slab_stride <= 0 || (slab_stride & 3)))) return VLT5_ERR_ARG;
    if ((d & 3) || d > 2048) return VLT5_ERR_ALIGN;
    uint32_t thr = drop_p > 0.f ? drop_thr16(drop_p) : 0u;
    uint32_t rthr = resid_drop_p > 0.f ? drop_thr16(resid_drop_p) : 0u;
    const int rpw = rows < 2048 ? 1 : 4;          // few rows (decoder): one row per workgroup, so every CU gets work
    hipLaunchKernelGGL(ln_fwd_kernel<true>, dim3((rows + rpw - 1) / rpw), dim3(64 * rpw), 0, (hipStream_t)stream, slabs, w, (bf16_t*)y_bf16, y_f32,
                       rstd, rows, d, eps, thr, drop_seed, out_group, out_group_stride, nslabs, slab_stride, resid, x_out, rthr,
                       resid_drop_seed);
    LAUNCH_CHECK();
    return VLT5_OK;
}

extern "C" int vlt5_layernorm_bwd(const float* dy, const float* x, const float* w, const float* rstd, float* dx, float* dw,
                                  float* dw_partial, int rows, int d, int accum_dx, int accum_dw, float drop_p,
                                  uint32_t drop_seed, int in_group, int in_group_stride, void* dx_bf16, float dx_drop_p,
                                  uint32_t dx_drop_seed, void* stream) {
    return vlt5_layernorm_bwd_slabs(dy, 1, 0, x, w, rstd, dx, dw, dw_partial, rows, d, accum_dx, accum_dw, drop_p, drop_seed, in_group,
                                    in_group_stride, dx_bf16, dx_drop_p, dx_drop_seed, stream);
}

extern "C" int vlt5_layernorm_bwd_slabs(const float* dy, int nslabs, long long slab_stride, const float* x, const float* w,
                                        const float* rstd, float* dx, float* dw, float* dw_partial, int rows, int d, int accum_dx,
                                        int accum_dw, float drop_p, uint32_t drop_seed, int in_group, int in_group_stride,
                                        void* dx_bf16, float dx_drop_p, uint32_t dx_drop_seed, void* stream) {
    return vlt5_layernorm_bwd_full(dy, nslabs, slab_stride, x, w, rstd, dx, dw, dw_partial, rows, d, accum_dx, accum_dw, drop_p, drop_seed,
                                   in_group, in_group_stride, dx_bf16, dx_drop_p, dx_drop_seed, nullptr, stream);
}

extern "C" int vlt5_layernorm_bwd_full(const float* dy, int nslabs, long long slab_stride, const float* x, const float* w,
                                       const float* rstd, float* dx, float* dw, float* dw_partial, int rows, int d, int accum_dx,
                                       int accum_dw, float drop_p, uint32_t drop_seed, int in_group, int in_group_stride,
                                       void* dx_bf16, float dx_drop_p, uint32_t dx_drop_seed, void* xn_out_bf16, void* stream) {
    if (!dy || !x || !w || !rstd || !dx || !dw_partial || rows <= 0) return VLT5_ERR_ARG;
    if (nslabs < 1 || (nslabs > 1 && (slab_stride <= 0 || (slab_stride & 3)))) return VLT5_ERR_ARG;
    if ((d & 3) || d > 256 * LN_MAXCH_MAX) return VLT5_ERR_ALIGN;
    uint32_t thr = drop_p > 0.f ? drop_thr16(drop_p) : 0u;
    uint32_t thr2 = dx_drop_p > 0.f ? drop_thr16(dx_drop_p) : 0u;
    int nblk = vlt5_layernorm_bwd_blocks(rows);
    if (d <= 1024)
        hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(nblk), dim3(256), 4 * d * sizeof(float), (hipStream_t)stream, dy, x, w, rstd, dx,
                           dw_partial, rows, d, accum_dx, thr, drop_seed, in_group, in_group_stride, (bf16_t*)dx_bf16, thr2,
                           dx_drop_seed, nslabs, slab_stride, (bf16_t*)xn_out_bf16);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<8>, dim3(nblk), dim3(256), 4 * d * sizeof(float), (hipStream_t)stream, dy, x, w, rstd, dx,
                           dw_partial, rows, d, accum_dx, thr, drop_seed, in_group, in_group_stride, (bf16_t*)dx_bf16, thr2,
                           dx_drop_seed, nslabs, slab_stride, (bf16_t*)xn_out_bf16);
    LAUNCH_CHECK();
    if (dw) {
        hipLaunchKernelGGL(colsum_kernel, dim3((d + 63) / 64), dim3(256), 0, (hipStream_t)stream, dw_partial, dw, nblk, d, d,
                           accum_dw);
        LAUNCH_CHECK();
    }
    return VLT5_OK;
}

extern "C" int vlt5_colsum_multi(const float* partial, float* out_base, const long long* out_off, const int* nblk, int njobs,
                                 int slot_rows, int width, void* stream) {
    if (!partial || !out_base || !out_off || !nblk || njobs < 1 || njobs > 64 || slot_rows < 1 || width < 1) return VLT5_ERR_ARG;
    MultiColsum t;
    for (int j = 0; j < njobs; ++j) { t.out_off[j] = out_off[j]; t.nblk[j] = nblk[j]; }
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((width + 63) / 64, njobs), dim3(256), 0, (hipStream_t)stream, partial, out_base,
                       t, slot_rows, width);
    LAUNCH_CHECK();
    return VLT5_OK;
}

extern "C" int vlt5_colsum(const float* partial, float* out, int nblk, int width, int row_stride, int accum, void* stream) {
    if (!partial || !out || nblk <= 0 || width <= 0 || row_stride < width) return VLT5_ERR_ARG;
    hipLaunchKernelGGL(colsum_kernel, dim3((width + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial, out, nblk, width,
                       row_stride, accum);
    LAUNCH_CHECK();
    return VLT5_OK;
}
