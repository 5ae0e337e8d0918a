// Embedding-side kernels of the VL-T5 path: token gather (+ input dropout), shift-right, masks,
// the visual embedding tail (two RMS norms + 5->d box projection + order embeddings), and the
// relative-position-bias gather / its gradient.  All bandwidth- or latency-bound, no MFMA work.
#include "common.h"
#include <initializer_list>
#include "vlt5_hip.h"

namespace {

// ---------------- token embedding ---------------------------------------------------------------
__global__ void embed_fwd_kernel(const long long* __restrict__ ids, const float* __restrict__ table, float* __restrict__ out,
                                 long long sb, long long st, int B, int T, int d, int vocab, uint32_t thr, uint32_t seed,
                                 int drop_rows, int drop_row0) {
    const int row = blockIdx.x;                       // b*T + t
    const int b = row / T, t = row % T;
    long long id = ids[row];
    if (id < 0 || id >= vocab) id = 0;
    const float* src = table + (size_t)id * d;
    float* dst = out + b * sb + t * st;
    const float dsc = drop_scale(thr);
    for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
        float4 v = *reinterpret_cast<const float4*>(src + c);
        float o[4] = {v.x, v.y, v.z, v.w};
        if (thr) {
            uint32_t idx = (uint32_t)(((size_t)b * drop_rows + drop_row0 + t) * d + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = drop_keep(seed, idx + k, thr) ? o[k] * dsc : 0.f;
        }
        *reinterpret_cast<float4*>(dst + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Scatter-add of the token-embedding gradients, DETERMINISTIC: the table row of an id that occurs several times in the batch (the
// pad id: hundreds of rows) receives its contributions in a fixed order instead of in the order atomics happen to arrive.
// The sum over the occurrences of an id is a two-level tree with a fixed shape:
//   index pass, one wave per row: one sweep over the id list gives the row's position in the order sorted by (id, row) -- rows with
//     a smaller id + earlier rows with the same id (its rank) -- and the number of rows that share its id; sorted[pos] = row.
//   pass 1, one wave per (row, 256-column chunk): rows with rank % EMB_G == 0 ("leaders") sum the gradient rows of sorted[pos ..
//     pos + EMB_G) that carry their id, in that order; an id with at most EMB_G occurrences is finished here (added to the table).
//   pass 2: the first row of an id with more occurrences sums its leaders' partial sums, in order, and adds the total to the table.
// Every table row is written by exactly one wave of a launch, with plain loads / stores; eight rows are in flight per wave.
constexpr int EMB_G = 24;
struct EmbIdx { int pos, rank, cnt, pad; };          // per row: place in the sorted order, rank among the rows of its id, their number
__device__ __forceinline__ long long emb_clamp(long long v, int vocab) { return (v < 0 || v >= vocab) ? 0ll : v; }
__device__ __forceinline__ float4 emb_row(const float* __restrict__ dout, long long sb, long long st, int j, int T, int d, int c, uint32_t thr,
                                          uint32_t seed, float dsc, int drop_rows, int drop_row0) {
    const int b = j / T, t = j - b * T;
    float4 g = *reinterpret_cast<const float4*>(dout + b * sb + t * st + c);
    if (thr) {
        bool kp[4];
        drop_keep4(seed, (uint32_t)(((size_t)b * drop_rows + drop_row0 + t) * d + c), thr, kp);
        g.x = kp[0] ? g.x * dsc : 0.f; g.y = kp[1] ? g.y * dsc : 0.f; g.z = kp[2] ? g.z * dsc : 0.f; g.w = kp[3] ? g.w * dsc : 0.f;
    }
    return g;
}
__global__ __launch_bounds__(64) void embed_index_kernel(const long long* __restrict__ ids, EmbIdx* __restrict__ idx, int* __restrict__ sorted,
                                                         int n, int vocab) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const long long id = emb_clamp(ids[row], vocab);
    int less = 0, before = 0, equal = 0;
    for (int base = 0; base < n; base += 256) {             // four independent loads per lane and round
        long long v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ids[min(base + u * 64 + lane, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = base + u * 64 + lane;
            const long long w = emb_clamp(v[u], vocab);
            less += (j < n && w < id) ? 1 : 0;
            equal += (j < n && w == id) ? 1 : 0;
            before += (j < row && w == id) ? 1 : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        less += __shfl_xor(less, o, 64); equal += __shfl_xor(equal, o, 64); before += __shfl_xor(before, o, 64);
    }
    if (lane == 0) {
        idx[row].pos = less + before; idx[row].rank = before; idx[row].cnt = equal; idx[row].pad = 0;
        sorted[less + before] = row;
    }
}
__global__ __launch_bounds__(64) void embed_bwd_pass1_kernel(const float* __restrict__ dout, long long sb, long long st, const EmbIdx* __restrict__ idx,
                                                             const int* __restrict__ sorted, const long long* __restrict__ ids,
                                                             float* __restrict__ partial, float* __restrict__ dtable, int T, int d, int vocab,
                                                             uint32_t thr, uint32_t seed, int drop_rows, int drop_row0) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const EmbIdx e = idx[row];
    if (e.rank % EMB_G != 0) return;
    const int cnt = min(EMB_G, e.cnt - e.rank);
    const int mine = lane < cnt ? sorted[e.pos + lane] : row;          // this group's rows, in order
    const int c = blockIdx.y * 256 + lane * 4;
    const float dsc = drop_scale(thr);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < cnt; k += 8) {
        float4 g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = __shfl(mine, min(k + u, cnt - 1), 64);
            g[u] = (k + u < cnt && c < d) ? emb_row(dout, sb, st, j, T, d, c, thr, seed, dsc, drop_rows, drop_row0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
    }
    if (c >= d) return;
    if (e.cnt <= EMB_G) {                                     // the whole id: finish it here
        float* dst = dtable + (size_t)emb_clamp(ids[row], vocab) * d + c;
        float4 o = *reinterpret_cast<const float4*>(dst);
        o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
        *reinterpret_cast<float4*>(dst) = o;
    } else {
        *reinterpret_cast<float4*>(partial + (size_t)row * d + c) = acc;
    }
}
__global__ __launch_bounds__(64) void embed_bwd_pass2_kernel(const EmbIdx* __restrict__ idx, const int* __restrict__ sorted,
                                                             const long long* __restrict__ ids, const float* __restrict__ partial,
                                                             float* __restrict__ dtable, int d, int vocab) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const EmbIdx e = idx[row];
    if (e.rank != 0 || e.cnt <= EMB_G) return;
    const int c = blockIdx.y * 256 + lane * 4;
    if (c >= d) return;
    const int nlead = (e.cnt + EMB_G - 1) / EMB_G;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < nlead; k += 8) {
        float4 g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = sorted[e.pos + min(k + u, nlead - 1) * EMB_G];
            g[u] = k + u < nlead ? *reinterpret_cast<const float4*>(partial + (size_t)j * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
    }
    float* dst = dtable + (size_t)emb_clamp(ids[row], vocab) * d + c;
    float4 o = *reinterpret_cast<const float4*>(dst);
    o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
    *reinterpret_cast<float4*>(dst) = o;
}

__global__ void shift_right_kernel(const long long* __restrict__ labels, long long* __restrict__ out, int B, int T,
                                   int start_id, int pad_id) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    int t = i % T;
    long long v = (t == 0) ? (long long)start_id : labels[i - 1];
    if (v == -100) v = pad_id;
    out[i] = v;
}

__global__ void build_mask_kernel(const long long* __restrict__ ids, float* __restrict__ mask, int B, int L, int S, int pad_id) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * S) return;
    int b = i / S, s = i % S;
    mask[i] = (s < L) ? (ids[b * L + s] != pad_id ? 1.f : 0.f) : 1.f;
}

// ---------------- visual embedding tail -----------------------------------------------------------
__device__ __forceinline__ void load_pos5(const float* __restrict__ boxes, int row, float (&p5)[5]) {
    float4 bx = *reinterpret_cast<const float4*>(boxes + (size_t)row * 4);
    p5[0] = bx.x; p5[1] = bx.y; p5[2] = bx.z; p5[3] = bx.w;
    p5[4] = (bx.w - bx.z) * (bx.y - bx.x);       // get_area reads the columns as (x1,x2,y1,y2)
}
__device__ __forceinline__ float pos_lin(const float* __restrict__ Wp, const float* __restrict__ bp, int c, const float (&p5)[5]) {
    const float* w = Wp + (size_t)c * 5;
    return bp[c] + (p5[0] * w[0] + p5[1] * w[1] + p5[2] * w[2] + p5[3] * w[3] + p5[4] * w[4]);
}

__global__ __launch_bounds__(256) void vis_fwd_kernel(const float* __restrict__ G, const float* __restrict__ boxes,
                                                      const float* __restrict__ Wp, const float* __restrict__ bp,
                                                      const float* __restrict__ lnf_w, const float* __restrict__ lnp_w,
                                                      const float* __restrict__ img0, const float* __restrict__ shared,
                                                      float* __restrict__ out, long long sb, long long st,
                                                      float* __restrict__ rstd_f, float* __restrict__ rstd_p, int B, int V,
                                                      int d, int vocab, float eps, uint32_t thr, uint32_t seed,
                                                      int drop_rows, int drop_row0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * V) return;
    const int b = row / V, i = row % V;
    const float* gr = G + (size_t)row * d;
    float p5[5];
    load_pos5(boxes, row, p5);
    float ssf = 0.f, ssp = 0.f;
    for (int c = lane; c < d; c += 64) {
        float g = gr[c];
        float a = pos_lin(Wp, bp, c, p5);
        ssf += g * g;
        ssp += a * a;
    }
    ssf = wave_sum(ssf);
    ssp = wave_sum(ssp);
    const float rf = rsqrtf(ssf / (float)d + eps), rp = rsqrtf(ssp / (float)d + eps);
    if (lane == 0) { rstd_f[row] = rf; rstd_p[row] = rp; }
    const float* obj = shared + (size_t)(vocab - 1 - i) * d;
    float* dst = out + b * sb + i * st;
    const float dsc = drop_scale(thr);
    for (int c = lane; c < d; c += 64) {
        float a = pos_lin(Wp, bp, c, p5);
        float v = lnf_w[c] * (gr[c] * rf) + lnp_w[c] * (a * rp);
        v = v + img0[c];
        v = v + obj[c];
        if (thr) {
            uint32_t idx = (uint32_t)(((size_t)b * drop_rows + drop_row0 + i) * d + c);
            v = drop_keep(seed, idx, thr) ? v * dsc : 0.f;
        }
        dst[c] = v;
    }
}

// row-wise part of the backward: dG (bf16) and the per-row coefficient of the position branch
__global__ __launch_bounds__(256) void vis_bwd_rows_kernel(const float* __restrict__ dout, long long sb, long long st,
                                                           const float* __restrict__ G, const float* __restrict__ boxes,
                                                           const float* __restrict__ Wp, const float* __restrict__ bp,
                                                           const float* __restrict__ lnf_w, const float* __restrict__ lnp_w,
                                                           const float* __restrict__ rstd_f, const float* __restrict__ rstd_p,
                                                           bf16_t* __restrict__ dG, float* __restrict__ coef_p, float* __restrict__ coef_f,
                                                           int B, int V, int d, uint32_t thr, uint32_t seed, int drop_rows,
                                                           int drop_row0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * V) return;
    const int b = row / V, i = row % V;
    const float* gr = G + (size_t)row * d;
    const float* go = dout + b * sb + i * st;
    float p5[5];
    load_pos5(boxes, row, p5);
    const float rf = rstd_f[row], rp = rstd_p[row];
    const float dsc = drop_scale(thr);
    float sf = 0.f, sp = 0.f;
    for (int c = lane; c < d; c += 64) {
        float g = go[c];
        if (thr) {
            uint32_t idx = (uint32_t)(((size_t)b * drop_rows + drop_row0 + i) * d + c);
            g = drop_keep(seed, idx, thr) ? g * dsc : 0.f;
        }
        sf += g * lnf_w[c] * gr[c];
        sp += g * lnp_w[c] * pos_lin(Wp, bp, c, p5);
    }
    sf = wave_sum(sf);
    sp = wave_sum(sp);
    const float cf = rf * rf * rf * sf / (float)d;
    if (lane == 0) { coef_p[row] = rp * rp * rp * sp / (float)d; coef_f[row] = cf; }
    for (int c = lane; c < d; c += 64) {
        float g = go[c];
        if (thr) {
            uint32_t idx = (uint32_t)(((size_t)b * drop_rows + drop_row0 + i) * d + c);
            g = drop_keep(seed, idx, thr) ? g * dsc : 0.f;
        }
        dG[(size_t)row * d + c] = f32_to_bf16(rf * g * lnf_w[c] - gr[c] * cf);
    }
}

// The same two row kernels with 16-byte accesses (d, the strides and the bases multiples of 4 words): a lane owns 4 consecutive columns of
// each 256-column chunk, every operand of the row is requested before the first reduction, and the position branch a = pos_lin(c) is
// computed once and kept.  The scalar forms above issue ~150 four-byte wave loads per row -- the texture-address unit takes ~50 cycles
// for any wave load -- and ran 18-20 us on 2880 rows; element for element the arithmetic is the same, the per-lane order of the two row
// sums differs (each lane sums its own columns first).
struct Pos4 { float4 a; };
__device__ __forceinline__ float4 pos_lin4(const float* __restrict__ Wp, const float* __restrict__ bp, int c, const float (&p5)[5]) {
    const float4* w = reinterpret_cast<const float4*>(Wp + (size_t)c * 5);          // 4 columns x 5 weights = 20 consecutive words
    const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4], b4 = *reinterpret_cast<const float4*>(bp + c);
    const float q[20] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w, w3.x, w3.y, w3.z, w3.w, w4.x, w4.y, w4.z, w4.w};
    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float* ww = q + e * 5;
        r[e] = bb[e] + (p5[0] * ww[0] + p5[1] * ww[1] + p5[2] * ww[2] + p5[3] * ww[3] + p5[4] * ww[4]);
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}
__global__ __launch_bounds__(256) void vis_fwd_vec_kernel(const float* __restrict__ G, const float* __restrict__ boxes,
                                                          const float* __restrict__ Wp, const float* __restrict__ bp,
                                                          const float* __restrict__ lnf_w, const float* __restrict__ lnp_w,
                                                          const float* __restrict__ img0, const float* __restrict__ shared,
                                                          float* __restrict__ out, long long sb, long long st,
                                                          float* __restrict__ rstd_f, float* __restrict__ rstd_p, int B, int V,
                                                          int d, int vocab, float eps, uint32_t thr, uint32_t seed,
                                                          int drop_rows, int drop_row0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * V) return;
    const int b = row / V, i = row % V;
    const float* gr = G + (size_t)row * d;
    const float* obj = shared + (size_t)(vocab - 1 - i) * d;
    float p5[5];
    load_pos5(boxes, row, p5);
    float4 g[8], a[8], wf[8], wp[8], im[8], ob[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = lane * 4 + k * 256;
        if (c < d) {
            g[k] = *reinterpret_cast<const float4*>(gr + c);
            a[k] = pos_lin4(Wp, bp, c, p5);
            wf[k] = *reinterpret_cast<const float4*>(lnf_w + c);
            wp[k] = *reinterpret_cast<const float4*>(lnp_w + c);
            im[k] = *reinterpret_cast<const float4*>(img0 + c);
            ob[k] = *reinterpret_cast<const float4*>(obj + c);
        }
    }
    float ssf = 0.f, ssp = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (lane * 4 + k * 256 < d) {
            ssf += g[k].x * g[k].x; ssf += g[k].y * g[k].y; ssf += g[k].z * g[k].z; ssf += g[k].w * g[k].w;
            ssp += a[k].x * a[k].x; ssp += a[k].y * a[k].y; ssp += a[k].z * a[k].z; ssp += a[k].w * a[k].w;
        }
    }
    ssf = wave_sum(ssf);
    ssp = wave_sum(ssp);
    const float rf = rsqrtf(ssf / (float)d + eps), rp = rsqrtf(ssp / (float)d + eps);
    if (lane == 0) { rstd_f[row] = rf; rstd_p[row] = rp; }
    float* dst = out + b * sb + i * st;
    const float dsc = drop_scale(thr);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = lane * 4 + k * 256;
        if (c < d) {
            float v[4] = {wf[k].x * (g[k].x * rf) + wp[k].x * (a[k].x * rp), wf[k].y * (g[k].y * rf) + wp[k].y * (a[k].y * rp),
                          wf[k].z * (g[k].z * rf) + wp[k].z * (a[k].z * rp), wf[k].w * (g[k].w * rf) + wp[k].w * (a[k].w * rp)};
            v[0] = v[0] + im[k].x; v[1] = v[1] + im[k].y; v[2] = v[2] + im[k].z; v[3] = v[3] + im[k].w;
            v[0] = v[0] + ob[k].x; v[1] = v[1] + ob[k].y; v[2] = v[2] + ob[k].z; v[3] = v[3] + ob[k].w;
            if (thr) {
                bool kp[4];
                drop_keep4(seed, (uint32_t)(((size_t)b * drop_rows + drop_row0 + i) * d + c), thr, kp);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = kp[e] ? v[e] * dsc : 0.f;
            }
            *reinterpret_cast<float4*>(dst + c) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}
__global__ __launch_bounds__(256) void vis_bwd_rows_vec_kernel(const float* __restrict__ dout, long long sb, long long st,
                                                               const float* __restrict__ G, const float* __restrict__ boxes,
                                                               const float* __restrict__ Wp, const float* __restrict__ bp,
                                                               const float* __restrict__ lnf_w, const float* __restrict__ lnp_w,
                                                               const float* __restrict__ rstd_f, const float* __restrict__ rstd_p,
                                                               bf16_t* __restrict__ dG, float* __restrict__ coef_p, float* __restrict__ coef_f,
                                                               int B, int V, int d, uint32_t thr, uint32_t seed, int drop_rows,
                                                               int drop_row0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * V) return;
    const int b = row / V, i = row % V;
    const float* gr = G + (size_t)row * d;
    const float* go = dout + b * sb + i * st;
    float p5[5];
    load_pos5(boxes, row, p5);
    const float rf = rstd_f[row], rp = rstd_p[row];
    const float dsc = drop_scale(thr);
    float4 g[8], gv[8], a[8], wf[8], wp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = lane * 4 + k * 256;
        if (c < d) {
            g[k] = *reinterpret_cast<const float4*>(go + c);
            gv[k] = *reinterpret_cast<const float4*>(gr + c);
            a[k] = pos_lin4(Wp, bp, c, p5);
            wf[k] = *reinterpret_cast<const float4*>(lnf_w + c);
            wp[k] = *reinterpret_cast<const float4*>(lnp_w + c);
        }
    }
    float sf = 0.f, sp = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = lane * 4 + k * 256;
        if (c < d) {
            if (thr) {
                bool kp[4];
                drop_keep4(seed, (uint32_t)(((size_t)b * drop_rows + drop_row0 + i) * d + c), thr, kp);
                g[k].x = kp[0] ? g[k].x * dsc : 0.f; g[k].y = kp[1] ? g[k].y * dsc : 0.f;
                g[k].z = kp[2] ? g[k].z * dsc : 0.f; g[k].w = kp[3] ? g[k].w * dsc : 0.f;
            }
            sf += g[k].x * wf[k].x * gv[k].x; sf += g[k].y * wf[k].y * gv[k].y; sf += g[k].z * wf[k].z * gv[k].z; sf += g[k].w * wf[k].w * gv[k].w;
            sp += g[k].x * wp[k].x * a[k].x; sp += g[k].y * wp[k].y * a[k].y; sp += g[k].z * wp[k].z * a[k].z; sp += g[k].w * wp[k].w * a[k].w;
        }
    }
    sf = wave_sum(sf);
    sp = wave_sum(sp);
    const float cf = rf * rf * rf * sf / (float)d;
    if (lane == 0) { coef_p[row] = rp * rp * rp * sp / (float)d; coef_f[row] = cf; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = lane * 4 + k * 256;
        if (c < d) {
            uint2 pk;
            pk.x = pack_bf16x2(rf * g[k].x * wf[k].x - gv[k].x * cf, rf * g[k].y * wf[k].y - gv[k].y * cf);
            pk.y = pack_bf16x2(rf * g[k].z * wf[k].z - gv[k].z * cf, rf * g[k].w * wf[k].w - gv[k].w * cf);
            *reinterpret_cast<uint2*>(dG + (size_t)row * d + c) = pk;
        }
    }
}

// column-wise part: parameter gradients, rows split over blockIdx.y, partial [split][9*d]
__global__ void vis_bwd_cols_kernel(const float* __restrict__ dout, long long sb, long long st, const float* __restrict__ G,
                                    const float* __restrict__ boxes, const float* __restrict__ Wp, const float* __restrict__ bp,
                                    const float* __restrict__ lnf_w, const float* __restrict__ lnp_w,
                                    const float* __restrict__ rstd_f, const float* __restrict__ rstd_p,
                                    const float* __restrict__ coef_p, const float* __restrict__ coef_f,
                                    float* __restrict__ partial, int B, int V, int d, int rows_per_split, uint32_t thr,
                                    uint32_t seed, int drop_rows, int drop_row0) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    const int r0 = blockIdx.y * rows_per_split, r1 = min(B * V, r0 + rows_per_split);
    const float dsc = drop_scale(thr);
    float a_lnf = 0.f, a_lnp = 0.f, a_bp = 0.f, a_img = 0.f, a_bf = 0.f, a_w[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const float wl = lnp_w[c], wf = lnf_w[c];
    for (int row = r0; row < r1; ++row) {
        const int b = row / V, i = row % V;
        float g = dout[b * sb + i * st + c];
        if (thr) {
            uint32_t idx = (uint32_t)(((size_t)b * drop_rows + drop_row0 + i) * d + c);
            g = drop_keep(seed, idx, thr) ? g * dsc : 0.f;
        }
        float p5[5];
        load_pos5(boxes, row, p5);
        const float a = pos_lin(Wp, bp, c, p5);
        const float rf = rstd_f[row], rp = rstd_p[row];
        const float gv = G[(size_t)row * d + c];
        a_lnf += g * gv * rf;
        a_bf += rf * g * wf - gv * coef_f[row];
        a_lnp += g * a * rp;
        a_img += g;
        const float da = rp * g * wl - a * coef_p[row];
        a_bp += da;
#pragma unroll
        for (int k = 0; k < 5; ++k) a_w[k] += da * p5[k];
    }
    float* pp = partial + (size_t)blockIdx.y * 10 * d;
    pp[c] = a_lnf;
    pp[d + c] = a_lnp;
    pp[2 * d + c] = a_bp;
#pragma unroll
    for (int k = 0; k < 5; ++k) pp[3 * d + c * 5 + k] = a_w[k];
    pp[8 * d + c] = a_img;
    pp[9 * d + c] = a_bf;
}

// dshared[vocab-1-i, c] += sum_b dout[b, i, c]   (fixed order over b: four interleaved sums s_g = sum of b = g, g + 4, ... in ascending
// order, combined as (s0 + s1) + (s2 + s3)).  64 columns x the four sums per workgroup: wave g owns sum g and has ALL its loads in flight
// at once (B / 4 per lane) -- with one thread per column walking b in steps of four the 216 two-wave workgroups of B = 80, V = 36 ran ten
// dependent round trips on an almost empty chip (21 us).
__global__ __launch_bounds__(256) void vis_bwd_obj_kernel(const float* __restrict__ dout, long long sb, long long st, float* __restrict__ dshared,
                                                          int B, int V, int d, int vocab, uint32_t thr, uint32_t seed, int drop_rows, int drop_row0) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int i = blockIdx.y;
    const float dsc = drop_scale(thr);
    float s = 0.f;
    if (c < d) {
        for (int b0 = g; b0 < B; b0 += 4 * 8) {                  // eight of this sum's terms per round (all of them for B <= 32)
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * 4;
                t[u] = b < B ? dout[b * sb + i * st + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + u * 4;
                if (b < B) {
                    float gq = t[u];
                    if (thr) {
                        uint32_t idx = (uint32_t)(((size_t)b * drop_rows + drop_row0 + i) * d + c);
                        gq = drop_keep(seed, idx, thr) ? gq * dsc : 0.f;
                    }
                    s += gq;
                }
            }
        }
    }
    part[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < d) {
        const int l = threadIdx.x & 63;
        dshared[(size_t)(vocab - 1 - i) * d + c] += (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
    }
}

// reduced column sums [10*d] (layout of one vis_bwd_cols partial row) -> the six parameter gradients; rows >= 1 of the
// img_order_embedding gradient are zero (only row 0 is ever looked up, src/modeling_t5_our.py:121-124)
__global__ void vis_grad_scatter_kernel(const float* __restrict__ red, float* __restrict__ g_lnf, float* __restrict__ g_lnp,
                                        float* __restrict__ g_bp, float* __restrict__ g_wp, float* __restrict__ g_img,
                                        float* __restrict__ g_bf, int d, int n_images) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d) { g_lnf[i] = red[i]; g_lnp[i] = red[d + i]; g_bp[i] = red[2 * d + i]; g_bf[i] = red[9 * d + i]; }
    if (i < 5 * d) g_wp[i] = red[3 * d + i];
    if (i < n_images * d) g_img[i] = i < d ? red[8 * d + i] : 0.f;
}

// ---------------- relative position bias ----------------------------------------------------------
__global__ void relbias_build_kernel(const float* __restrict__ table, const int* __restrict__ lut, float* __restrict__ bias,
                                     int H, int Lq, int Lk) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * Lq * Lk) return;
    int h = i / (Lq * Lk), pos = i % (Lq * Lk);
    bias[i] = table[lut[pos] * H + h];
}
// R[g][i] = sum over the matrices of group g (blockIdx.y) of dS[m][i]; second stage sums the groups in a fixed order
__global__ void relbias_sum_mats_kernel(const float* __restrict__ dS, float* __restrict__ R, int nmat, int n, int per_group) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int m0 = blockIdx.y * per_group, m1 = min(nmat, m0 + per_group);
    float s = 0.f;
    int m = m0;
    for (; m + 7 < m1; m += 8) {                     // eight loads in flight, summed in matrix order
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = dS[(size_t)(m + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; m < m1; ++m) s += dS[(size_t)m * n + i];
    R[(size_t)blockIdx.y * n + i] = s;
}
// one workgroup per (bucket, head): lanes stride over the positions, the four waves over the partial matrices; fixed order
__global__ __launch_bounds__(256) void relbias_scatter_kernel(const float* __restrict__ R, const int* __restrict__ lut,
                                                              float* __restrict__ dtable, int H, int npos, int nbuckets, int ngroups,
                                                              int accum) {
    __shared__ float part[4];
    const int i = blockIdx.x;                 // bucket * H + h
    const int bucket = i / H, h = i % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    // (the bucket ids of eight positions are requested together: one id per dependent round trip made the 49 iterations of a lane at
    // 56 x 56 positions 49 round trips for ~1.5 hits; positions and groups are still visited in ascending order)
    for (int pos0 = lane; pos0 < npos; pos0 += 8 * 64) {
        int id[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) id[u] = lut[min(pos0 + u * 64, npos - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pos = pos0 + u * 64;
            if (pos < npos && id[u] == bucket) {
                int g = wave;
                for (; g + 28 < ngroups; g += 32) {              // eight of this wave's groups per round (all sixteen of 64 groups in two)
                    float t[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) t[q] = R[((size_t)(g + 4 * q) * H + h) * npos + pos];
#pragma unroll
                    for (int q = 0; q < 8; ++q) s += t[q];
                }
                for (; g < ngroups; g += 4) s += R[((size_t)g * H + h) * npos + pos];
            }
        }
    }
    s = wave_sum(s);
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (part[0] + part[1]) + (part[2] + part[3]);
        dtable[i] = accum ? dtable[i] + t : t;
    }
}

// ---------------- the inputs of a T5 stack in one launch ---------------------------------------------
// key mask + relative-position bias block + token embeddings (+ the decoder's shift-right): four launch-latency-bound kernels of
// ~5 us each on the step's dependency chain become one.  Block ranges: [0, nM) mask, [nM, nM + nR) bias, then one block per token row.
// Element for element the arithmetic of build_mask_kernel / relbias_build_kernel / shift_right_kernel / embed_fwd_kernel.
struct StackIn {
    const long long* mask_ids; float* mask; int B, L, S;
    const float* rel_table; const int* lut; float* bias; int H, Lq, Lk;
    const long long* ids; const long long* labels; long long* ids_out; int T, start_id, pad_id;
    const float* table; float* out; long long out_sb, out_st; int d, vocab; uint32_t thr, seed; int drop_rows, drop_row0;
    int nM, nR;
};
__global__ __launch_bounds__(256) void stack_inputs_kernel(StackIn a) {
    int blk = blockIdx.x;
    if (blk < a.nM) {
        const int i = blk * 256 + threadIdx.x;
        if (i < a.B * a.S) {
            const int b = i / a.S, s = i % a.S;
            a.mask[i] = (s < a.L) ? (a.mask_ids[b * a.L + s] != a.pad_id ? 1.f : 0.f) : 1.f;
        }
        return;
    }
    blk -= a.nM;
    if (blk < a.nR) {
        const int i = blk * 256 + threadIdx.x;
        if (i < a.H * a.Lq * a.Lk) {
            const int h = i / (a.Lq * a.Lk), pos = i % (a.Lq * a.Lk);
            a.bias[i] = a.rel_table[a.lut[pos] * a.H + h];
        }
        return;
    }
    const int row = blk - a.nR, b = row / a.T, t = row % a.T;
    long long id;
    if (a.labels) {                                  // decoder: HF _shift_right of the labels, kept for the backward's scatter
        id = (t == 0) ? (long long)a.start_id : a.labels[row - 1];
        if (id == -100) id = a.pad_id;
        if (threadIdx.x == 0) a.ids_out[row] = id;
    } else {
        id = a.ids[row];
    }
    if (id < 0 || id >= a.vocab) id = 0;
    const float* src = a.table + (size_t)id * a.d;
    float* dst = a.out + b * a.out_sb + t * a.out_st;
    const float dsc = drop_scale(a.thr);
    for (int c = threadIdx.x * 4; c < a.d; c += 256 * 4) {
        const float4 v = *reinterpret_cast<const float4*>(src + c);
        float o[4] = {v.x, v.y, v.z, v.w};
        if (a.thr) {
            const uint32_t idx = (uint32_t)(((size_t)b * a.drop_rows + a.drop_row0 + t) * a.d + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = drop_keep(a.seed, idx + k, a.thr) ? o[k] * dsc : 0.f;
        }
        *reinterpret_cast<float4*>(dst + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace

#define ST ((hipStream_t)stream)
static inline uint32_t thr_of(float p) { return p > 0.f ? drop_thr16(p) : 0u; }

extern "C" int vlt5_embed_fwd(const long long* ids, const float* table, float* out, long long out_sb, long long out_st, int B,
                              int T, int d, int vocab, float drop_p, uint32_t drop_seed, int drop_rows, int drop_row0,
                              void* stream) {
    if (!ids || !table || !out || B <= 0 || T <= 0) return VLT5_ERR_ARG;
    if ((d & 3) || (out_sb & 3) || (out_st & 3)) return VLT5_ERR_ALIGN;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(B * T), dim3(d >= 1024 ? 256 : (d >= 512 ? 128 : 64)), 0, ST, ids, table, out,
                       out_sb, out_st, B, T, d, vocab, thr_of(drop_p), drop_seed, drop_rows, drop_row0);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" long long vlt5_embed_bwd_scratch_bytes(int B, int T, int d) {
    if (B < 1 || T < 1 || d < 1) return -1;
    const long long n = (long long)B * T;
    return (n * d * 4 + 255) / 256 * 256 + n * 16 + n * 4;        // leaders' partial sums, per-row index records, the sorted order
}
extern "C" int vlt5_embed_bwd(const long long* ids, const float* dout, long long sb, long long st, float* dtable, int B, int T,
                              int d, int vocab, float drop_p, uint32_t drop_seed, int drop_rows, int drop_row0, void* scratch,
                              void* stream) {
    if (!ids || !dout || !dtable || !scratch || B <= 0 || T <= 0) return VLT5_ERR_ARG;
    if ((d & 3) || (sb & 3) || (st & 3) || (((uintptr_t)scratch) & 15)) return VLT5_ERR_ALIGN;
    const long long n = (long long)B * T;
    float* partial = (float*)scratch;
    EmbIdx* idx = (EmbIdx*)((char*)scratch + (n * d * 4 + 255) / 256 * 256);
    int* sorted = (int*)(idx + n);
    const dim3 grid((unsigned)n, (d + 255) / 256);
    hipLaunchKernelGGL(embed_index_kernel, dim3((unsigned)n), dim3(64), 0, ST, ids, idx, sorted, (int)n, vocab);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(embed_bwd_pass1_kernel, grid, dim3(64), 0, ST, dout, sb, st, (const EmbIdx*)idx, (const int*)sorted, ids, partial, dtable,
                       T, d, vocab, thr_of(drop_p), drop_seed, drop_rows, drop_row0);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(embed_bwd_pass2_kernel, grid, dim3(64), 0, ST, (const EmbIdx*)idx, (const int*)sorted, ids, (const float*)partial, dtable, d,
                       vocab);
    LAUNCH_CHECK();
    return VLT5_OK;
}
// bf16 staging mirror of the rows of a scatter-added gradient table (data-parallel bf16 buckets): one workgroup per listed row
__global__ __launch_bounds__(64) void mirror_rows_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int vocab, int d,
                                                         const long long* __restrict__ ids0, int n0, const long long* __restrict__ ids1, int n1,
                                                         int tail_rows) {
    const int j = blockIdx.x;
    long long row;
    if (j < n0) row = emb_clamp(ids0[j], vocab);
    else if (j < n0 + n1) row = emb_clamp(ids1[j - n0], vocab);
    else row = vocab - tail_rows + (j - n0 - n1);
    const float* s = src + (size_t)row * d;
    bf16_t* o = dst + (size_t)row * d;
    for (int c = threadIdx.x * 4; c < d; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(s + c);
        *reinterpret_cast<uint2*>(o + c) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
}
extern "C" int vlt5_mirror_rows_bf16(const float* src, void* dst_bf16, int vocab, int d, const long long* ids0, int n0,
                                     const long long* ids1, int n1, int tail_rows, void* stream) {
    if (!src || !dst_bf16 || vocab < 1 || d < 4 || n0 < 0 || n1 < 0 || tail_rows < 0 || tail_rows > vocab) return VLT5_ERR_ARG;
    if ((n0 > 0 && !ids0) || (n1 > 0 && !ids1)) return VLT5_ERR_ARG;
    if ((d & 3) || (((uintptr_t)src) & 15) || (((uintptr_t)dst_bf16) & 7)) return VLT5_ERR_ALIGN;
    const int rows = n0 + n1 + tail_rows;
    if (rows == 0) return VLT5_OK;
    hipLaunchKernelGGL(mirror_rows_kernel, dim3(rows), dim3(64), 0, ST, src, (bf16_t*)dst_bf16, vocab, d, ids0, n0, ids1, n1, tail_rows);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_shift_right(const long long* labels, long long* out, int B, int T, int start_id, int pad_id, void* stream) {
    if (!labels || !out || B <= 0 || T <= 0) return VLT5_ERR_ARG;
    hipLaunchKernelGGL(shift_right_kernel, dim3((B * T + 255) / 256), dim3(256), 0, ST, labels, out, B, T, start_id, pad_id);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_build_mask(const long long* ids, float* mask, int B, int L, int S, int pad_id, void* stream) {
    if (!ids || !mask || B <= 0 || L < 0 || S < L) return VLT5_ERR_ARG;
    hipLaunchKernelGGL(build_mask_kernel, dim3((B * S + 255) / 256), dim3(256), 0, ST, ids, mask, B, L, S, pad_id);
    LAUNCH_CHECK();
    return VLT5_OK;
}

// key mask + relative-position bias block + token embeddings of one T5 stack (for the decoder: of the shift-right of the labels)
// in ONE launch -- see stack_inputs_kernel
extern "C" int vlt5_stack_inputs_fwd(const vlt5_stack_inputs_desc* s, void* stream) {
    if (!s || !s->mask_ids || !s->mask || !s->rel_table || !s->lut || !s->bias || !s->table || !s->out) return VLT5_ERR_ARG;
    if ((!s->ids && !s->labels) || (s->labels && !s->ids_out)) return VLT5_ERR_ARG;
    if (s->B <= 0 || s->L < 0 || s->S < s->L || s->H <= 0 || s->Lq <= 0 || s->Lk <= 0 || s->T <= 0) return VLT5_ERR_ARG;
    if ((s->d & 3) || (s->out_sb & 3) || (s->out_st & 3)) return VLT5_ERR_ALIGN;
    StackIn a;
    a.mask_ids = s->mask_ids; a.mask = s->mask; a.B = s->B; a.L = s->L; a.S = s->S;
    a.rel_table = s->rel_table; a.lut = s->lut; a.bias = s->bias; a.H = s->H; a.Lq = s->Lq; a.Lk = s->Lk;
    a.ids = s->ids; a.labels = s->labels; a.ids_out = s->ids_out; a.T = s->T; a.start_id = s->start_id; a.pad_id = s->pad_id;
    a.table = s->table; a.out = s->out; a.out_sb = s->out_sb; a.out_st = s->out_st; a.d = s->d; a.vocab = s->vocab;
    a.thr = thr_of(s->drop_p); a.seed = s->drop_seed; a.drop_rows = s->drop_rows; a.drop_row0 = s->drop_row0;
    a.nM = (s->B * s->S + 255) / 256; a.nR = (s->H * s->Lq * s->Lk + 255) / 256;
    hipLaunchKernelGGL(stack_inputs_kernel, dim3(a.nM + a.nR + s->B * s->T), dim3(256), 0, ST, a);
    LAUNCH_CHECK();
    return VLT5_OK;
}

// the 16-byte forms of the two row kernels: d and the row strides multiples of 4 words, d <= 2048, 16-byte aligned bases
static bool vis_vec_ok(int d, long long sb, long long st, std::initializer_list<const void*> ptrs) {
    if ((d & 3) || d > 2048 || (sb & 3) || (st & 3)) return false;
    for (const void* q : ptrs) if (reinterpret_cast<uintptr_t>(q) & 15) return false;
    return true;
}
extern "C" int vlt5_vis_embed_fwd(const float* G, const float* boxes, const float* Wp, const float* bp, const float* lnf_w,
                                  const float* lnp_w, const float* img0, const float* shared, float* out, long long out_sb,
                                  long long out_st, float* rstd_f, float* rstd_p, int B, int V, int d, int vocab, float eps,
                                  float drop_p, uint32_t drop_seed, int drop_rows, int drop_row0, void* stream) {
    if (!G || !boxes || !Wp || !bp || !lnf_w || !lnp_w || !img0 || !shared || !out || !rstd_f || !rstd_p) return VLT5_ERR_ARG;
    if (B <= 0 || V <= 0 || V > vocab) return VLT5_ERR_ARG;
    const bool vec = vis_vec_ok(d, out_sb, out_st, {G, Wp, bp, lnf_w, lnp_w, img0, shared, out});
    hipLaunchKernelGGL(vec ? vis_fwd_vec_kernel : vis_fwd_kernel, dim3((B * V + 3) / 4), dim3(256), 0, ST, G, boxes, Wp, bp, lnf_w, lnp_w, img0, shared, out,
                       out_sb, out_st, rstd_f, rstd_p, B, V, d, vocab, eps, thr_of(drop_p), drop_seed, drop_rows, drop_row0);
    LAUNCH_CHECK();
    return VLT5_OK;
}
// row splits of the column-wise pass: ~16 rows per thread (the per-row chain of loads is latency-bound), at most 256
extern "C" int vlt5_vis_embed_bwd_blocks(int rows) {
    int s = (rows + 15) / 16;
    return s < 1 ? 1 : (s > 256 ? 256 : s);
}
extern "C" int vlt5_vis_embed_bwd(const float* dout, long long sb, long long st, const float* G, const float* boxes,
                                  const float* Wp, const float* bp, const float* lnf_w, const float* lnp_w, const float* rstd_f,
                                  const float* rstd_p, void* dG_bf16, float* partial, float* dshared, int B, int V, int d,
                                  int vocab, float drop_p, uint32_t drop_seed, int drop_rows, int drop_row0, void* stream) {
    if (!dout || !G || !boxes || !Wp || !bp || !lnf_w || !lnp_w || !rstd_f || !rstd_p || !dG_bf16 || !partial || !dshared)
        return VLT5_ERR_ARG;
    const int rows = B * V;
    const int nsplit = vlt5_vis_embed_bwd_blocks(rows);
    const int rps = (rows + nsplit - 1) / nsplit;
    float* coef_p = partial + (size_t)nsplit * 10 * d;            // 2 x [rows] scratch behind the partials
    float* coef_f = coef_p + rows;
    uint32_t thr = thr_of(drop_p);
    const bool vec = vis_vec_ok(d, sb, st, {dout, G, Wp, bp, lnf_w, lnp_w, dG_bf16});
    hipLaunchKernelGGL(vec ? vis_bwd_rows_vec_kernel : vis_bwd_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, ST, dout, sb, st, G, boxes, Wp, bp, lnf_w, lnp_w,
                       rstd_f, rstd_p, (bf16_t*)dG_bf16, coef_p, coef_f, B, V, d, thr, drop_seed, drop_rows, drop_row0);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(vis_bwd_cols_kernel, dim3((d + 127) / 128, nsplit), dim3(128), 0, ST, dout, sb, st, G, boxes, Wp, bp, lnf_w, lnp_w,
                       rstd_f, rstd_p, coef_p, coef_f, partial, B, V, d, rps, thr, drop_seed, drop_rows, drop_row0);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(vis_bwd_obj_kernel, dim3((d + 63) / 64, V), dim3(256), 0, ST, dout, sb, st, dshared, B, V, d, vocab, thr,
                       drop_seed, drop_rows, drop_row0);
    LAUNCH_CHECK();
    return VLT5_OK;
}

extern "C" int vlt5_vis_grad_scatter(const float* reduced, float* g_lnf, float* g_lnp, float* g_bp, float* g_wp, float* g_img,
                                     float* g_bf, int d, int n_images, void* stream) {
    if (!reduced || !g_lnf || !g_lnp || !g_bp || !g_wp || !g_img || !g_bf || d <= 0 || n_images <= 0) return VLT5_ERR_ARG;
    const int n = (n_images > 5 ? n_images : 5) * d;
    hipLaunchKernelGGL(vis_grad_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, reduced, g_lnf, g_lnp, g_bp, g_wp, g_img,
                       g_bf, d, n_images);
    LAUNCH_CHECK();
    return VLT5_OK;
}

extern "C" int vlt5_relbias_build(const float* table, const int* lut, float* bias, int H, int Lq, int Lk, int nbuckets,
                                  void* stream) {
    if (!table || !lut || !bias || H <= 0 || Lq <= 0 || Lk <= 0 || nbuckets <= 0) return VLT5_ERR_ARG;
    int n = H * Lq * Lk;
    hipLaunchKernelGGL(relbias_build_kernel, dim3((n + 255) / 256), dim3(256), 0, ST, table, lut, bias, H, Lq, Lk);
    LAUNCH_CHECK();
    return VLT5_OK;
}
extern "C" int vlt5_relbias_bwd(const float* dS, const int* lut, float* dtable, float* scratch, int nmat, int H, int Lq, int Lk,
                                int nbuckets, int accum, void* stream) {
    if (!dS || !lut || !dtable || !scratch || nmat <= 0) return VLT5_ERR_ARG;
    int n = H * Lq * Lk;
    const int ngroups = nmat < 64 ? nmat : 64;                 // scratch holds ngroups partial matrices
    const int per_group = (nmat + ngroups - 1) / ngroups;
    hipLaunchKernelGGL(relbias_sum_mats_kernel, dim3((n + 255) / 256, ngroups), dim3(256), 0, ST, dS, scratch, nmat, n, per_group);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(relbias_scatter_kernel, dim3(H * nbuckets), dim3(256), 0, ST, scratch, lut, dtable, H, Lq * Lk,
                       nbuckets, (nmat + per_group - 1) / per_group, accum);
    LAUNCH_CHECK();
    return VLT5_OK;
}
