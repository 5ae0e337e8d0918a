// Internal interface between csrc/decode.hip (the kernels of a greedy-decoding step) and csrc/engine.hip (which composes them).
#pragma once
#include "common.h"
#include "vlt5_hip.h"
#include <string.h>

// out[m, n] = epi(rowscale[m] * alpha * sum_k A[m, k] W[n, k]) on `rows` (<= a few hundred) rows; see declin_kernel
struct DecLinArgs {
    const float* xf;            // A as f32 [rows, K] with the T5 RMS norm folded in (ln_w, eps): operand = bf16(x * ln_w), rows scaled by rstd
    const bf16_t* xb;           // ... or A as bf16 [rows, K]
    long long ldx;              // row stride of A in elements
    const float* ln_w; float eps;
    const bf16_t* W;            // [N, K] row-major
    int rows, N, K;
    float alpha;
    bf16_t* out_b; long long ldo;             // bf16 output, columns [0, split_col)
    int split_col; bf16_t* out_b2; long long ldo2;   // columns >= split_col go to out_b2 + t*t_stride2 + m*ldo2 + (n - split_col)
    float* out_f; long long ldf;              // f32 output
    const float* resid; long long ldr;        // f32 residual added before the activation / store
    int relu;
    float* pmax; int* pidx; int ptiles;       // per (row, column tile) first maximum and its column: [rows][CT]
    const int* t_ptr; long long t_stride2;    // optional device-side step index (for out_b2)
    // T5 RMS norm split between two launches (the decode step's chain): the PRODUCER of a residual-stream row also emits the operand of the
    // projection behind the next norm, bf16(v * nx_w[n]) (half the bytes of the f32 row, staged by LDS-DMA), and the sum of v^2 of every
    // (row, 16-column fragment) -- nx_parts = N / 16 per row; the CONSUMER (bf16 A) scales its rows by rsqrt(sum_j rs_part[m][j] / K + eps)
    const float* nx_w; bf16_t* nx_b; long long ld_nx; float* nx_ssq; int nx_parts;
    const float* rs_part; int rs_n;           // rs_n a multiple of 4, <= 64
    int* t_inc;                               // optional device-side step index incremented by this launch when it ends (t_ptr must be null)
    int force_nfrag;                          // > 0: column-tile width in 16-column fragments (vlt5_tuning.decode_nfrag; ignored with pmax)
    int RB, CT, ct_per_xcd;                   // filled by vlt5_declin_launch
    long long* tl;                            // -DDECLIN_TIMELINE builds: [workgroup][8] shader-clock stamps of wave 0 (tools/declin_timeline.py)
};

int vlt5_declin_launch(DecLinArgs a, hipStream_t st);
int vlt5_declin_tiles(int rows, int N, int K, int af32);
extern "C" int vlt5_decode_linear_supported(int K, int norm_folded);

// softmax(q K^T + bias + mask) V for ONE query per (sample, head); <= 64 keys, d_kv in {16, 32, 64}
struct DecCoreArgs {
    const bf16_t* q; long long q_ld;          // head h of sample b at q + b*q_ld + h*d_kv
    const bf16_t* k; const bf16_t* v; long long kv_sb, kv_st;   // key j at k + b*kv_sb + j*kv_st + h*d_kv
    bf16_t* ctx; long long ctx_ld;
    const float* bias; int bias_ld;           // [H][bias_ld] additive row of this query position, or null
    const float* key_mask; int mask_ld; float mask_value;       // [B][mask_ld] 1 = keep: adds (1 - m) * mask_value, or null
    int B, H, Tk;
    const int* t_ptr;                         // optional device-side step index: Tk = *t_ptr + 1
};
int vlt5_dec_core_launch(const DecCoreArgs& a, int d_kv, hipStream_t st);

// what happens between two steps, one workgroup per sample: [argmax finish + greedy bookkeeping of step t] + [input row and bias row of
// step t + 1]; every part optional
struct DecIoArgs {
    const float* pmax; const int* pidx; int ptiles;    // argmax partials of the vocabulary projection, or null (then `tokens` is the input)
    long long* next_ids;                      // [B] out: argmax
    int* done; int eos_id, pad_id;            // [B] in/out: HF greedy search flags (null: no bookkeeping)
    long long* out_tokens; long long out_ld; int out_col;       // emitted token -> out_tokens[b*out_ld + out_col]
    const long long* tokens;                  // [B] decoder input ids when there is no argmax to finish
    const float* table; int d, vocab; float* emb_out;           // emb_out[b] = table[token_b] (null: skip)
    const float* nx_w; bf16_t* nx_b; float* nx_ssq; int nx_parts;    // with emb_out: bf16(row * nx_w) and the row's sum of squares (in part 0, zeros behind it)
    const float* rel_table; const int* lut; int lut_ld, tq, H; float* bias_out; int bias_ld;   // bias_out[h][j] = rel_table[lut[tq][j]][h], j <= tq
    const int* t_ptr; int Tcap;               // optional device-side step index: out_col = tq = *t_ptr; the next input / bias row only while tq < Tcap
};
int vlt5_dec_io_launch(const DecIoArgs& a, int B, hipStream_t st);
