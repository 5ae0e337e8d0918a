/* Experiment, NOT part of libvlt5_hip.so: the row-panel ("skinny") GEMM of tools/experiments/skinny.hip.
 * Built on demand into vqacl_amd/libvlt5_exp.so (make -C vqacl_amd/csrc exp) and driven by tools/skinny_probe.py / skinny_timeline.py.
 * Result (profiles/r03_skinny_probe.txt): at M = 400 rows it is no faster than the tiled kernel -- both sit on the same latency
 * floor (kernel arguments + first tile + 12 k-steps + stores landing); the fused norm prologue costs what a norm launch costs. */
#ifndef VLT5_SKINNY_H
#define VLT5_SKINNY_H
#include "vlt5_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* ---- row-panel ("skinny") GEMM for few-row activations (the decoder stack: M = B*T rows) ---------------------------
 * C[M,N] = epi(alpha * A W^T), W bf16 [N,K] row-major.  A is either bf16 [M,K] (A != NULL) or the T5 RMS norm of the f32 rows
 * ln_x [M,K] computed in the kernel's prologue: A = bf16(ln_x * rsqrt(mean(ln_x^2) + eps) * ln_w) -- HF T5LayerNorm.forward
 * followed by nn.Linear (T5LayerSelfAttention / T5LayerCrossAttention / T5LayerFF of the decoder T5Block,
 * VL-T5/src/modeling_t5_our.py:641-655); rstd_out [M] and xn_out_bf16 [M,K] (both optional) receive what the backward needs.
 * Epilogue: ReLU, inverted dropout on element index m*N+n (same counters as vlt5_gemm_bf16), f32 residual add (f32 output
 * only).  A workgroup keeps a panel of 16 / 32 rows of A in LDS and streams its weight rows into registers (csrc/skinny.hip).
 * K % 64 == 0, 16*K*2 bytes <= 144 KB, ln_x: K <= 1024.  panel_rows / chunk_cols: 0 = heuristic, else 16|32 and 64|128|192. */
typedef struct {
    const void* A; int lda;
    const float* ln_x; int ldx; const float* ln_w; float eps; float* rstd_out; void* xn_out_bf16;
    const void* W; int ldw;
    void* C; int ldc; int out_f32;
    int M, N, K;
    float alpha;
    int relu; float drop_p; uint32_t drop_seed;
    const float* resid; int ldr;
    int panel_rows, chunk_cols;
} vlt5_skinny_desc;
int vlt5_skinny_gemm(const vlt5_skinny_desc* d, void* stream);
int vlt5_skinny_ok(int M, int N, int K, int with_norm);     /* 1 if vlt5_skinny_gemm takes the shape */

#ifdef __cplusplus
}
#endif
#endif
