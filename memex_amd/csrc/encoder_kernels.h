// encoder_kernels.h -- device-side contract of the sentence encoder (kernels in
// encoder_kernels.hip, host logic in encoder.hip).
//
// Token layout: sequences are packed back to back ("varlen"), each sequence start aligned to 8
// tokens so that 16-byte vector accesses along the token axis stay aligned; the packed length is
// padded to a multiple of 128 rows (GEMM tile height).  Rows that belong to no sequence hold
// finite garbage and are never read by attention (masked by length) or pooling.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mx {

typedef __bf16 bf16_t;

constexpr int kSeqAlign = 8;     // sequence start alignment (tokens)
constexpr int kRowPad = 256;     // packed token count is padded to this (GEMM BM)

enum GemmEpilogue {
    EPI_BIAS = 0,         // out = bf16(acc + bias)
    EPI_BIAS_GELU = 1,    // out = bf16(gelu_erf(acc + bias))
    EPI_QKV = 2,          // N = 2H: q (pre-scaled) and k, token-major into out / out_k
    EPI_VT = 4,           // N = H: v, feature-major (transposed) into out_vt
    EPI_BIAS_RES_LN = 3,  // out = bf16(LayerNorm(acc + bias + residual))   (BN == N == hidden)
    EPI_BIAS_RES = 5,     // out = bf16(bf16(acc + bias) + residual): pgemm_kernel only, launch_ln_rows applies the LayerNorm
    // the bf16x3 ("precise") encoder, encoder_precise.hip: operands arrive split, results leave in f32 or split again
    EPI_F32 = 6,          // out_f32 = acc + bias                              ([M, N] f32, row pitch ldo)
    EPI_GELU_SPLIT = 7,   // out = split3(gelu_erf(acc + bias))                ([M, 3N] bf16, row pitch ldo: hi | lo | hi)
    // the mixed mode (MX_PREC_MIXED): the MLP's two GEMMs as TWO fp16 products per product -- the activation as fp16 hi + lo
    // ([hi | lo], width 2K), the weight as ONE fp16 value ([w | w]) -- on v_mfma_f32_32x32x16_f16
    EPI_F32_H = 8,        // EPI_F32 on fp16 operands
    EPI_GELU_SPLIT_H = 9, // out = split2h(gelu_erf(acc + bias))               ([M, 2N] fp16, row pitch ldo: hi | lo)
};

struct GemmParams {
    const bf16_t *a;      // [M, K] activations, row pitch lda (elements)
    int lda;
    const bf16_t *w;      // weights, K-blocked: [K/32][w_rows][32] (element (n,k) at ((k>>5)*w_rows + n)*32 + (k&31))
    int w_rows;           // rows of the blocked weight matrix (e.g. 3H for the concatenated QKV)
    int w_row0;           // first weight row this GEMM uses (V third: 2H)
    const float *bias;    // [N]
    int m, n, k;          // m multiple of 128 (or 64 for the 768-wide LN variant), n multiple of 384, k of 64
    bf16_t *out;          // EPI_BIAS/GELU/LN: [M, N] pitch ldo;  EPI_QKV: q [M, H]
    int ldo;
    bf16_t *out_k;        // EPI_QKV: k [M, H]
    bf16_t *out_vt;       // EPI_QKV: v^T [H, ldvt]
    int ldvt;
    int hidden;           // EPI_QKV / EPI_VT: H
    float qscale;         // EPI_QKV: multiplies q (1/sqrt(d_head) * log2(e))
    const bf16_t *res;    // EPI_BIAS_RES_LN: residual [M, N] pitch ldres
    int ldres;
    const float *gamma;   // EPI_BIAS_RES_LN
    const float *beta;
    float eps;
    float *out_f32;       // EPI_F32
    int single;           // EPI_GELU_SPLIT_H only: 1 = the output is ONE fp16 value per element (no lo block: MX_PREC_MIXED1), 0 = [hi | lo]
    int ksplit;           // EPI_F32 only, 0 / 1 = off: the launch has `ksplit` k-chunks of p.k columns each (blockIdx.y = chunk): chunk z reads
                          // a + z k and the weights' k-blocks from z k / 32 on, and writes its partial product to out_f32 + z m ldo
};

// the layer tail (encoder_tail.hip), hidden = 384:
//     x1 = LayerNorm1(x + Wo ctx + bo);  out = LayerNorm2(x1 + W2 gelu(W1 x1 + b1) + b2)
// ctx == nullptr: the MLP block alone with x = x1
struct TailParams {
    const bf16_t *ctx;    // [M, 384] attention output, row pitch ldc (nullptr: MLP block only)
    int ldc;
    const bf16_t *x;      // [M, 384] layer input = residual of LayerNorm1 (MLP only: x1), row pitch ldx
    int ldx;
    const bf16_t *wf;     // Wo, W1, W2 as one bf16 stream per wave in consumption order (tail_stream_layout)
    const float *bo;      // [384] out-projection bias
    const float *ln1g, *ln1b;
    const float *b1;      // [f]
    const float *b2;      // [384]
    int f;                // ffn width, multiple of 128, 256 .. 1536
    int m;                // rows, multiple of 64
    bf16_t *out;          // [M, 384] pitch ldo (may alias x: a workgroup reads its 64 rows before it writes them)
    int ldo;
    const float *gamma, *beta;  // LayerNorm2
    float eps;
    // start-up skew (speed only): workgroups b < skew_hi with bit skew_shift of b set sleep skew_iters x ~4 us
    // before they start, so that the two workgroups of a CU stop running their load / LayerNorm / store phases
    // at the same time
    int skew_shift, skew_hi, skew_iters;
    unsigned long long *trace;  // scripts/tail_ubench.hip only (MX_TAIL_TRACE): [blocks][8] phase timestamps
    int stop_after_ln1;         // ctx != nullptr only: write x1 = LayerNorm1(x + Wo ctx + bo) to `out` and stop (small passes)
};
hipError_t tail_setup();
bool tail_supported(int hidden, int ffn);
hipError_t launch_tail(hipStream_t s, const TailParams &p);
// wo [384][384], w1 [f][384], w2 [384][f] (nn.Linear layouts, f32) -> out [tail_stream_elems(f)] bf16
size_t tail_stream_elems(int F);
void tail_stream_layout(const float *wo, const float *w1, const float *w2, int F, uint16_t *out, uint16_t (*to_bf16)(float));

hipError_t encoder_kernels_setup();
hipError_t launch_gemm(hipStream_t s, int epi, const GemmParams &p);

// the large-pass GEMM (encoder_pgemm.hip): persistent 256 x 256 tiles, two wave rows running half a phase apart
hipError_t pgemm_setup();
bool pgemm_supported(int epi, const GemmParams &p);
hipError_t launch_pgemm(hipStream_t s, int epi, const GemmParams &p);
// x[r] = LayerNorm(x[r]) * gamma + beta in place, rows of `hidden` (384 / 768) bf16, row pitch ld
// out[r] = LayerNorm(bf16(sum_z part[z][r] + bias) + res[r]) for rows < m, hidden n in {384, 768}: closes a split-k GEMM (small passes of
// the hidden-768 models: encoder.hip)
hipError_t launch_reduce_res_ln(hipStream_t s, const float *part, int nsplit, int m, int n, const float *bias, const bf16_t *res, int ldres,
                                const float *gamma, const float *beta, float eps, bf16_t *out, int ldo);
// q | k | v^T from the f32 partials of a split-k QKV product [nsplit][m][3 hidden]: q = bf16((sum + b) qscale), k = bf16(sum + b) token-major,
// v^T = bf16(sum + b) feature-major [hidden][ldvt] -- the roundings of gemm_kernel's EPI_QKV / EPI_VT epilogues
hipError_t launch_reduce_qkv(hipStream_t s, const float *part, int nsplit, int m, int hidden, const float *bias, float qscale, bf16_t *q,
                             bf16_t *k, bf16_t *vt, int ldvt);
hipError_t launch_ln_rows(hipStream_t s, bf16_t *x, int ld, int rows, int hidden, const float *gamma, const float *beta, float eps);

// token maps from sequence lengths: cu[b] (aligned starts), tok_seq / tok_pos for every packed row
// (+ the attention work list of the pass, see launch_attention)
// `max_len`: the longest sequence of the pass (host side); it decides the head grouping of the attention work list
hipError_t launch_token_map(hipStream_t s, const int32_t *lens, int B, int S, int32_t *cu, int32_t *tok_seq,
                            int32_t *tok_pos, int t_pad, int heads, int d_head, int max_len, void *attn_plan);

// x[t] = LayerNorm(word[id] + pos[p] + type[0])
hipError_t launch_embed_ln(hipStream_t s, const int32_t *ids, int S, const int32_t *tok_seq, const int32_t *tok_pos,
                           int t_pad, int hidden, const float *word, const float *pos, const float *type0,
                           const float *gamma, const float *beta, float eps, int vocab, bf16_t *x);

// softmax(q k^T + mask) v per (sequence, head); q is pre-scaled by 1/sqrt(d)*log2(e)
// `plan`: kAttnPlanBytesPerSeq bytes per sequence, written by launch_token_map once per pass (the work list: one
// item per (sequence, head group), longest sequences first) and read by every layer's launch_attention
constexpr size_t kAttnPlanBytesPerSeq = 16 * 16;
int attention_groups(int heads, int d_head, int max_len, int B);
hipError_t launch_attention(hipStream_t s, const bf16_t *q, const bf16_t *k, const bf16_t *vt, int ldvt, const void *plan, int B,
                            int heads, int d_head, int hidden, int max_len, bf16_t *ctx);

// masked mean (or CLS) over tokens + optional L2 normalise -> out [B, H] f32; xf != nullptr: the f32 hidden state instead of x
hipError_t launch_pool(hipStream_t s, const bf16_t *x, const float *xf, const int32_t *cu, const int32_t *lens, int B, int hidden,
                       int pooling_cls, int normalize, float *out);

// ---- small passes of the hidden-384 encoder (encoder_small.hip): query-time embedding.  m = padded packed rows (multiple
// of 64), rows = the packed rows that can hold tokens (the rest of the pass is padding and is not computed)
// passes of at most this many packed rows take the small-pass layer (encoder_small.hip).  Measured crossover against the bulk
// kernels (scripts/gpu_small_rows_sweep.py, profiles/r5_small_rows_sweep.txt): L12 at 128 tokens 1024 rows 0.69 vs 0.96 ms, 2048 rows
// 0.85 vs 0.97, 3072 rows 1.00 vs 1.06, 4096 rows 1.10 vs 1.07; L6 at 256 tokens 2048 rows 0.51 vs 0.59, 4096 rows 0.61 vs 0.59
constexpr int kSmallRows = 2048;
constexpr int kQueryRows = 512;   // a call of one pass up to this size is polled through by the host (0.2-0.5 ms: a nap would show)
hipError_t small_setup();
hipError_t launch_sp_qkv(hipStream_t s, const bf16_t *x, const bf16_t *wqkv, const float *bqkv, int m, int rows, float qscale, bf16_t *q,
                         bf16_t *k, bf16_t *vt, int ldvt);
// x1 = LayerNorm1(xres + ctx Wo^T + bo); wo in the GEMMs' K-blocked layout
hipError_t launch_sp_out_ln(hipStream_t s, const bf16_t *ctx, const bf16_t *xres, const bf16_t *wo, const float *bo, const float *gamma,
                            const float *beta, float eps, int m, int rows, bf16_t *x1);
// part [f / 128][m][384] f32: the MLP's partial products per ffn chunk (wf: tail_kernel's weight streams)
hipError_t launch_sp_ffn(hipStream_t s, const bf16_t *x1, const bf16_t *wf, const float *b1, int f, int m, int rows, float *part);
hipError_t launch_sp_reduce_ln(hipStream_t s, const float *part, int f, int m, int rows, const float *b2, const bf16_t *x1, const float *gamma,
                               const float *beta, float eps, bf16_t *out);

// ---- the bf16x3 ("precise") encoder (encoder_precise.hip): every GEMM operand travels as THREE bf16 column blocks
// [hi | lo | hi] (activations, width 3K) against [hi | hi | lo] (weights), so that the unchanged bf16 MFMA loop sums
// hi*hi + lo*hi + hi*lo -- 16 significant bits per operand instead of 8; the hidden state, the GEMM results and the whole
// attention core stay in f32.
hipError_t precise_setup();
// x[t] = LayerNorm(word[id] + pos[p] + type[0]) -> xf [t_pad, H] f32 and xs [t_pad, 3H] split
hipError_t launch_embed_ln_precise(hipStream_t s, const int32_t *ids, int S, const int32_t *tok_seq, const int32_t *tok_pos,
                                   int t_pad, int hidden, const float *word, const float *pos, const float *type0,
                                   const float *gamma, const float *beta, float eps, int vocab, float *xf, bf16_t *xs);
// xf[r] = LayerNorm(a[r] + xf[r]) in place (f32), xs[r] = split3(xf[r])
// half2 = 1: xs[r] = [hi | lo] in fp16 (2H wide) instead of the three bf16 blocks: the operand of the mixed mode's W1; 2: ONE fp16
// value per element (H wide): MX_PREC_MIXED1
hipError_t launch_add_ln_split(hipStream_t s, const float *a, float *xf, bf16_t *xs, int rows, int hidden, const float *gamma,
                               const float *beta, float eps, int half2 = 0);
// softmax(q k^T / sqrt(d) + mask) v in f32 (v_mfma_f32_32x32x2_f32) from qkv [t_pad, 3H] f32 (q | k | v) -> ctxs [t_pad, 3H] split
hipError_t launch_attention_f32(hipStream_t s, const float *qkv, const int32_t *cu, const int32_t *lens, int B, int S, int heads,
                                int d_head, int hidden, bf16_t *ctxs, bool f32_mfma, bool p_single);

}  // namespace mx
