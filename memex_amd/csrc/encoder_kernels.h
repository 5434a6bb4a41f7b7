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
};

// fused MLP block (encoder_mlp.hip): out = LayerNorm(x + W2 gelu(W1 x + b1) + b2), hidden = 384
struct MlpParams {
    const bf16_t *x;      // [M, 384] input and residual, row pitch ldx
    int ldx;
    const bf16_t *w1;     // intermediate weights, K-blocked [384/32][f][32]
    const float *b1;      // [f]
    const bf16_t *w2;     // output weights, K-blocked [f/32][384][32]
    const float *b2;      // [384]
    int f;                // ffn width, multiple of 128
    int m;                // rows, multiple of 128
    bf16_t *out;          // [M, 384] pitch ldo
    int ldo;
    const float *gamma, *beta;
    float eps;
    // mlp2_kernel (encoder_mlp2.hip): W1 and W2 as one bf16 stream per wave in consumption order
    // (mlp2_stream_layout), 2 * f * 384 elements
    const bf16_t *wf;
};
hipError_t mlp_setup();
hipError_t mlp2_setup();
bool mlp2_supported(int hidden, int ffn);
hipError_t launch_mlp2(hipStream_t s, const MlpParams &p);
// w1 [f][384], w2 [384][f] (nn.Linear layouts, f32) -> out [2 * f * 384] bf16 in mlp2_kernel's stream order
void mlp2_stream_layout(const float *w1, const float *w2, int F, uint16_t *out, uint16_t (*to_bf16)(float));
bool mlp_supported(int hidden, int ffn);
hipError_t launch_mlp(hipStream_t s, const MlpParams &p);

hipError_t encoder_kernels_setup();
hipError_t launch_gemm(hipStream_t s, int epi, const GemmParams &p);

// token maps from sequence lengths: cu[b] (aligned starts), tok_seq / tok_pos for every packed row
hipError_t launch_token_map(hipStream_t s, const int32_t *lens, int B, int S, int32_t *cu, int32_t *tok_seq,
                            int32_t *tok_pos, int t_pad);

// x[t] = LayerNorm(word[id] + pos[p] + type[0])
hipError_t launch_embed_ln(hipStream_t s, const int32_t *ids, int S, const int32_t *tok_seq, const int32_t *tok_pos,
                           int t_pad, int hidden, const float *word, const float *pos, const float *type0,
                           const float *gamma, const float *beta, float eps, int vocab, bf16_t *x);

// softmax(q k^T + mask) v per (sequence, head); q is pre-scaled by 1/sqrt(d)*log2(e)
hipError_t launch_attention(hipStream_t s, const bf16_t *q, const bf16_t *k, const bf16_t *vt, int ldvt,
                            const int32_t *cu, const int32_t *lens, int B, int max_len, int heads, int d_head,
                            int hidden, bf16_t *ctx);

// masked mean (or CLS) over tokens + optional L2 normalise -> out [B, H] f32
hipError_t launch_pool(hipStream_t s, const bf16_t *x, const int32_t *cu, const int32_t *lens, int B, int hidden,
                       int pooling_cls, int normalize, float *out);

}  // namespace mx
