// encoder_precise.hip -- the kernels of the bf16x3 ("precise") sentence encoder, mx_encoder_cfg.precision = MX_PREC_BF16X3.
//
// Why it exists: the reference embeds in f32 (rust-bert `model.encode`, lib/libmemex/src/llm/embedding.rs:109) and
// north_star asks for cosine SCORES within 1e-3 of that path.  A bf16 forward holds 1 - cos <= 1e-3 row by row, but the
// cosines BETWEEN embeddings move by up to 1e-2 under checkpoint-like weights (outlier hidden dimensions, attention logits
// of +-60: a logit of 60 carries a bf16 rounding error of 0.1, i.e. 10 % on the softmax weight it decides).  A numpy
// emulation of every rounding point (scripts/encoder_rounding_sim.py, profiles/r5_encoder_rounding_sim.txt) shows that no
// single remedy helps -- an f32 residual stream, f32 GEMM results, fp16 operands, 16-bit logits alone all stay at 2e-3 ..
// 2e-2 -- and that EVERY operand of EVERY product needs ~13+ significant bits.  So this mode gives them 16:
//   * a GEMM operand a travels as [hi | lo | hi] (hi = bf16(a), lo = bf16(a - hi)), a weight w as [hi | hi | lo]: the
//     unchanged bf16 MFMA loop of gemm_kernel over 3K columns sums hi*hi + lo*hi + hi*lo in f32 (what is dropped, lo*lo,
//     is 2^-18 relative); results leave the GEMM in f32 (EPI_F32) or split again behind the exact-erf GELU (EPI_GELU_SPLIT);
//   * the hidden state (residual stream), Add & LayerNorm and pooling are f32;
//   * the attention core is f32 end to end on v_mfma_f32_32x32x2_f32 (exact products, f32 sums, running-maximum softmax).
// Measured against the f64 oracle (tests/test_encoder_gpu.py): pairwise cosine error <= 1e-4 where the bf16 path shows 1e-2.
// Cost: three times the MFMA work of the bf16 path on gemm_kernel's small tiles plus an f32 attention at 1/16 of the bf16
// MFMA rate -- a mode for deployments that need f32-grade scores, not the ingest default.
#include <cmath>

#include "encoder_kernels.h"
#include "mx_layernorm.h"

namespace mx {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

// 4 consecutive f32 -> the three blocks of a split row: o[0..3] = hi, o[blk..] = lo, o[2 blk..] = hi
__device__ __forceinline__ void store_split4(bf16_t *o, int blk, const f32x4 v) {
    bf16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (__bf16)v[e];
        lo[e] = (__bf16)(v[e] - (float)hi[e]);
    }
    *reinterpret_cast<bf16x4 *>(o) = hi;
    *reinterpret_cast<bf16x4 *>(o + blk) = lo;
    *reinterpret_cast<bf16x4 *>(o + 2 * blk) = hi;
}
// the mixed mode's two-block fp16 image: o[0..3] = fp16(v) (saturating), o[blk..] = fp16(v - hi)
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
__device__ __forceinline__ void store_split4h(bf16_t *o, int blk, const f32x4 v) {
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (_Float16)fminf(fmaxf(v[e], -65504.0f), 65504.0f);
        lo[e] = (_Float16)(v[e] - (float)hi[e]);
    }
    *reinterpret_cast<f16x4 *>(o) = hi;
    *reinterpret_cast<f16x4 *>(o + blk) = lo;
}
// MX_PREC_MIXED1: ONE fp16 value per element (saturating)
__device__ __forceinline__ void store_half4(bf16_t *o, const f32x4 v) {
    f16x4 hi;
#pragma unroll
    for (int e = 0; e < 4; ++e) hi[e] = (_Float16)fminf(fmaxf(v[e], -65504.0f), 65504.0f);
    *reinterpret_cast<f16x4 *>(o) = hi;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// embeddings + LayerNorm (embed_ln_kernel's arithmetic), f32 + split outputs.  Half a wave per packed row.
// ---------------------------------------------------------------------------------------------
template <int PER>
__global__ __launch_bounds__(256) void embed_ln_precise_kernel(const int32_t *__restrict__ ids, int S, const int32_t *__restrict__ tok_seq,
                                                                const int32_t *__restrict__ tok_pos, int t_pad,
                                                                const float *__restrict__ word, const float *__restrict__ pos,
                                                                const float *__restrict__ type0, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float eps, int vocab,
                                                                float *__restrict__ xf, bf16_t *__restrict__ xs) {
    constexpr int H = 128 * PER;
    const int l = threadIdx.x & 31;
    const int t = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (t >= t_pad) return;
    const int b = tok_seq[t];
    float *xo = xf + (size_t)t * H;
    bf16_t *so = xs + (size_t)t * 3 * H;
    if (b < 0) {  // padding row: keep it finite
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            *reinterpret_cast<f32x4 *>(xo + 4 * l + 128 * j) = z;
            store_split4(so + 4 * l + 128 * j, H, z);
        }
        return;
    }
    const int ps = tok_pos[t];
    int id = ids[(size_t)b * S + ps];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float *w = word + (size_t)id * H;
    const float *pp = pos + (size_t)ps * H;
    f32x4 v[PER];
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = 4 * l + 128 * j;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(w + c);
        const f32x4 p4 = *reinterpret_cast<const f32x4 *>(pp + c);
        const f32x4 t4 = *reinterpret_cast<const f32x4 *>(type0 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j][e] = a[e] + p4[e] + t4[e];
            sum += v[j][e];
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)H;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < PER; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) sq += (v[j][e] - mean) * (v[j][e] - mean);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)H + eps);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int c = 4 * l + 128 * j;
        const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + c);
        const f32x4 bt = *reinterpret_cast<const f32x4 *>(beta + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * g[e] + bt[e];
        *reinterpret_cast<f32x4 *>(xo + c) = o;
        store_split4(so + c, H, o);
    }
}

hipError_t launch_embed_ln_precise(hipStream_t s, const int32_t *ids, int S, const int32_t *tok_seq, const int32_t *tok_pos,
                                   int t_pad, int hidden, const float *word, const float *pos, const float *type0,
                                   const float *gamma, const float *beta, float eps, int vocab, float *xf, bf16_t *xs) {
    const dim3 g8((t_pad + 7) / 8);
    if (hidden == 384)
        hipLaunchKernelGGL(embed_ln_precise_kernel<3>, g8, dim3(256), 0, s, ids, S, tok_seq, tok_pos, t_pad, word, pos, type0, gamma, beta, eps, vocab, xf, xs);
    else if (hidden == 768)
        hipLaunchKernelGGL(embed_ln_precise_kernel<6>, g8, dim3(256), 0, s, ids, S, tok_seq, tok_pos, t_pad, word, pos, type0, gamma, beta, eps, vocab, xf, xs);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Add & LayerNorm in f32: xf[r] = LN(a[r] + xf[r]) in place, xs[r] = split3(xf[r]).  TPR lanes per row, 24 values per lane
// in 8-element chunks (hidden = 24 TPR), the statistics / affine helpers of the bf16 path (mx_layernorm.h).
// ---------------------------------------------------------------------------------------------
template <int TPR>
__global__ __launch_bounds__(256) void add_ln_split_kernel(const float *__restrict__ a, float *__restrict__ xf, bf16_t *__restrict__ xs,
                                                            int rows, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            float eps, int half2) {
    constexpr int H = 24 * TPR;
    const int l = threadIdx.x % TPR;
    const int row = (int)(blockIdx.x * (256 / TPR) + threadIdx.x / TPR);
    if (row >= rows) return;
    const float *ar = a + (size_t)row * H;
    float *xr = xf + (size_t)row * H;
    // half2 = 1: the mixed mode's [hi | lo] fp16 image, 2H wide; 2: MX_PREC_MIXED1's single fp16 value, H wide
    bf16_t *sr = xs + (size_t)row * (half2 == 2 ? 1 : half2 ? 2 : 3) * H;
    float y[24];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int col = (c * TPR + l) * 8 + 4 * hf;
            const f32x4 av = *reinterpret_cast<const f32x4 *>(ar + col);
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[c * 8 + 4 * hf + e] = av[e] + xv[e];
        }
    float mean, rstd;
    ln_row_stats<TPR, 24>(y, eps, mean, rstd);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int col = (c * TPR + l) * 8 + 4 * hf;
            const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + col);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + col);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ln_affine(y[c * 8 + 4 * hf + e], mean, rstd, g[e], b[e]);
            *reinterpret_cast<f32x4 *>(xr + col) = o;
            if (half2 == 2) store_half4(sr + col, o);
            else if (half2) store_split4h(sr + col, H, o);
            else store_split4(sr + col, H, o);
        }
}

hipError_t launch_add_ln_split(hipStream_t s, const float *a, float *xf, bf16_t *xs, int rows, int hidden, const float *gamma,
                               const float *beta, float eps, int half2) {
    if (hidden == 768)
        hipLaunchKernelGGL(add_ln_split_kernel<32>, dim3((rows + 7) / 8), dim3(256), 0, s, a, xf, xs, rows, gamma, beta, eps, half2);
    else if (hidden == 384)
        hipLaunchKernelGGL(add_ln_split_kernel<16>, dim3((rows + 15) / 16), dim3(256), 0, s, a, xf, xs, rows, gamma, beta, eps, half2);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// attention in f32.  One workgroup = 4 waves = 128 queries of one (sequence, head); a wave owns 32 queries and walks the
// keys in blocks of 32 staged in LDS (K [32][DH] and V [32][DH], f32).  Scores are computed TRANSPOSED as in the bf16 kernel
// (A = 32 keys x 2 features, B = 2 features x 32 queries per v_mfma_f32_32x32x2_f32), so a lane owns one query: the running
// maximum, the row sum and the rescale are lane-local (one exchange between the wave's halves per block), and P feeds the
// PV product without leaving its lane: register r of lane (query, h) holds key j = (r & 3) + 8 (r >> 2) + 4 h of the block,
// so k-step r of O^T += V^T P^T contracts the two keys j_0(r), j_1(r) -- the A operand of lane (dv, h) is V[j_h(r)][dv].
// Keys >= len are excluded (the reference's additive -10000 mask underflows to exactly that in f32).
// ---------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attention_f32_kernel(const float *__restrict__ qkv, int hidden, const int32_t *__restrict__ cu,
                                                             const int32_t *__restrict__ lens, int S, float qscale,
                                                             bf16_t *__restrict__ ctxs) {
    constexpr int KP = DH + 2;  // K tile pitch (words): lane (key l31, half h) reads word l31 * KP + 2 s + h -- 64 distinct banks
    constexpr int VP = DH + 8;  // V tile pitch: lane (dv l31, half h) reads word (j + 4 h) * VP + l31 -- halves 32 banks apart
    __shared__ __attribute__((aligned(16))) float Ks[32 * KP];
    __shared__ __attribute__((aligned(16))) float Vs[32 * VP];
    const int b = blockIdx.z, head = blockIdx.y, qb = blockIdx.x;
    int len = lens[b];
    len = len < 1 ? 1 : (len > S ? S : len);
    if (qb * 128 >= len) return;  // workgroup-uniform
    const int tok0 = cu[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int ld = 3 * hidden;
    const int qi = qb * 128 + wave * 32 + l31;  // this lane's query
    const bool q_ok = qi < len;
    const float *qrow = qkv + (size_t)(tok0 + (q_ok ? qi : 0)) * ld + head * DH;
    float qreg[DH / 2];
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) qreg[s] = q_ok ? qrow[2 * s + h] * qscale : 0.0f;
    f32x16 o[DH / 32];
#pragma unroll
    for (int t = 0; t < DH / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;
    const float *kbase = qkv + (size_t)tok0 * ld + hidden + head * DH;
    const float *vbase = kbase + hidden;
    const int nkb = (len + 31) / 32;
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();  // everybody is done with the previous block's tiles
        // stage K and V rows kb*32 .. +31 (zeros past the sequence's end): 32 * DH / 4 float4 per tile over 256 threads
        for (int i = tid; i < 32 * DH / 4; i += 256) {
            const int r = i / (DH / 4), c4 = i % (DH / 4);
            const int key = kb * 32 + r;
            f32x4 kv = {0.0f, 0.0f, 0.0f, 0.0f}, vv = kv;
            if (key < len) {
                kv = *reinterpret_cast<const f32x4 *>(kbase + (size_t)key * ld + 4 * c4);
                vv = *reinterpret_cast<const f32x4 *>(vbase + (size_t)key * ld + 4 * c4);
            }
            float *kd = Ks + r * KP + 4 * c4;  // (pitch DH + 2: 8-byte aligned)
            kd[0] = kv[0]; kd[1] = kv[1]; kd[2] = kv[2]; kd[3] = kv[3];
            *reinterpret_cast<f32x4 *>(Vs + r * VP + 4 * c4) = vv;
        }
        __syncthreads();
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < DH / 2; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l31 * KP + 2 * s + h], qreg[s], sc, 0, 0, 0);
        float bm = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            sc[r] = key < len ? sc[r] : -1e30f;
            bm = fmaxf(bm, sc[r]);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float m_new = fmaxf(m_run, bm);
        const float alpha = exp2f(m_run - m_new);
        m_run = m_new;
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            sc[r] = key < len ? exp2f(sc[r] - m_new) : 0.0f;
            ps += sc[r];
        }
        l_run = l_run * alpha + ps;
#pragma unroll
        for (int t = 0; t < DH / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
            for (int t = 0; t < DH / 32; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[j * VP + 32 * t + l31], sc[r], o[t], 0, 0, 0);
        }
    }
    const float l_row = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_row;
    if (!q_ok) return;
    // O^T: col = query l31, row = dv (r & 3) + 8 (r >> 2) + 4 h (+ 32 t): 4 consecutive dv per register group
    bf16_t *crow = ctxs + (size_t)(tok0 + qi) * ld + head * DH;
#pragma unroll
    for (int t = 0; t < DH / 32; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = o[t][rg * 4 + e] * inv;
            store_split4(crow + 32 * t + 8 * rg + 4 * h, hidden, v);
        }
}

// ---------------------------------------------------------------------------------------------
// The same attention on the bf16 matrix cores, every product as three (hi hi + lo hi + hi lo, the GEMMs' split): the f32 MFMA
// above has 1/16 of their rate.  Same workgroup shape, same key blocks, same lane-owns-a-query softmax; what changes:
//   * a key block is staged as bf16 PAIRS -- K_hi / K_lo [key][d] (A operand of the transposed scores: 8 consecutive d per
//     lane), V^T_hi / V^T_lo [d][slot] (A operand of O^T += V^T P^T: 8 consecutive key slots per lane) -- the same LDS bytes
//     as the f32 tiles;
//   * the scores' D layout gives lane (query, h) the keys (r & 3) + 8 (r >> 2) + 4 h; a 16-key k-step of the PV product wants
//     that lane to supply key slots 8 h .. 8 h + 7, so slot s of a k-step holds key s with bits 2 and 3 swapped (V^T is
//     staged in that order) and the lane's P registers 8 t .. 8 t + 7 ARE its B operand of k-step t, split hi / lo in place;
//   * q is split once per lane (registers), scaled first.
// Dropped: the lo x lo products (2^-18 relative), as in the GEMMs.  MEMEX_HIP_DEBUG attn_f32=1 (read when an encoder is created) keeps the f32-MFMA kernel
// (tests hold the two against each other).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8p;

// P1 (MX_PREC_MIXED1): P enters the PV product as ONE bf16 value against V's pair -- two products, no split of P in the loop.  The
// rounding simulator puts that within 1.0e-4 of the f64 evaluation on scores (profiles/r6_encoder_rounding_sim.txt, "r6b": P bf16, V
// 16-bit pair; on the bench's bge-base sample it tripled MX_PREC_MIXED's error, 6.2e-5 -> 1.8e-4, for 1-2 % of its time: that mode went
// back to three products); QK^T keeps its three products in every mode (one 16-bit value on either side costs 3e-4 ... 8e-3).
// NW waves per workgroup = 32 NW queries of one (sequence, head); key blocks of KB = 32 or 64 keys (two barriers per block).  With 16
// waves a 512-token sequence's K and V are split and staged once instead of once per 128 queries.
template <int DH, bool P1, int NW, int KB>
__global__ __launch_bounds__(64 * NW) void attention_x3_kernel(const float *__restrict__ qkv, int hidden, const int32_t *__restrict__ cu,
                                                                const int32_t *__restrict__ lens, int S, float qscale,
                                                                bf16_t *__restrict__ ctxs) {
    constexpr int NT = 64 * NW;  // threads
    constexpr int NH = KB / 32;  // halves of a key block (KB = 64, or 32 where 16 waves leave no register for 64)
    constexpr int KP = DH + 8;   // K tile pitch (bf16): rows 16 bytes apart modulo 128 -> conflict-free ds_read_b128
    constexpr int VP = KB + 8;   // V^T tile pitch (bf16)
    __shared__ __attribute__((aligned(16))) __bf16 Kh[KB * KP], Kl[KB * KP];
    __shared__ __attribute__((aligned(16))) __bf16 Vh[DH * VP], Vl[DH * VP];
    const int b = blockIdx.z, head = blockIdx.y, qb = blockIdx.x;
    int len = lens[b];
    len = len < 1 ? 1 : (len > S ? S : len);
    if (qb * 32 * NW >= len) return;  // workgroup-uniform
    const int tok0 = cu[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int ld = 3 * hidden;
    const int qi = qb * 32 * NW + wave * 32 + l31;  // this lane's query
    const bool q_ok = qi < len;
    const bool wave_ok = qb * 32 * NW + wave * 32 < len;  // a wave past the sequence's end only stages and keeps the barriers
    const float *qrow = qkv + (size_t)(tok0 + (q_ok ? qi : 0)) * ld + head * DH;
    // B operand of the scores: lane (query, h) holds q[16 s + 8 h .. + 7] of k-step s, hi and lo
    bf16x8p qh[DH / 16], ql[DH / 16];
#pragma unroll
    for (int s = 0; s < DH / 16; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = q_ok ? qrow[16 * s + 8 * h + e] * qscale : 0.0f;
            qh[s][e] = (__bf16)v;
            ql[s][e] = (__bf16)(v - (float)qh[s][e]);
        }
    f32x16 o[DH / 32];
#pragma unroll
    for (int t = 0; t < DH / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;
    const float *kbase = qkv + (size_t)tok0 * ld + hidden + head * DH;
    const int nkb = (len + KB - 1) / KB;
    // this thread's float4s of a key block's K rows (index < KB * DH / 4) and V rows (the rest); zeros past the sequence's end;
    // block kb + 1 is fetched while block kb is multiplied.  (V goes into LDS transposed, one 16-bit write per element with the
    // lanes running over the dims.  Lanes over key PAIRS and packed 32-bit writes -- no bank conflict -- was measured 18-40 % slower:
    // the global reads of a wave then touch 128 rows instead of 4.)
    constexpr int NF = KB * DH / 4;          // float4s of K (and of V) per block
    constexpr int NP = (2 * NF + NT - 1) / NT;  // float4s per thread
    f32x4 pre[NP];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + NT * u, isv = i >= NF ? 1 : 0, ii = i - isv * NF, r = ii / (DH / 4), c4 = ii % (DH / 4);
            const int key = kb * KB + r;
            pre[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (i < 2 * NF && key < len) pre[u] = *reinterpret_cast<const f32x4 *>(kbase + (size_t)key * ld + isv * hidden + 4 * c4);
        }
    };
    fetch(0);
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();  // everybody is done with the previous block's tiles
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + NT * u, isv = i >= NF ? 1 : 0, ii = i - isv * NF, r = ii / (DH / 4), c4 = ii % (DH / 4);
            if (i >= 2 * NF) continue;
            const f32x4 x = pre[u];
            bf16x4 xhi, xlo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xhi[e] = (__bf16)x[e];
                xlo[e] = (__bf16)(x[e] - (float)xhi[e]);
            }
            if (!isv) {
                *reinterpret_cast<bf16x4 *>(Kh + r * KP + 4 * c4) = xhi;
                *reinterpret_cast<bf16x4 *>(Kl + r * KP + 4 * c4) = xlo;
            } else {
                // key r of the block sits in slot (r with bits 2 and 3 swapped) of its 16-key k-step
                const int slot = (r & 48) | (r & 3) | ((r & 8) >> 1) | ((r & 4) << 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    Vh[(4 * c4 + e) * VP + slot] = xhi[e];
                    Vl[(4 * c4 + e) * VP + slot] = xlo[e];
                }
            }
        }
        if (kb + 1 < nkb) fetch(kb + 1);
        __syncthreads();
        if (!wave_ok) continue;
        f32x16 sc[NH];  // [half of the block]: lane (query, h) holds keys 32 hb + (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
        for (int hb = 0; hb < NH; ++hb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[hb][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < DH / 16; ++s) {
                const bf16x8p ah = *reinterpret_cast<const bf16x8p *>(Kh + (32 * hb + l31) * KP + 16 * s + 8 * h);
                const bf16x8p al = *reinterpret_cast<const bf16x8p *>(Kl + (32 * hb + l31) * KP + 16 * s + 8 * h);
                sc[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[s], sc[hb], 0, 0, 0);
                sc[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[s], sc[hb], 0, 0, 0);
                sc[hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[s], sc[hb], 0, 0, 0);
            }
        }
        // only the sequence's last key block can hold keys past its end: every other block skips the two selects per score
        // (the loop is bound by its VALU work per score, profiles/r6_attention_x3_geometry.txt)
        const bool ragged = (kb + 1) * KB > len;  // workgroup-uniform
        float bm = -1e30f;
        if (ragged) {
#pragma unroll
            for (int hb = 0; hb < NH; ++hb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb * KB + 32 * hb + (r & 3) + 8 * (r >> 2) + 4 * h;
                    sc[hb][r] = key < len ? sc[hb][r] : -1e30f;
                }
        }
#pragma unroll
        for (int hb = 0; hb < NH; ++hb)
#pragma unroll
            for (int r = 0; r < 16; ++r) bm = fmaxf(bm, sc[hb][r]);
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float m_new = fmaxf(m_run, bm);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // (arguments <= 0: the raw v_exp_f32 is all there is to do)
        const bool moved = m_new != m_run;
        m_run = m_new;
        float ps = 0.0f;
#pragma unroll
        for (int hb = 0; hb < NH; ++hb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // (a masked score is -1e30: exp2 of it minus a finite maximum is exactly 0, the value the select used to supply)
                sc[hb][r] = __builtin_amdgcn_exp2f(sc[hb][r] - m_new);
                ps += sc[hb][r];
            }
        l_run = l_run * alpha + ps;
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {  // after the first blocks a row's maximum rarely moves: alpha = 1 for the whole wave
#pragma unroll
            for (int t = 0; t < DH / 32; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        // k-step t2 of O^T += V^T P^T: keys 16 t2 .. 16 t2 + 15 of the block in slot order; this lane's registers 8 (t2 & 1) .. + 7 of
        // half t2 >> 1 are the keys 16 t2 + {0..3, 8..11} + 4 h = the slots 8 h .. 8 h + 7
#pragma unroll
        for (int t2 = 0; t2 < KB / 16; ++t2) {
            bf16x8p ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pv = sc[t2 >> 1][8 * (t2 & 1) + e];
                ph[e] = (__bf16)pv;
                if constexpr (!P1) pl[e] = (__bf16)(pv - (float)ph[e]);
            }
#pragma unroll
            for (int t = 0; t < DH / 32; ++t) {
                const bf16x8p vh = *reinterpret_cast<const bf16x8p *>(Vh + (32 * t + l31) * VP + 16 * t2 + 8 * h);
                const bf16x8p vl = *reinterpret_cast<const bf16x8p *>(Vl + (32 * t + l31) * VP + 16 * t2 + 8 * h);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o[t], 0, 0, 0);
                if constexpr (!P1) o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o[t], 0, 0, 0);
            }
        }
    }
    const float l_row = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_row;
    if (!q_ok) return;
    bf16_t *crow = ctxs + (size_t)(tok0 + qi) * ld + head * DH;
#pragma unroll
    for (int t = 0; t < DH / 32; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = o[t][rg * 4 + e] * inv;
            store_split4(crow + 32 * t + 8 * rg + 4 * h, hidden, v);
        }
}

// Geometry by sweep (profiles/r6_attention_x3_geometry.txt: 4 / 8 / 16 waves x 32 / 64 keys, all within 3 % of each other -- the loop is
// bound by its VALU work per score, not by staging): 16 waves at d_head 64 and 8 at d_head 32 for long sequences, 32-key blocks.
template <int DH, bool P1>
static void launch_x3(hipStream_t s, int B, int S, int heads, const float *qkv, int hidden, const int32_t *cu, const int32_t *lens,
                      float qscale, bf16_t *ctxs) {
#define MX_X3(NW) hipLaunchKernelGGL((attention_x3_kernel<DH, P1, NW, 32>), dim3((S + 32 * NW - 1) / (32 * NW), heads, B), dim3(64 * NW), 0, s, qkv, hidden, cu, lens, S, qscale, ctxs)
    if (S > 256 && DH > 32) MX_X3(16);
    else if (S > 128) MX_X3(8);
    else MX_X3(4);
#undef MX_X3
}

hipError_t launch_attention_f32(hipStream_t s, const float *qkv, const int32_t *cu, const int32_t *lens, int B, int S, int heads,
                                int d_head, int hidden, bf16_t *ctxs, bool f32_mfma, bool p_single) {
    if (B < 1 || S < 1 || S > 512 || heads * d_head != hidden) return hipErrorInvalidValue;
    const float qscale = (float)(1.4426950408889634 / sqrt((double)d_head));
    const dim3 grid((S + 127) / 128, heads, B);
    if (!f32_mfma) {
        if (d_head == 32 && p_single) launch_x3<32, true>(s, B, S, heads, qkv, hidden, cu, lens, qscale, ctxs);
        else if (d_head == 32) launch_x3<32, false>(s, B, S, heads, qkv, hidden, cu, lens, qscale, ctxs);
        else if (d_head == 64 && p_single) launch_x3<64, true>(s, B, S, heads, qkv, hidden, cu, lens, qscale, ctxs);
        else if (d_head == 64) launch_x3<64, false>(s, B, S, heads, qkv, hidden, cu, lens, qscale, ctxs);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (d_head == 32)
        hipLaunchKernelGGL(attention_f32_kernel<32>, grid, dim3(256), 0, s, qkv, hidden, cu, lens, S, qscale, ctxs);
    else if (d_head == 64)
        hipLaunchKernelGGL(attention_f32_kernel<64>, grid, dim3(256), 0, s, qkv, hidden, cu, lens, S, qscale, ctxs);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t precise_setup() { return hipSuccess; }  // static LDS only: nothing to raise

}  // namespace mx
