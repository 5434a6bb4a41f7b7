// encoder_kernels.hip -- HIP kernels of the sentence encoder (BERT forward + pooling).
//
// Replaces the libtorch CPU forward that rust-bert runs for `model.encode(&segments)`
// (reference lib/libmemex/src/llm/embedding.rs:109; arithmetic in the un-vendored crates
// rust-bert 0.21.0 / tch 0.13.0, semantics restated in oracle/bert_oracle.py):
//   K1 embed_ln_kernel     word + position + type(0) -> LayerNorm
//   K2 gemm_kernel         bf16 MFMA GEMM, f32 accumulate, fused epilogues:
//                          bias | bias+GELU(erf) | QKV split (q pre-scaled, v transposed) |
//                          bias + residual + LayerNorm                                (K4)
//   K3 attention_kernel    QK^T -> masked softmax -> PV, scores never leave registers
//   K5 pool_kernel         masked mean / CLS + L2 normalise
// All matrix math is v_mfma_f32_32x32x16_bf16; LayerNorm / softmax / GELU statistics are f32.
#include "encoder_kernels.h"

#include <cmath>

namespace mx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // native vector (HIP's uint4 struct defeats SROA)

// ---------------------------------------------------------------------------------------------
// K2: GEMM  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue)
//   workgroup = 8 waves as WM x WN; wave tile 64 x 96 = 2 x 3 MFMA 32x32 tiles (96 acc VGPRs,
//   6 MFMAs per 5 fragment reads).  Tiles are staged global -> registers -> LDS (padded rows:
//   pitch = BK*2+16 bytes keeps ds_read_b128 fragment reads conflict-free), double buffered with
//   one barrier per k-tile; the next k-tile's global loads are in flight during the MFMAs.
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int BK>
struct GemmGeom {
    static constexpr int BM = 64 * WM;
    static constexpr int BN = 96 * WN;
    static constexpr int P = BK * 2 + 16;                 // staged row pitch (bytes)
    static constexpr int STAGE = (BM + BN) * P;           // one stage: A rows then W rows
    static constexpr int PO = BN * 2 + 16;                // output tile pitch, row-major
    static constexpr int POT = BM * 2 + 16;               // output tile pitch, transposed (v^T)
    static constexpr int OUT_BYTES = (BM * PO > BN * POT) ? BM * PO : BN * POT;
    static constexpr int LDS = (2 * STAGE > OUT_BYTES) ? 2 * STAGE : OUT_BYTES;
    static constexpr int CPR = BK * 2 / 16;               // 16-B chunks per staged row
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int EPI, int WM, int WN, int BK>
__global__ __launch_bounds__(512, 2) void gemm_kernel(const GemmParams p) {
    using G = GemmGeom<WM, WN, BK>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * G::BM;
    const int n0 = blockIdx.y * G::BN;

    // kernel-argument fields used inside the staging lambdas are copied to locals first: capturing
    // the by-value struct by reference makes hipcc materialise it in scratch memory
    const bf16_t *const pa = p.a;
    const bf16_t *const pw = p.w;
    const int lda = p.lda, kdim = p.k;

    constexpr int NA = (G::BM * G::CPR + 511) / 512;
    constexpr int NW = (G::BN * G::CPR + 511) / 512;
    u32x4 ra[NA], rw[NW];

    auto load_tile = [&ra, &rw, pa, pw, lda, kdim, m0, n0, tid](int kt) __attribute__((always_inline)) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + i * 512;
            if (c < G::BM * G::CPR) {
                const int row = c / G::CPR, cc = c % G::CPR;
                ra[i] = *reinterpret_cast<const u32x4 *>(pa + (size_t)(m0 + row) * lda + k0 + cc * 8);
            }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = tid + i * 512;
            if (c < G::BN * G::CPR) {
                const int row = c / G::CPR, cc = c % G::CPR;
                rw[i] = *reinterpret_cast<const u32x4 *>(pw + (size_t)(n0 + row) * kdim + k0 + cc * 8);
            }
        }
    };
    auto store_tile = [&ra, &rw, tid](int buf) __attribute__((always_inline)) {
        char *sa = smem + buf * G::STAGE;
        char *sw = sa + G::BM * G::P;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = tid + i * 512;
            if (c < G::BM * G::CPR) *reinterpret_cast<u32x4 *>(sa + (c / G::CPR) * G::P + (c % G::CPR) * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int c = tid + i * 512;
            if (c < G::BN * G::CPR) *reinterpret_cast<u32x4 *>(sw + (c / G::CPR) * G::P + (c % G::CPR) * 16) = rw[i];
        }
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = kdim / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    const int a_off = (wm * 64 + l31) * G::P + h * 16;
    const int w_off = G::BM * G::P + (wn * 96 + l31) * G::P + h * 16;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile(kt + 1);
        const char *st = smem + (kt & 1) * G::STAGE;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            bf16x8 af[2], bf[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(st + a_off + i * 32 * G::P + ks * 32);
#pragma unroll
            for (int j = 0; j < 3; ++j) bf[j] = *reinterpret_cast<const bf16x8 *>(st + w_off + j * 32 * G::P + ks * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile((kt + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue pass 1: registers -> bf16 tile in LDS (staging buffers are free now)
    // D layout of a 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    int part = 0;
    if (EPI == EPI_QKV) part = n0 / p.hidden;  // block-uniform: BN divides hidden
    const bool transposed = (EPI == EPI_QKV) && part == 2;
    const float oscale = (EPI == EPI_QKV && part == 0) ? p.qscale : 1.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int col = wn * 96 + j * 32 + l31;
        const float b = p.bias[n0 + col];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int row0 = wm * 64 + i * 32 + 8 * rg + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[i][j][rg * 4 + e] + b;
                    if (EPI == EPI_BIAS_GELU) t = gelu_erf(t);
                    v[e] = t * oscale;
                }
                if (transposed) {
                    bf16x4 pk;
                    pk[0] = (__bf16)v[0]; pk[1] = (__bf16)v[1]; pk[2] = (__bf16)v[2]; pk[3] = (__bf16)v[3];
                    *reinterpret_cast<bf16x4 *>(smem + col * G::POT + row0 * 2) = pk;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        *reinterpret_cast<__bf16 *>(smem + (row0 + e) * G::PO + col * 2) = (__bf16)v[e];
                }
            }
        }
    }
    __syncthreads();

    // ---- epilogue pass 2: coalesced 16-byte copy-out (+ residual + LayerNorm)
    if (EPI == EPI_BIAS_RES_LN) {
        constexpr int TPR = 512 / G::BM;           // threads per row
        constexpr int CPT = G::BN / TPR / 8;       // 16-B chunks per thread
        static_assert(G::BN % (TPR * 8) == 0, "row split");
        const int row = tid / TPR, prt = tid % TPR;
        float y[CPT * 8];
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int col = (prt * CPT + c) * 8;
            const bf16x8 o = *reinterpret_cast<const bf16x8 *>(smem + row * G::PO + col * 2);
            const bf16x8 rs = *reinterpret_cast<const bf16x8 *>(p.res + (size_t)(m0 + row) * p.ldres + n0 + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                y[c * 8 + e] = (float)o[e] + (float)rs[e];
                sum += y[c * 8 + e];
            }
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)G::BN;
        float sq = 0.0f;
#pragma unroll
        for (int e = 0; e < CPT * 8; ++e) {
            const float dlt = y[e] - mean;
            sq += dlt * dlt;
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.0f / sqrtf(sq / (float)G::BN + p.eps);
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int col = (prt * CPT + c) * 8;
            const f32x4 g0 = *reinterpret_cast<const f32x4 *>(p.gamma + n0 + col);
            const f32x4 g1 = *reinterpret_cast<const f32x4 *>(p.gamma + n0 + col + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(p.beta + n0 + col);
            const f32x4 b1 = *reinterpret_cast<const f32x4 *>(p.beta + n0 + col + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (__bf16)((y[c * 8 + e] - mean) * rstd * g0[e] + b0[e]);
                o[4 + e] = (__bf16)((y[c * 8 + 4 + e] - mean) * rstd * g1[e] + b1[e]);
            }
            *reinterpret_cast<bf16x8 *>(p.out + (size_t)(m0 + row) * p.ldo + n0 + col) = o;
        }
    } else if (transposed) {
        constexpr int CPF = G::BM / 8;  // chunks per feature row
        const int nloc = n0 - 2 * p.hidden;
        for (int c = tid; c < G::BN * CPF; c += 512) {
            const int f = c / CPF, tc = c % CPF;
            const uint4 v = *reinterpret_cast<const uint4 *>(smem + f * G::POT + tc * 16);
            *reinterpret_cast<uint4 *>(p.out_vt + (size_t)(nloc + f) * p.ldvt + m0 + tc * 8) = v;
        }
    } else {
        constexpr int CPO = G::BN / 8;  // chunks per output row
        bf16_t *dst = p.out;
        int nloc = n0;
        if (EPI == EPI_QKV) {
            dst = part == 0 ? p.out : p.out_k;
            nloc = n0 - part * p.hidden;
        }
        for (int c = tid; c < G::BM * CPO; c += 512) {
            const int row = c / CPO, cc = c % CPO;
            const uint4 v = *reinterpret_cast<const uint4 *>(smem + row * G::PO + cc * 16);
            *reinterpret_cast<uint4 *>(dst + (size_t)(m0 + row) * p.ldo + nloc + cc * 8) = v;
        }
    }
}

template <int EPI, int WM, int WN, int BK>
static hipError_t gemm_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kernel<EPI, WM, WN, BK>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, GemmGeom<WM, WN, BK>::LDS);
}

template <int EPI, int WM, int WN, int BK>
static hipError_t gemm_go(hipStream_t s, const GemmParams &p) {
    using G = GemmGeom<WM, WN, BK>;
    if (p.m % G::BM || p.n % G::BN || p.k % BK) return hipErrorInvalidValue;
    dim3 grid(p.m / G::BM, p.n / G::BN);
    hipLaunchKernelGGL((gemm_kernel<EPI, WM, WN, BK>), grid, dim3(512), G::LDS, s, p);
    return hipGetLastError();
}

hipError_t launch_gemm(hipStream_t s, int epi, const GemmParams &p) {
    switch (epi) {
        case EPI_BIAS: return gemm_go<EPI_BIAS, 2, 4, 64>(s, p);
        case EPI_BIAS_GELU: return gemm_go<EPI_BIAS_GELU, 2, 4, 64>(s, p);
        case EPI_QKV: return gemm_go<EPI_QKV, 2, 4, 64>(s, p);
        case EPI_BIAS_RES_LN:
            if (p.n == 384) return gemm_go<EPI_BIAS_RES_LN, 2, 4, 64>(s, p);
            if (p.n == 768) return gemm_go<EPI_BIAS_RES_LN, 1, 8, 32>(s, p);
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// token map
// ---------------------------------------------------------------------------------------------
__global__ void token_map_kernel(const int32_t *__restrict__ lens, int B, int S, int32_t *cu, int32_t *tok_seq,
                                 int32_t *tok_pos, int t_pad) {
    // single block: exclusive scan of aligned lengths (B is small), then fill the maps
    __shared__ int s_cu[1025];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < B; ++b) {
            s_cu[b] = acc;
            int l = lens[b];
            l = l < 1 ? 1 : (l > S ? S : l);
            acc += (l + kSeqAlign - 1) / kSeqAlign * kSeqAlign;
        }
        s_cu[B] = acc;
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= B; b += blockDim.x) cu[b] = s_cu[b];
    for (int t = threadIdx.x; t < t_pad; t += blockDim.x) {
        tok_seq[t] = -1;
        tok_pos[t] = 0;
    }
    __syncthreads();
    for (int b = 0; b < B; ++b) {
        int l = lens[b];
        l = l < 1 ? 1 : (l > S ? S : l);
        for (int i = threadIdx.x; i < l; i += blockDim.x) {
            tok_seq[s_cu[b] + i] = b;
            tok_pos[s_cu[b] + i] = i;
        }
    }
}

hipError_t launch_token_map(hipStream_t s, const int32_t *lens, int B, int S, int32_t *cu, int32_t *tok_seq,
                            int32_t *tok_pos, int t_pad) {
    if (B > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(token_map_kernel, dim3(1), dim3(1024), 0, s, lens, B, S, cu, tok_seq, tok_pos, t_pad);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K1: embeddings + LayerNorm, one wave per packed row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t *__restrict__ ids, int S,
                                                       const int32_t *__restrict__ tok_seq,
                                                       const int32_t *__restrict__ tok_pos, int t_pad, int hidden,
                                                       const float *__restrict__ word, const float *__restrict__ pos,
                                                       const float *__restrict__ type0, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float eps, int vocab,
                                                       bf16_t *__restrict__ x) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= t_pad) return;
    const int b = tok_seq[t];
    bf16_t *xo = x + (size_t)t * hidden;
    if (b < 0) {  // padding row: keep it finite
        for (int c = lane; c < hidden; c += 64) xo[c] = (__bf16)0.0f;
        return;
    }
    const int ps = tok_pos[t];
    int id = ids[(size_t)b * S + ps];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float *w = word + (size_t)id * hidden;
    const float *pp = pos + (size_t)ps * hidden;
    float v[16];  // hidden <= 1024
    float sum = 0.0f;
    const int per = hidden / 64;
    for (int i = 0; i < per; ++i) {
        const int c = lane + i * 64;
        v[i] = w[c] + pp[c] + type0[c];
        sum += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)hidden;
    float sq = 0.0f;
    for (int i = 0; i < per; ++i) sq += (v[i] - mean) * (v[i] - mean);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)hidden + eps);
    for (int i = 0; i < per; ++i) {
        const int c = lane + i * 64;
        xo[c] = (__bf16)((v[i] - mean) * rstd * gamma[c] + beta[c]);
    }
}

hipError_t launch_embed_ln(hipStream_t s, const int32_t *ids, int S, const int32_t *tok_seq, const int32_t *tok_pos,
                           int t_pad, int hidden, const float *word, const float *pos, const float *type0,
                           const float *gamma, const float *beta, float eps, int vocab, bf16_t *x) {
    if (hidden % 64 || hidden > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_ln_kernel, dim3((t_pad + 3) / 4), dim3(256), 0, s, ids, S, tok_seq, tok_pos, t_pad, hidden,
                       word, pos, type0, gamma, beta, eps, vocab, x);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K3: attention.  grid = (query blocks of 128, heads, sequences); 4 waves x 32 queries.
// The whole K ([keys][d]) and V^T ([d][keys]) of the (sequence, head) sit in LDS.  Scores are
// computed TRANSPOSED (A = 32 keys, B = 32 queries) so that a lane owns one query: the running
// max / sum and the rescale factor are lane-local, P converts to the PV B-operand without any
// cross-lane movement, and O^T = V^T P^T accumulates with the query still in the lane.
// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attention_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ k,
                                                        const bf16_t *__restrict__ vt, int ldvt,
                                                        const int32_t *__restrict__ cu, const int32_t *__restrict__ lens,
                                                        int hidden, bf16_t *__restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KP = D * 2 + 16;  // K row pitch (bytes)
    const int b = blockIdx.z, hd = blockIdx.y, qb = blockIdx.x;
    const int len = lens[b];
    if (qb * 128 >= len) return;
    const int tok0 = cu[b];
    const int sb = (len + 31) / 32 * 32;  // keys rounded to MFMA blocks
    const int VP = sb * 2 + 16;           // V^T row pitch (bytes)
    char *ks = smem;
    char *vs = smem + (size_t)sb * KP;
    const int tid = threadIdx.x;

    // ---- stage K (rows >= len zero-filled) and V^T (keys >= len zero-filled)
    constexpr int KC = D * 2 / 16;
    for (int c = tid; c < sb * KC; c += 256) {
        const int row = c / KC, cc = c % KC;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < len) v = *reinterpret_cast<const uint4 *>(k + (size_t)(tok0 + row) * hidden + hd * D + cc * 8);
        *reinterpret_cast<uint4 *>(ks + row * KP + cc * 16) = v;
    }
    const int vc = sb / 8;
    for (int c = tid; c < D * vc; c += 256) {
        const int f = c / vc, kc = c % vc;
        uint4 v = *reinterpret_cast<const uint4 *>(vt + (size_t)(hd * D + f) * ldvt + tok0 + kc * 8);
        if (kc * 8 + 8 > len) {  // mask the tail so that 0 * garbage can never be NaN
            bf16x8 t = *reinterpret_cast<bf16x8 *>(&v);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (kc * 8 + e >= len) t[e] = (__bf16)0.0f;
            v = *reinterpret_cast<uint4 *>(&t);
        }
        *reinterpret_cast<uint4 *>(vs + f * VP + kc * 16) = v;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int q0 = qb * 128 + wave * 32;
    if (q0 >= len) return;
    const int qi = q0 + l31;

    // Q^T B-fragments: lane (query l31, half h) holds q[query][ks*16 + 8h .. +8]
    bf16x8 qf[D / 16];
#pragma unroll
    for (int s = 0; s < D / 16; ++s) {
        if (qi < len)
            qf[s] = *reinterpret_cast<const bf16x8 *>(q + (size_t)(tok0 + qi) * hidden + hd * D + s * 16 + h * 8);
        else
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = (__bf16)0.0f;
    }

    f32x16 o[D / 32];
#pragma unroll
    for (int t = 0; t < D / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;

    for (int kb = 0; kb < sb / 32; ++kb) {
        // S^T tile: rows = keys kb*32 + (r&3) + 8*(r>>2) + 4h, col = query l31
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < D / 16; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(ks + (kb * 32 + l31) * KP + s * 32 + h * 16);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sc, 0, 0, 0);
        }
        float bm = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            sc[r] = key < len ? sc[r] : -1e30f;  // reference: additive -10000 mask == exclusion in f32
            bm = fmaxf(bm, sc[r]);
        }
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float m_new = fmaxf(m_run, bm);
        const float alpha = exp2f(m_run - m_new);
        float ps = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = exp2f(sc[r] - m_new);
            ps += sc[r];
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < D / 32; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        // O^T += V^T P^T: k16 step s uses this lane's p[8s .. 8s+7] = keys 16s + 8(i>>2) + 4h + (i&3)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (__bf16)sc[8 * s + e];
#pragma unroll
            for (int t = 0; t < D / 32; ++t) {
                const char *vrow = vs + (t * 32 + l31) * VP + (kb * 32 + 16 * s + 4 * h) * 2;
                const bf16x4 v0 = *reinterpret_cast<const bf16x4 *>(vrow);
                const bf16x4 v1 = *reinterpret_cast<const bf16x4 *>(vrow + 16);
                bf16x8 vf;
                vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
                vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[t], 0, 0, 0);
            }
        }
    }
    l_run += __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_run;
    if (qi < len) {
        // O^T layout: col = query l31, row = dv (r&3) + 8*(r>>2) + 4h (+32t): 4 consecutive dv per group
        bf16_t *dst = ctx + (size_t)(tok0 + qi) * hidden + hd * D;
#pragma unroll
        for (int t = 0; t < D / 32; ++t)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(o[t][rg * 4 + e] * inv);
                *reinterpret_cast<bf16x4 *>(dst + t * 32 + 8 * rg + 4 * h) = pk;
            }
    }
}

static size_t attn_lds(int max_len, int d) {
    const int sb = (max_len + 31) / 32 * 32;
    return (size_t)sb * (d * 2 + 16) + (size_t)d * (sb * 2 + 16);
}

hipError_t launch_attention(hipStream_t s, const bf16_t *q, const bf16_t *k, const bf16_t *vt, int ldvt,
                            const int32_t *cu, const int32_t *lens, int B, int max_len, int heads, int d_head,
                            int hidden, bf16_t *ctx) {
    if (max_len > 512 || max_len < 1) return hipErrorInvalidValue;
    dim3 grid((max_len + 127) / 128, heads, B);
    const size_t lds = attn_lds(max_len, d_head);
    if (d_head == 32)
        hipLaunchKernelGGL((attention_kernel<32>), grid, dim3(256), lds, s, q, k, vt, ldvt, cu, lens, hidden, ctx);
    else if (d_head == 64)
        hipLaunchKernelGGL((attention_kernel<64>), grid, dim3(256), lds, s, q, k, vt, ldvt, cu, lens, hidden, ctx);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K5: pooling + L2 normalise
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_kernel(const bf16_t *__restrict__ x, const int32_t *__restrict__ cu,
                                                   const int32_t *__restrict__ lens, int hidden, int pooling_cls,
                                                   int normalize, float *__restrict__ out) {
    __shared__ float s_red[4];
    const int b = blockIdx.x;
    const int tok0 = cu[b];
    const int len = lens[b];
    const int tid = threadIdx.x;
    float v[4] = {0.f, 0.f, 0.f, 0.f};  // hidden <= 1024
    const int per = (hidden + 255) / 256;
    for (int i = 0; i < per; ++i) {
        const int c = tid + i * 256;
        if (c >= hidden) break;
        float acc = 0.0f;
        if (pooling_cls) {
            acc = (float)x[(size_t)tok0 * hidden + c];
        } else {
            for (int t = 0; t < len; ++t) acc += (float)x[(size_t)(tok0 + t) * hidden + c];
            acc = acc / fmaxf((float)len, 1e-9f);  // sum(h*m) / clamp(sum(m), 1e-9)
        }
        v[i] = acc;
    }
    float ss = 0.0f;
    for (int i = 0; i < per; ++i) ss += v[i] * v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((tid & 63) == 0) s_red[tid >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    const float sc = normalize ? 1.0f / fmaxf(nrm, 1e-12f) : 1.0f;
    for (int i = 0; i < per; ++i) {
        const int c = tid + i * 256;
        if (c < hidden) out[(size_t)b * hidden + c] = v[i] * sc;
    }
}

hipError_t launch_pool(hipStream_t s, const bf16_t *x, const int32_t *cu, const int32_t *lens, int B, int hidden,
                       int pooling_cls, int normalize, float *out) {
    if (hidden > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pool_kernel, dim3(B), dim3(256), 0, s, x, cu, lens, hidden, pooling_cls, normalize, out);
    return hipGetLastError();
}

hipError_t encoder_kernels_setup() {
    hipError_t e;
    if ((e = gemm_attr<EPI_BIAS, 2, 4, 64>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_BIAS_GELU, 2, 4, 64>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_QKV, 2, 4, 64>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_BIAS_RES_LN, 2, 4, 64>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_BIAS_RES_LN, 1, 8, 32>()) != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_kernel<32>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds(512, 32));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_kernel<64>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds(512, 64));
}

}  // namespace mx
