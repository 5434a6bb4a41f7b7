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
#include <cstdlib>
#include <type_traits>

#include "encoder_kernels.h"
#include "mx_gelu.h"
#include "mx_layernorm.h"
#include "mx_debug.h"

#include <cmath>

// Ablation switch for scripts/gemm_ubench.hip only (0 = production kernels); bits for gemm_kernel:
//   1 = no DMA inside the k loop (the ring keeps the prologue's tiles), 2 = no epilogue (accumulators kept alive),
//   4 = MFMA shape probe (16x16x32 on the same operand traffic; outputs meaningless)
#ifndef MX_GEMM_ABLATE
#define MX_GEMM_ABLATE 0
#endif

namespace mx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // native vector (HIP's uint4 struct defeats SROA)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

// one 32 x 32 x 16 MFMA on the fragments as they sit in LDS: bf16, or (the mixed mode's GEMMs) fp16 -- same bytes, same layout
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const bf16x8 &a, const bf16x8 &b, const f32x16 &c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// fp16 hi of a value (saturating: fp16 ends at 65504) and what is left of it
__device__ __forceinline__ _Float16 half_hi(float g) { return (_Float16)fminf(fmaxf(g, -65504.0f), 65504.0f); }

// ---------------------------------------------------------------------------------------------
// K2: GEMM  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogue)
//   workgroup = WM x WN waves, wave tile 64 (m) x 96 (n) = 2 x 3 MFMA 32x32x16 tiles (96 acc VGPRs,
//   6 MFMAs per 5 fragment reads).  Tiles are small enough for 2-3 workgroups per CU so that one
//   workgroup's prologue / epilogue overlaps another's MFMAs (K is only 384..3072 here).
//   Staging: global -> registers -> LDS, rows padded to BK*2+16 bytes (conflict-free
//   ds_read_b128 fragment reads), double buffered, one barrier per k-tile, next tile's global
//   loads in flight during the MFMAs.
//   Operand roles: by default the WEIGHT fragment is the MFMA A operand and the activation
//   fragment the B operand, i.e. the wave accumulates C^T tiles: a lane then owns one output row m
//   and 4 consecutive output columns n per register group -> the epilogue packs 4 bf16 into one
//   ds_write_b64 of the row-major output tile.  The V third of the QKV projection wants the
//   transposed (feature-major) output and uses the opposite roles for the same reason.
//   Epilogue: registers -> bf16 tile in LDS -> 16-byte coalesced copy-out (+ residual + LayerNorm).
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int MI, int BK, int S>
struct GemmGeom {
    static_assert(BK == 32, "staging layout below is written for 64-byte (BK = 32) rows");
    static constexpr int NT = 64 * WM * WN;               // threads
    static constexpr int NWAVE = WM * WN;
    static constexpr int BM = 32 * MI * WM;               // wave tile = (32*MI) x 96
    static constexpr int GR = BM < 128 ? BM : 128;        // rows per epilogue group (LDS output tile)
    static constexpr int NGROUP = BM / GR;
    static constexpr int BN = 96 * WN;
    static constexpr int ROWS = BM + BN;                  // staged rows per k-tile: A rows then W rows
    static constexpr int STAGE = ROWS * BK * 2;           // bytes per stage (unpadded, lane-linear DMA image)
    static constexpr int PIECES = ROWS / 16;              // 1 KiB DMA pieces (16 rows x 64 B) per stage
    static constexpr int PPW = (PIECES + NWAVE - 1) / NWAVE;  // pieces each wave issues per k-tile
    static constexpr int PO = BN * 2 + 16;                // output tile pitch, row-major [m][n]
    static constexpr int POT = GR * 2 + 16;               // output tile pitch, feature-major [n][m]
    static constexpr int OUT_BYTES = (GR * PO > BN * POT) ? GR * PO : BN * POT;
    static constexpr int LDS = (S * STAGE > OUT_BYTES) ? S * STAGE : OUT_BYTES;
    static_assert(ROWS % 16 == 0 && ((32 * MI) % GR == 0 || GR % (32 * MI) == 0), "piece / group split");
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int EPI_, int WM, int WN, int MI, int BK, int S>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmParams p) {
    // the fp16 variants (EPI_F32_H, EPI_GELU_SPLIT_H) are their bf16 namesakes with another MFMA and another split of the output
    constexpr bool F16 = EPI_ == EPI_F32_H || EPI_ == EPI_GELU_SPLIT_H;
    constexpr int EPI = EPI_ == EPI_F32_H ? (int)EPI_F32 : EPI_ == EPI_GELU_SPLIT_H ? (int)EPI_GELU_SPLIT : EPI_;
    using G = GemmGeom<WM, WN, MI, BK, S>;
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-aware order: workgroup b runs on XCD b%8 (observed dispatch rule, speed only).
    // Within an XCD consecutive workgroups walk the n-tiles of one m-tile, so the activation rows
    // they share are fetched into that XCD's L2 once.
    const int n_tiles = p.n / G::BN, m_tiles = p.m / G::BM;
    int m_idx, n_idx;
    {
        const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
        const int full = (m_tiles / 8) * 8;  // m-tiles covered by the XCD-interleaved part
        if (b < full * n_tiles) {
            m_idx = (q / n_tiles) * 8 + xcd;
            n_idx = q % n_tiles;
        } else {  // tail (m_tiles % 8 m-tiles): plain order
            const int r = b - full * n_tiles;
            m_idx = full + r / n_tiles;
            n_idx = r % n_tiles;
        }
    }
    const int m0 = m_idx * G::BM;
    const int n0 = n_idx * G::BN;
    const int kdim = p.k;
    const int nk = kdim / BK;

    // ---- staging by LDS-DMA (global_load_lds_dwordx4): stage = ROWS x 64 B, lane-linear.
    // Piece P = rows 16P..16P+15; lane l -> row 16P + (l>>2), physical 16-B chunk l&3.  Bank
    // conflicts of the fragment reads are removed by permuting the SOURCE chunk:
    // physical chunk pc holds logical chunk pc ^ ((row>>2)&3).
    const char *src[G::PPW];   // per-lane source address of each of this wave's pieces at k = 0
    size_t kstep[G::PPW];      // bytes between consecutive k-tiles of that piece's source
    uint32_t dst[G::PPW];      // wave-uniform LDS offset of the piece inside a stage
#pragma unroll
    for (int i = 0; i < G::PPW; ++i) {
        int piece = wave + i * G::NWAVE;
        piece = piece < G::PIECES ? piece : G::PIECES - 1;  // surplus issues re-load the last piece
        const int row = piece * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        // activations: row-major [M][K]; weights: K-blocked [K/32][w_rows][32] so that a weight piece
        // (16 rows x 64 B) is one contiguous KiB of full cache lines
        // (split-k launches, EPI_F32: chunk blockIdx.y starts p.k columns / p.k / 32 weight k-blocks further on)
        const size_t kz = EPI_ == EPI_F32 ? (size_t)blockIdx.y : 0;
        const bf16_t *base = row < G::BM ? p.a + (size_t)(m0 + row) * p.lda + kz * p.k
                                         : p.w + (size_t)(p.w_row0 + n0 + row - G::BM) * 32 + kz * (p.k / 32) * p.w_rows * 32;
        src[i] = reinterpret_cast<const char *>(base + c * 8);
        kstep[i] = row < G::BM ? (size_t)(BK * 2) : (size_t)p.w_rows * (BK * 2);
        dst[i] = (uint32_t)piece * 1024u;
    }
    auto issue_tile = [&](int kt) __attribute__((always_inline)) {
        const uint32_t sbase = (uint32_t)(kt % S) * G::STAGE;
#pragma unroll
        for (int i = 0; i < G::PPW; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t *)(src[i] + (size_t)kt * kstep[i]),
                                             (lds_void_t *)(smem + __builtin_amdgcn_readfirstlane(sbase + dst[i])), 16, 0, 0);
    };

    int part = 0;
    if (EPI == EPI_QKV) part = n0 / p.hidden;  // 0 = q, 1 = k (block-uniform: BN divides hidden)
    constexpr bool feature_major = (EPI == EPI_VT);  // V projection: transposed output, opposite MFMA roles

    f32x16 acc[MI][3];  // [i: 32-row m block][j: 32-col n block]
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment read offsets: row (.. + l31), logical chunk (2*ks + h) -> physical ^ ((l31>>2)&3)
    const uint32_t sw0 = (uint32_t)((h ^ ((l31 >> 2) & 3)) << 4), sw1 = sw0 ^ 32u;
    const uint32_t a_row = (uint32_t)(wm * 32 * MI + l31) * (BK * 2);
    const uint32_t w_row = (uint32_t)(G::BM + wn * 96 + l31) * (BK * 2);

#pragma unroll 1
    for (int kt = 0; kt < S - 1 && kt < nk; ++kt) issue_tile(kt);
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the (S-2) newer tiles' pieces of this wave are in flight
        if (MX_GEMM_ABLATE & 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (kt + S - 2 < nk) {
            constexpr int kInFlight = G::PPW * (S - 2);  // DMA ops of newer tiles that may stay outstanding
            if (kInFlight == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (kInFlight == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (kInFlight == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (kInFlight == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (kInFlight == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // every wave's pieces of tile kt landed; stage (kt-1)%S is free
        if (!(MX_GEMM_ABLATE & 1) && kt + S - 1 < nk) issue_tile(kt + S - 1);
        const char *st = smem + (kt % S) * G::STAGE;
#if (MX_GEMM_ABLATE & 4)  /* scripts/gemm_ubench.hip: the k-tile's operand traffic on v_mfma_f32_16x16x32_bf16 (4 per pair of 32x32x16; values meaningless) */
        {
            bf16x8 af2[2][MI], bf2[2][3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint32_t sw = ks == 0 ? sw0 : sw1;
#pragma unroll
                for (int i = 0; i < MI; ++i) af2[ks][i] = *reinterpret_cast<const bf16x8 *>(st + a_row + i * 32 * (BK * 2) + sw);
#pragma unroll
                for (int j = 0; j < 3; ++j) bf2[ks][j] = *reinterpret_cast<const bf16x8 *>(st + w_row + j * 32 * (BK * 2) + sw);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x4 t = {acc[i][j][4 * c], acc[i][j][4 * c + 1], acc[i][j][4 * c + 2], acc[i][j][4 * c + 3]};
                        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf2[c & 1][j], af2[c >> 1][i], t, 0, 0, 0);
                        acc[i][j][4 * c] = t[0], acc[i][j][4 * c + 1] = t[1], acc[i][j][4 * c + 2] = t[2], acc[i][j][4 * c + 3] = t[3];
                    }
        }
        if (false)
#endif
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const uint32_t sw = ks == 0 ? sw0 : sw1;
            bf16x8 af[MI], bf[3];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(st + a_row + i * 32 * (BK * 2) + sw);
#pragma unroll
            for (int j = 0; j < 3; ++j) bf[j] = *reinterpret_cast<const bf16x8 *>(st + w_row + j * 32 * (BK * 2) + sw);
            if (feature_major) {  // D[m][n]: lane owns column n, 4 consecutive rows m per group
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[i][j] = mfma16<F16>(af[i], bf[j], acc[i][j]);
            } else {              // D^T[n][m]: lane owns row m, 4 consecutive columns n per group
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[i][j] = mfma16<F16>(bf[j], af[i], acc[i][j]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if (MX_GEMM_ABLATE & 2) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
#endif
    if constexpr (EPI == EPI_F32) {
        // ---- the bf16x3 encoder's f32 epilogue, straight from the accumulators (a lane owns row m = its l31 and 4 consecutive
        // columns per register group: 16-byte stores; these GEMMs multiply three times the k of the bf16 path)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ncol = n0 + wn * 96 + j * 32 + 8 * rg + 4 * h;
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(p.bias + ncol);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const size_t grow = (size_t)(m0 + wm * 32 * MI + i * 32 + l31);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e] + b4[e];
                    *reinterpret_cast<f32x4 *>(p.out_f32 + (F16 ? (size_t)0 : (size_t)blockIdx.y * p.m * p.ldo) + grow * p.ldo + ncol) = v;
                }
            }
        return;
    }
    if constexpr (EPI == EPI_GELU_SPLIT) {
        // ---- bias + erf GELU (f32 accuracy, mx_gelu.h) + split into [hi | lo | hi] column blocks.  Through the LDS output tile like
        // the bf16 epilogues below, one pass per half: 8-byte stores from the accumulators' layout (a lane = a row) left the
        // 1.2 GB of this output to 150 M scattered stores -- as long again as the 3 x MFMA loop in front of them.
        static_assert(G::NGROUP == 1, "the whole tile is one output group");
        __syncthreads();  // all waves are done with the ring: its space becomes the output tile
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(p.bias + n0 + wn * 96 + j * 32 + 8 * rg + 4 * h);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const gelu_f32x2 g = gelu_erf2_precise(gelu_f32x2{acc[i][j][rg * 4 + e] + b4[e], acc[i][j][rg * 4 + e + 1] + b4[e + 1]});
                        acc[i][j][rg * 4 + e] = g[0];
                        acc[i][j][rg * 4 + e + 1] = g[1];
                    }
            }
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {  // 0: hi = bf16(g) (fp16(g) in the mixed modes), 1: lo = g - hi rounded the same way
            if (F16 && half == 1 && p.single) break;  // MX_PREC_MIXED1: one fp16 value per element (uniform over the launch)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int nloc = wn * 96 + j * 32 + 8 * rg + 4 * h;
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int mrow = wm * 32 * MI + i * 32 + l31;
                        if constexpr (F16) {
                            f16x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float g = acc[i][j][rg * 4 + e];
                                const _Float16 hi = half_hi(g);
                                pk[e] = half == 0 ? hi : (_Float16)(g - (float)hi);
                            }
                            *reinterpret_cast<f16x4 *>(smem + mrow * G::PO + nloc * 2) = pk;
                        } else {
                            bf16x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float g = acc[i][j][rg * 4 + e];
                                const __bf16 hi = (__bf16)g;
                                pk[e] = half == 0 ? hi : (__bf16)(g - (float)hi);
                            }
                            *reinterpret_cast<bf16x4 *>(smem + mrow * G::PO + nloc * 2) = pk;
                        }
                    }
                }
            __syncthreads();
            constexpr int CPO = G::BN / 8;  // 16-byte chunks per output row
            for (int c = tid; c < G::GR * CPO; c += NT) {
                const int row = c / CPO, cc = c % CPO;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(smem + row * G::PO + cc * 16);
                bf16_t *o = p.out + (size_t)(m0 + row) * p.ldo + n0 + cc * 8;
                if (half == 0) {
                    *reinterpret_cast<u32x4 *>(o) = v;
                    if (!F16) *reinterpret_cast<u32x4 *>(o + 2 * p.n) = v;  // (the mixed mode's image is [hi | lo])
                } else {
                    *reinterpret_cast<u32x4 *>(o + p.n) = v;
                }
            }
            __syncthreads();  // the tile is rewritten by the next half
        }
        return;
    }
    __syncthreads();  // all waves are done with the ring: its space becomes the output tile

    // ---- epilogue, one 128-row group at a time (the bf16 output tile of a group fits the ring's
    // LDS): pass 1 registers -> LDS tile, pass 2 coalesced 16-byte copy-out (+ residual + LayerNorm).
    // 32x32 D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const float oscale = (EPI == EPI_QKV && part == 0) ? p.qscale : 1.0f;
#pragma unroll 1
    for (int grp = 0; grp < G::NGROUP; ++grp) {
        const int g0 = grp * G::GR;                 // first tile row of this group
        const int wrow0 = wm * 32 * MI;             // first tile row of this wave
        const bool mine = wrow0 >= g0 && wrow0 < g0 + G::GR;  // wave-uniform
        if (mine) {
            if (feature_major) {
                // col = n (one bias per lane), rows = m: write [n][m .. m+3] as one 8-byte store
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int col = wn * 96 + j * 32 + l31;
                    const float b = p.bias[n0 + col];
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int row0 = wrow0 - g0 + i * 32 + 8 * rg + 4 * h;
                            bf16x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(acc[i][j][rg * 4 + e] + b);
                            *reinterpret_cast<bf16x4 *>(smem + col * G::POT + row0 * 2) = pk;
                        }
                }
            } else {
                // col = m (this lane's output row), rows = n: write [m][n .. n+3] as one 8-byte store
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int nloc = wn * 96 + j * 32 + 8 * rg + 4 * h;
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(p.bias + n0 + nloc);
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            const int mrow = wrow0 - g0 + i * 32 + l31;
                            bf16x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                gelu_f32x2 t = {acc[i][j][rg * 4 + e] + b4[e], acc[i][j][rg * 4 + e + 1] + b4[e + 1]};
                                if (EPI == EPI_BIAS_GELU) t = gelu_erf2(t);
                                pk[e] = (__bf16)(t[0] * oscale);
                                pk[e + 1] = (__bf16)(t[1] * oscale);
                            }
                            *reinterpret_cast<bf16x4 *>(smem + mrow * G::PO + nloc * 2) = pk;
                        }
                    }
            }
        }
        __syncthreads();
        const int mg = m0 + g0;  // first global row of the group
        if (EPI == EPI_BIAS_RES_LN) {
            constexpr int TPR = NT / G::GR;            // threads per row
            constexpr int CPT = G::BN / TPR / 8;       // 16-B chunks per thread
            static_assert(G::BN % (TPR * 8) == 0 && NT % G::GR == 0, "row split");
            const int row = tid / TPR, prt = tid % TPR;
            float y[CPT * 8];
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                // interleave the threads of a row chunk-wise: consecutive threads read consecutive 16 B
                const int col = (c * TPR + prt) * 8;
                const bf16x8 o = *reinterpret_cast<const bf16x8 *>(smem + row * G::PO + col * 2);
                const bf16x8 rs = *reinterpret_cast<const bf16x8 *>(p.res + (size_t)(mg + row) * p.ldres + n0 + col);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[c * 8 + e] = (float)o[e] + (float)rs[e];
            }
            float mean, rstd;
            ln_row_stats<TPR, CPT * 8>(y, p.eps, mean, rstd);
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const int col = (c * TPR + prt) * 8;
                const f32x4 g0v = *reinterpret_cast<const f32x4 *>(p.gamma + n0 + col);
                const f32x4 g1v = *reinterpret_cast<const f32x4 *>(p.gamma + n0 + col + 4);
                const f32x4 b0v = *reinterpret_cast<const f32x4 *>(p.beta + n0 + col);
                const f32x4 b1v = *reinterpret_cast<const f32x4 *>(p.beta + n0 + col + 4);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (__bf16)ln_affine(y[c * 8 + e], mean, rstd, g0v[e], b0v[e]);
                    o[4 + e] = (__bf16)ln_affine(y[c * 8 + 4 + e], mean, rstd, g1v[e], b1v[e]);
                }
                *reinterpret_cast<bf16x8 *>(p.out + (size_t)(mg + row) * p.ldo + n0 + col) = o;
            }
        } else if (feature_major) {
            constexpr int CPF = G::GR / 8;  // chunks per feature row
            const int nloc = n0;
            for (int c = tid; c < G::BN * CPF; c += NT) {
                const int f = c / CPF, tc = c % CPF;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(smem + f * G::POT + tc * 16);
                *reinterpret_cast<u32x4 *>(p.out_vt + (size_t)(nloc + f) * p.ldvt + mg + tc * 8) = v;
            }
        } else {
            constexpr int CPO = G::BN / 8;  // chunks per output row
            bf16_t *dst = p.out;
            int nloc = n0;
            if (EPI == EPI_QKV) {
                dst = part == 0 ? p.out : p.out_k;
                nloc = n0 - part * p.hidden;
            }
            for (int c = tid; c < G::GR * CPO; c += NT) {
                const int row = c / CPO, cc = c % CPO;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(smem + row * G::PO + cc * 16);
                *reinterpret_cast<u32x4 *>(dst + (size_t)(mg + row) * p.ldo + nloc + cc * 8) = v;
            }
        }
        if (grp + 1 < G::NGROUP) __syncthreads();  // the next group reuses the LDS tile
    }
}

template <int EPI, int WM, int WN, int MI, int BK, int S>
static hipError_t gemm_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_kernel<EPI, WM, WN, MI, BK, S>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, GemmGeom<WM, WN, MI, BK, S>::LDS);
}

template <int EPI, int WM, int WN, int MI, int BK, int S>
static hipError_t gemm_go(hipStream_t s, const GemmParams &p) {
    using G = GemmGeom<WM, WN, MI, BK, S>;
    if (p.m % G::BM || p.n % G::BN || p.k % BK) return hipErrorInvalidValue;
    if (EPI == EPI_QKV && (p.hidden % G::BN || p.n != 2 * p.hidden)) return hipErrorInvalidValue;
    dim3 grid((p.m / G::BM) * (p.n / G::BN), EPI == EPI_F32 && p.ksplit > 1 ? p.ksplit : 1);
    hipLaunchKernelGGL((gemm_kernel<EPI, WM, WN, MI, BK, S>), grid, dim3(G::NT), G::LDS, s, p);
    return hipGetLastError();
}

// tile configurations (measured): the epilogues (GELU, LDS staging, HBM writes)
// take as long as the K = 384 MFMA loops, so the plain GEMMs use 128 x 192 tiles with 4 waves and an
// 80 KiB 4-stage ring -> TWO workgroups per CU, one's epilogue overlapping the other's MFMAs.  The
// LayerNorm GEMMs need a full row per tile: 128 x 384 (8 waves, 128 KiB ring) and 64 x 768
// (3-stage 52 KiB ring); 64 x 384 at 2 workgroups/CU was measured slower (weight re-reads double).
// 256 x 384 tiles (8 waves, wave tile 128 x 96: 12 MFMAs per 7 fragment reads instead of 6 per 5, a 3-stage
// 120 KiB ring, one workgroup per CU) for large passes.  Round 3: alone on the chip with synthetic operands
// (scripts/gemm_ubench.hip, 131k tokens) they run the QK projection in 116 us instead of 133 at K = 384, in 337
// instead of 387 at K = 768, and the FFN-up GEMM in 742 instead of 807 -- bit-identical outputs, every
// configuration sums k in the same order.  INSIDE the encoder (scripts/r3_enc_ab.sh: same box, same process,
// real operands, the package at its power cap) the gain is +1.2 % chunks/s for the hidden-768 models with the QK
// projection alone on big tiles, -1.3 % at hidden 384, and the V projection loses 30 us per layer (its
// feature-major epilogue): so the QK projection alone takes them, for K >= 768 only (the A/B mask of that measurement --
// bit 0 QK, bit 1 V, bit 2 bias / GELU GEMMs -- is gone from the library; scripts/gemm_ubench.hip still runs every tile).
constexpr int kBigTileRows = 32768;

hipError_t launch_gemm(hipStream_t s, int epi, const GemmParams &p) {
    // mask: bit 0 = QK projection, bit 1 = V projection, bit 2 = bias / GELU GEMMs on the 256 x 384 tiles
    const bool fits = p.m >= kBigTileRows && p.n % 384 == 0;
    const int mask = p.k >= 768 ? 1 : 0;
    const bool big_qk = fits && (mask & 1), big_vt = fits && (mask & 2), big_ff = fits && (mask & 4);
    switch (epi) {
        case EPI_BIAS: return big_ff ? gemm_go<EPI_BIAS, 2, 4, 4, 32, 3>(s, p) : gemm_go<EPI_BIAS, 2, 2, 2, 32, 4>(s, p);
        case EPI_BIAS_GELU: return big_ff ? gemm_go<EPI_BIAS_GELU, 2, 4, 4, 32, 3>(s, p) : gemm_go<EPI_BIAS_GELU, 2, 2, 2, 32, 4>(s, p);
        case EPI_QKV: return big_qk ? gemm_go<EPI_QKV, 2, 4, 4, 32, 3>(s, p) : gemm_go<EPI_QKV, 2, 2, 2, 32, 4>(s, p);
        case EPI_VT: return big_vt ? gemm_go<EPI_VT, 2, 4, 4, 32, 3>(s, p) : gemm_go<EPI_VT, 2, 2, 2, 32, 4>(s, p);
        case EPI_F32: return (fits && (mask & 8)) ? gemm_go<EPI_F32, 2, 4, 4, 32, 3>(s, p) : gemm_go<EPI_F32, 2, 2, 2, 32, 4>(s, p);
        case EPI_GELU_SPLIT: return gemm_go<EPI_GELU_SPLIT, 2, 2, 2, 32, 4>(s, p);
        case EPI_F32_H: return gemm_go<EPI_F32_H, 2, 2, 2, 32, 4>(s, p);
        case EPI_GELU_SPLIT_H: return gemm_go<EPI_GELU_SPLIT_H, 2, 2, 2, 32, 4>(s, p);
        case EPI_BIAS_RES_LN:
            if (p.n == 384) return gemm_go<EPI_BIAS_RES_LN, 2, 4, 2, 32, 4>(s, p);
            if (p.n == 768) return gemm_go<EPI_BIAS_RES_LN, 1, 8, 2, 32, 3>(s, p);
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------
// token map
// ---------------------------------------------------------------------------------------------
// The attention work list of a pass (attention_kernel below): one ITEM per (sequence, head group), sequences in order of
// decreasing length (all groups of a sequence adjacent), written by block 0 of token_map_kernel.  Workgroup j of the
// persistent attention grid takes the items j, j + G, j + 2G, ... -- one from every tier of the sorted list, so the loads
// differ by about one short item.
struct AttnItem {
    int32_t tok0, len, group, pad;
};

__global__ __launch_bounds__(1024) void token_map_kernel(const int32_t *__restrict__ lens, int B, int S, int32_t *cu,
                                                         int32_t *tok_seq, int32_t *tok_pos, int t_pad, int groups,
                                                         AttnItem *__restrict__ plan) {
    // blocks of 1024 threads (B <= 1024): every block scans the aligned lengths itself (1024 elements: cheaper
    // than a second launch), then the packed rows, dealt over the grid, find their sequence by binary
    // search in the scanned starts
    __shared__ int s_cu[1025];
    __shared__ int s_len[1024];
    const int tid = threadIdx.x;
    int l = 0, al = 0;
    if (tid < B) {
        l = lens[tid];
        l = l < 1 ? 1 : (l > S ? S : l);
        al = (l + kSeqAlign - 1) / kSeqAlign * kSeqAlign;
    }
    s_len[tid] = l;
    s_cu[tid + 1] = al;
    if (tid == 0) s_cu[0] = 0;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan of s_cu[1..1024]
        const int v = tid + 1 > off ? s_cu[tid + 1 - off] : 0;
        __syncthreads();
        s_cu[tid + 1] += v;
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid <= B) cu[tid] = s_cu[tid];
    if (plan && B <= 128) {  // few sequences: one thread per sequence counts for itself (no barriers: query-sized passes)
        if (blockIdx.x == gridDim.x - 1 && tid < B) {
            int rank = 0;
            for (int j = 0; j < B; ++j) {
                const int o = s_len[j];
                rank += (o > l || (o == l && j < tid)) ? 1 : 0;
            }
            for (int g = 0; g < groups; ++g) plan[(size_t)rank * groups + g] = AttnItem{s_cu[tid], l, g, 0};
        }
    } else if (plan) {
        // rank of sequence b = how many come first: longer ones, and equally long ones with a smaller index.  Block j ranks the
        // sequences j, j + gridDim.x, ...: every thread compares ITS sequence with b, the block counts (one thread per sequence
        // looping over all the others in the last block alone took 48 us per pass of 1024 sequences)
        __shared__ int s_rank;
        for (int b = blockIdx.x; b < B; b += gridDim.x) {
            const int lb = s_len[b];
            if (tid == 0) s_rank = 0;
            __syncthreads();
            const bool first = tid < B && (l > lb || (l == lb && tid < b));
            const unsigned long long m = __ballot(first);
            if ((tid & 63) == 0 && m) atomicAdd(&s_rank, __popcll(m));
            __syncthreads();
            if (tid < groups) plan[(size_t)s_rank * groups + tid] = AttnItem{s_cu[b], lb, tid, 0};
            __syncthreads();
        }
    }
    const int total = s_cu[B];
    for (int t = blockIdx.x * 1024 + tid; t < t_pad; t += gridDim.x * 1024) {
        int seq = -1, pos = 0;
        if (t < total) {
            int lo = 0, hi = B - 1;  // last b with s_cu[b] <= t
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (s_cu[mid] <= t) lo = mid;
                else hi = mid - 1;
            }
            const int p_ = t - s_cu[lo];
            if (p_ < s_len[lo]) {  // rows between a sequence's end and the next aligned start stay unmapped
                seq = lo;
                pos = p_;
            }
        }
        tok_seq[t] = seq;
        tok_pos[t] = pos;
    }
}

hipError_t launch_token_map(hipStream_t s, const int32_t *lens, int B, int S, int32_t *cu, int32_t *tok_seq,
                            int32_t *tok_pos, int t_pad, int heads, int d_head, int max_len, void *attn_plan) {
    if (B > 1024) return hipErrorInvalidValue;
    const int blocks = t_pad / 1024 < 1 ? 1 : (t_pad / 1024 > 256 ? 256 : t_pad / 1024);
    hipLaunchKernelGGL(token_map_kernel, dim3(blocks), dim3(1024), 0, s, lens, B, S, cu, tok_seq, tok_pos, t_pad,
                       attention_groups(heads, d_head, max_len, B), reinterpret_cast<AttnItem *>(attn_plan));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K1: embeddings + LayerNorm, half a wave per packed row
// ---------------------------------------------------------------------------------------------
// hidden = 128 * PER (384, 768: what mx_encoder_create accepts).  HALF a wave per token, 16-byte loads of the
// three table rows (lane l of the half owns columns 4l + 128j .. +3), 8-byte stores; 8 tokens per workgroup:
// 58 us per 131k tokens = 5.2 TB/s (the one-wave-per-token form with 4-byte loads it replaces: 123 us).
template <int PER>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t *__restrict__ ids, int S,
                                                        const int32_t *__restrict__ tok_seq,
                                                        const int32_t *__restrict__ tok_pos, int t_pad,
                                                        const float *__restrict__ word, const float *__restrict__ pos,
                                                        const float *__restrict__ type0, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float eps, int vocab,
                                                        bf16_t *__restrict__ x) {
    constexpr int H = 128 * PER;
    const int l = threadIdx.x & 31;
    const int t = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (t >= t_pad) return;
    const int b = tok_seq[t];
    bf16_t *xo = x + (size_t)t * H;
    if (b < 0) {  // padding row: keep it finite
        const bf16x4 z = {(__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f, (__bf16)0.0f};
#pragma unroll
        for (int j = 0; j < PER; ++j) *reinterpret_cast<bf16x4 *>(xo + 4 * l + 128 * j) = z;
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
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)((v[j][e] - mean) * rstd * g[e] + bt[e]);
        *reinterpret_cast<bf16x4 *>(xo + c) = o;
    }
}

hipError_t launch_embed_ln(hipStream_t s, const int32_t *ids, int S, const int32_t *tok_seq, const int32_t *tok_pos,
                           int t_pad, int hidden, const float *word, const float *pos, const float *type0,
                           const float *gamma, const float *beta, float eps, int vocab, bf16_t *x) {
    const dim3 g8((t_pad + 7) / 8);
    if (hidden == 384)
        hipLaunchKernelGGL(embed_ln_kernel<3>, g8, dim3(256), 0, s, ids, S, tok_seq, tok_pos, t_pad, word, pos, type0, gamma, beta, eps, vocab, x);
    else if (hidden == 768)
        hipLaunchKernelGGL(embed_ln_kernel<6>, g8, dim3(256), 0, s, ids, S, tok_seq, tok_pos, t_pad, word, pos, type0, gamma, beta, eps, vocab, x);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K4b: row LayerNorm in place (large passes of hidden-768 models: pgemm_kernel's EPI_BIAS_RES leaves
// y = attention / MLP output + bias + residual, this turns it into LayerNorm(y)).  TPR lanes per row, 3 x 16 bytes per
// lane (hidden = 24 TPR), same statistics / affine helpers as the fused epilogues.  HBM-bound: 2 x 201 MB per call at
// 131k tokens x 768.
// ---------------------------------------------------------------------------------------------
template <int TPR>
__global__ __launch_bounds__(256) void ln_rows_kernel(bf16_t *__restrict__ x, int ld, int rows, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float eps, int rev) {
    const int l = threadIdx.x % TPR;
    // rev: rows from the END of the matrix -- the GEMM that reads the result starts where this launch wrote last (worth 0.4 %
    // of a hidden-768 layer: W1 664 -> 657 us, QK 283.6 -> 281; the launch itself 67 us either way)
    const int row = rev ? rows - 1 - (int)(blockIdx.x * (256 / TPR) + threadIdx.x / TPR) : (int)(blockIdx.x * (256 / TPR) + threadIdx.x / TPR);
    if (row >= rows || row < 0) return;
    bf16_t *xr = x + (size_t)row * ld;
    float y[24];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(xr + (c * TPR + l) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[c * 8 + e] = (float)v[e];
    }
    float mean, rstd;
    ln_row_stats<TPR, 24>(y, eps, mean, rstd);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int col = (c * TPR + l) * 8;
        const f32x4 g0v = *reinterpret_cast<const f32x4 *>(gamma + col);
        const f32x4 g1v = *reinterpret_cast<const f32x4 *>(gamma + col + 4);
        const f32x4 b0v = *reinterpret_cast<const f32x4 *>(beta + col);
        const f32x4 b1v = *reinterpret_cast<const f32x4 *>(beta + col + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (__bf16)ln_affine(y[c * 8 + e], mean, rstd, g0v[e], b0v[e]);
            o[4 + e] = (__bf16)ln_affine(y[c * 8 + 4 + e], mean, rstd, g1v[e], b1v[e]);
        }
        *reinterpret_cast<bf16x8 *>(xr + col) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// closes a split-k GEMM: out[r] = LayerNorm(bf16(sum_z part[z][r] + bias) + res[r]).  TPR lanes per row, 24 columns per lane (hidden
// 384: 16 lanes, 768: 32); all chunks' loads in flight together, summed in chunk order; the rounding points of the fused
// Add & LayerNorm epilogue (bf16 of product + bias, then + residual in f32).
// ---------------------------------------------------------------------------------------------
template <int TPR>
__global__ __launch_bounds__(256) void reduce_res_ln_kernel(const float *__restrict__ part, int nsplit, int m, const float *__restrict__ bias,
                                                             const bf16_t *__restrict__ res, int ldres, const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float eps, bf16_t *__restrict__ out, int ldo) {
    constexpr int N = TPR * 24, kMaxSplit = 8;
    const int l = threadIdx.x % TPR;
    const int row = (int)(blockIdx.x * (256 / TPR) + threadIdx.x / TPR);
    if (row >= m) return;
    float y[24];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        const int col = (cc * TPR + l) * 8;
        f32x4 pa[kMaxSplit], pb[kMaxSplit];
#pragma unroll
        for (int z = 0; z < kMaxSplit; ++z) {
            const float *pp = part + ((size_t)(z < nsplit ? z : 0) * m + row) * N + col;
            pa[z] = *reinterpret_cast<const f32x4 *>(pp);
            pb[z] = *reinterpret_cast<const f32x4 *>(pp + 4);
        }
        f32x4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
#pragma unroll
        for (int z = 0; z < kMaxSplit; ++z)
            if (z < nsplit) {
                s0 += pa[z];
                s1 += pb[z];
            }
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bias + col);
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bias + col + 4);
        const bf16x8 rs = *reinterpret_cast<const bf16x8 *>(res + (size_t)row * ldres + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y[cc * 8 + e] = (float)(__bf16)(s0[e] + b0[e]) + (float)rs[e];
            y[cc * 8 + 4 + e] = (float)(__bf16)(s1[e] + b1[e]) + (float)rs[4 + e];
        }
    }
    float mean, rstd;
    ln_row_stats<TPR, 24>(y, eps, mean, rstd);
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        const int col = (cc * TPR + l) * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4 *>(gamma + col);
        const f32x4 g1 = *reinterpret_cast<const f32x4 *>(gamma + col + 4);
        const f32x4 e0 = *reinterpret_cast<const f32x4 *>(beta + col);
        const f32x4 e1 = *reinterpret_cast<const f32x4 *>(beta + col + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = (__bf16)ln_affine(y[cc * 8 + e], mean, rstd, g0[e], e0[e]);
            o[4 + e] = (__bf16)ln_affine(y[cc * 8 + 4 + e], mean, rstd, g1[e], e1[e]);
        }
        *reinterpret_cast<bf16x8 *>(out + (size_t)row * ldo + col) = o;
    }
}

hipError_t launch_reduce_res_ln(hipStream_t s, const float *part, int nsplit, int m, int n, const float *bias, const bf16_t *res, int ldres,
                                const float *gamma, const float *beta, float eps, bf16_t *out, int ldo) {
    if (nsplit < 1 || nsplit > 8 || m < 1) return hipErrorInvalidValue;
    if (n == 768)
        hipLaunchKernelGGL(reduce_res_ln_kernel<32>, dim3((m + 7) / 8), dim3(256), 0, s, part, nsplit, m, bias, res, ldres, gamma, beta, eps, out, ldo);
    else if (n == 384)
        hipLaunchKernelGGL(reduce_res_ln_kernel<16>, dim3((m + 15) / 16), dim3(256), 0, s, part, nsplit, m, bias, res, ldres, gamma, beta, eps, out, ldo);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// closes a split-k QKV product (small passes of the hidden-768 models): one thread per 8 columns of a row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_qkv_kernel(const float *__restrict__ part, int nsplit, int m, int hidden, const float *__restrict__ bias,
                                                          float qscale, bf16_t *__restrict__ q, bf16_t *__restrict__ k, bf16_t *__restrict__ vt,
                                                          int ldvt) {
    const int cpr = 3 * hidden / 8;  // 8-column chunks per row
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= m * cpr) return;
    // consecutive threads take consecutive ROWS of one chunk: the v^T stores of a wave are then 64 consecutive tokens of a feature
    const int row = i % m, col = (i / m) * 8;
    f32x4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
    for (int z = 0; z < nsplit; ++z) {
        const float *pp = part + ((size_t)z * m + row) * (3 * hidden) + col;
        s0 += *reinterpret_cast<const f32x4 *>(pp);
        s1 += *reinterpret_cast<const f32x4 *>(pp + 4);
    }
    const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bias + col), b1 = *reinterpret_cast<const f32x4 *>(bias + col + 4);
    const int part_id = col / hidden;  // 0 = q, 1 = k, 2 = v (8 divides hidden)
    const float sc = part_id == 0 ? qscale : 1.0f;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o[e] = (__bf16)((s0[e] + b0[e]) * sc);
        o[4 + e] = (__bf16)((s1[e] + b1[e]) * sc);
    }
    if (part_id < 2) {
        *reinterpret_cast<bf16x8 *>((part_id == 0 ? q : k) + (size_t)row * hidden + (col - part_id * hidden)) = o;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) vt[(size_t)(col - 2 * hidden + e) * ldvt + row] = o[e];
    }
}

hipError_t launch_reduce_qkv(hipStream_t s, const float *part, int nsplit, int m, int hidden, const float *bias, float qscale, bf16_t *q,
                             bf16_t *k, bf16_t *vt, int ldvt) {
    if (nsplit < 1 || m < 1 || hidden % 8) return hipErrorInvalidValue;
    const long n = (long)m * (3 * hidden / 8);
    hipLaunchKernelGGL(reduce_qkv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, nsplit, m, hidden, bias, qscale, q, k, vt, ldvt);
    return hipGetLastError();
}

hipError_t launch_ln_rows(hipStream_t s, bf16_t *x, int ld, int rows, int hidden, const float *gamma, const float *beta, float eps) {
    const int rev = 1;  // last rows first (profiles/r4_ln_rows_order.txt)
    if (hidden == 768)
        hipLaunchKernelGGL(ln_rows_kernel<32>, dim3((rows + 7) / 8), dim3(256), 0, s, x, ld, rows, gamma, beta, eps, rev);
    else if (hidden == 384)
        hipLaunchKernelGGL(ln_rows_kernel<16>, dim3((rows + 15) / 16), dim3(256), 0, s, x, ld, rows, gamma, beta, eps, rev);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K3: attention.  A persistent grid of one workgroup per CU, 16 waves x 32 queries (sequences of <= 512 tokens); a workgroup
// walks its share of the pass's (sequence, head group) items, longest sequences first (the plan token_map_kernel writes).
// K ([keys][d]) and V^T ([d][keys]) stream through LDS in STAGES of 256 keys, double-buffered, moved by LDS-DMA (global -> LDS, 1 KiB per wave-instruction, no registers): the DMA of stage
// j+1 -- the other half of the keys, or the next item -- is issued (a piece per key block) behind the barrier that ends stage
// j-1 and lands under the key loop of stage j; a counted s_waitcnt + ONE barrier per stage.  (Round 4: the form this replaces
// loaded a head into registers, waited, wrote LDS, ran the loop; with the loop skipped a launch still took 209 of 341
// us at d = 64 and 113 of 161 at d = 32 -- 805 / 403 MB at ~4 TB/s, none of it overlapped with the loop.  A register
// prefetch of the next head under the loop needs 32 more VGPRs than the 128 a 16-wave workgroup has at d = 64.)
// Scores are computed TRANSPOSED (A = 32 keys, B = 32 queries) so that a lane owns one query: the running max / sum and
// the rescale factor are lane-local, P converts to the PV B-operand without any cross-lane movement, and O^T = V^T P^T
// accumulates with the query still in the lane.  At d = 32 the kernel is softmax(VALU)-bound by construction (128 MFMA
// flops per score), so the VALU work per score is kept minimal: raw v_exp_f32, masking only in the tail key block, the
// O/l rescale only when some lane's running max actually grew.
// Key order: the lane that supplies A-row j of a score tile reads key row pi(j) (bits 2 and 3 of j swapped), so the 8
// scores a lane holds for one k16 step of PV are 8 CONSECUTIVE keys and a PV A-fragment is ONE ds_read_b128 of the
// V^T tile in natural key order (which is what a DMA can deliver).
// LDS tiles are unpadded; the DMA's source side applies the swizzles that make the fragment reads conflict-free:
//   K   row r (2D bytes):  16-byte chunk c at c ^ ((r / RW) & (CH - 1)), CH = D/8 chunks per row, RW = 128/D rows per 256 B
//   V^T row f (512 bytes): 16-byte chunk c at c ^ (f & 15)
// Rows >= len of q / k read as zero and the matching ctx stores are dropped by the buffer bounds check; V^T columns >=
// len hold other tokens' (finite or not) values: the tail key block masks its V fragments along with its scores.
// ---------------------------------------------------------------------------------------------
constexpr int kAttnWaves = 16;
constexpr int kAttnQ = kAttnWaves * 32;  // queries per workgroup
constexpr int kAttnStage = 256;          // keys per stage

template <int DH, int HP>
__global__ __launch_bounds__(kAttnWaves * 64, 4) void attention_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ k,
                                                                       const bf16_t *__restrict__ vt, int ldvt,
                                                                       const AttnItem *__restrict__ plan, int n_items, int hidden,
                                                                       bf16_t *__restrict__ ctx, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = DH * HP;                  // features per stage: HP adjacent heads of DH (a "head group": rows of 2D bytes)
    constexpr int RB = D * 2;                   // bytes of a K row
    constexpr int CH = RB / 16;                 // 16-byte chunks per K row
    constexpr int RW = 256 / RB;                // K rows per 256 bytes
    constexpr int KBYTES = kAttnStage * RB;     // K tile of a stage; the V^T tile (D rows x 512 B) is as large
    constexpr int SB = 2 * KBYTES;              // one stage buffer
    constexpr int KI = D / 32;                  // DMA instructions per wave and tile
    constexpr int NS = D / 32 * 2;              // ctx stores per item and lane
    const int G = gridDim.x;
    const int n_my = (n_items - (int)blockIdx.x + G - 1) / G;  // this workgroup's items: ordinals 0 .. n_my - 1 (<= 64)
    if (n_my <= 0) return;
    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int pitch = hidden * 2;
    // MEMEX_HIP_DEBUG attn_safe=2 .. 6 (measurement only): 2, 3, 4 no key loop; 3: no q loads and no ctx stores either; 4: no ctx
    // stores; 5: key loop only (no DMA traffic, no q loads, no ctx stores); 6: everything but the ctx stores
    const bool no_keys = mode >= 2 && mode <= 4;
    const bool q_live = mode != 3 && mode != 5, c_live = mode < 3, kv_live = mode != 5;
    uint64_t *wg_redo = reinterpret_cast<uint64_t *>(smem + 2 * SB);
    if (tid == 0) *wg_redo = 0ull;

    struct Item {  // wave-uniform
        int tok0, len, grp, sb, nh;
        bool live;
    };
    auto fetch = [&](int ord) __attribute__((always_inline)) {  // ordinal -> item (one 16-byte scalar load)
        Item it;
        it.live = ord >= 0 && ord < n_my;
        const AttnItem r = plan[it.live ? (int)blockIdx.x + ord * G : (int)blockIdx.x];
        it.tok0 = r.tok0;
        it.len = r.len;
        it.grp = r.group;
        it.sb = (r.len + 31) / 32 * 32;                          // keys rounded to MFMA blocks
        it.nh = (it.sb + kAttnStage - 1) / kAttnStage;           // stages of the item (1 or 2)
        return it;
    };

    // ---- one stage: K rows / V^T columns [256 half, 256 half + 256) of item `it` -> buffer buf, in NP pieces of one DMA
    // instruction per wave (pieces < KI: K, the others: V^T; rows >= len and key columns >= sb get an out-of-range offset:
    // zeros, no traffic; a dead item: descriptors of 0 bytes -- the operation counts in vmcnt and touches no memory).
    // A wave issues ONE piece per key block: 16 waves x 4-8 vector-memory instructions in a burst behind the stage's
    // barrier hold every wave at the issue until the address unit has taken them (~16-32 clocks each).
    constexpr int NP = 2 * KI;
    auto issue_piece = [&](const Item &it, int half, int buf, const int pc) __attribute__((always_inline)) {
        asm volatile("" : "+v"(lane));  // per-lane offsets are derived here: ~5 VALU, no registers held
        const uint32_t dst = (uint32_t)(buf * SB + wave * KI * 1024);
        const bool live = it.live && kv_live;
        if (pc < KI) {
            const int i = pc;
            const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void *)(k + (size_t)it.tok0 * hidden + it.grp * D), 0,
                                                                                   live ? (uint32_t)((it.len - 1) * pitch + RB) : 0u, 0x00020000);
            const int row = (wave * KI + i) * (1024 / RB) + lane / CH;
            const int lc = (lane % CH) ^ ((row / RW) & (CH - 1));
            const uint32_t vo = (uint32_t)((half * kAttnStage + row) * pitch + lc * 16);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (lds_void_t *)(smem + dst + i * 1024), 16, vo, 0, 0, 0);
        } else {
            const int i = pc - KI;
            const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(vt + (size_t)it.grp * D * ldvt + it.tok0), 0, live ? (uint32_t)(((size_t)D * ldvt - it.tok0) * 2) : 0u, 0x00020000);
            const int f = (wave * KI + i) * 2 + (lane >> 5);
            const int lc = (lane & 31) ^ (f & 15);
            const int key = half * kAttnStage + lc * 8;
            const uint32_t vo = key < it.sb ? (uint32_t)(f * ldvt * 2 + key * 2) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (lds_void_t *)(smem + dst + KBYTES + i * 1024), 16, vo, 0, 0, 0);
        }
    };
    // Q^T B-fragments: lane (query l31, half h) holds q[query][s*16 + 8h .. +8] (zero for queries >= len; dead item: zeros)
    auto load_q1 = [&](const Item &it, const int s) __attribute__((always_inline)) {
        asm volatile("" : "+v"(lane));
        const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(q + (size_t)it.tok0 * hidden + it.grp * D), 0, it.live && q_live ? (uint32_t)((it.len - 1) * pitch + RB) : 0u, 0x00020000);
        const int vo = (wave * 32 + (lane & 31)) * pitch + (lane >> 5) * 16;
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_q, vo + s * 32, 0, 0));
    };

    // fragment read offsets inside a stage buffer (see the swizzles above); k16 step s and key block kb are XORed / added in
    const int pr = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);  // pi(l31)
    const uint32_t k_frag = (uint32_t)(pr * RB + ((h ^ ((pr / RW) & (CH - 1))) << 4));
    const uint32_t v_frag = (uint32_t)(KBYTES + l31 * 512 + ((h ^ (l31 & 15)) << 4));

    uint64_t my_redo = 0ull;  // items (by ordinal) whose row sums left (1e-30, 1e30) in this wave: redone with the running maximum
    // ---- one pass over the items in `todo` (bit i = ordinal i); a wave computes the items in `mine`.  The stage pipeline
    // runs across items: the DMA of an item's first stage is issued under the previous item's last one.
    // SAFE: the textbook running maximum (max + cross-half exchange + compare per 32-key block, rescale when it grows).
    // !SAFE: NO shift at all, P = exp2(score) -- softmax is shift-invariant and P, l, O are floating point, so scores away
    // from 0 only move the exponents; what can go wrong is exp2 overflowing (a score above 127; the pre-scaled logits of
    // the models this runs stay within a few tens) or a whole row underflowing, and both leave the row sum outside
    // (1e-30, 1e30), which puts the item on the wave's redo list.  The max chain and the subtraction are 21 of ~105 VALU
    // issue slots per block, and the max sits on the MFMA -> exp dependency chain.
    auto pass = [&](auto safe_tag, const uint64_t todo, const uint64_t mine) __attribute__((always_inline)) {
        constexpr bool SAFE = decltype(safe_tag)::value;
        if (todo == 0ull) return;
        auto first_of = [](uint64_t m) { return m ? __builtin_ctzll(m) : -1; };
        uint64_t rest = todo;  // ordinals not yet fetched
        auto pop = [&]() __attribute__((always_inline)) {
            const int o = first_of(rest);
            rest &= rest - 1ull;
            return o;
        };
        int c_ord = pop(), x_ord = pop(), nn_ord = pop();
        Item cur = fetch(c_ord), nxt = fetch(x_ord), nn = fetch(nn_ord);  // nn: fetched an item early, becomes nxt at the next boundary
        int chalf = 0, cbuf = 0;
        bf16x8 qf[D / 16], qn[D / 16];
        f32x16 o[D / 32];
        float m_run[HP], l_run[HP];
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) issue_piece(cur, 0, 0, pc);
#pragma unroll
        for (int s = 0; s < D / 16; ++s) qf[s] = load_q1(cur, s);
        // Waits for a stage's DMA are the builtin, not inline asm: the compiler's wait-count pass must know that the q
        // fragments have landed as well, or it waits for them -- and with them for the DMA just issued -- at their first
        // use in the key loop.  For the same reason the loads, the stores and the wait that leaves the stores in flight sit
        // in one block under one condition (with more conditions hipcc's control flow grows paths on which its model
        // sees a q load without a wait).
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < D / 16; ++s) asm volatile("" : "+v"(qf[s]));  // (pins the loads above the wait: hipcc sank one below it)
#pragma unroll 1
        while (cur.live) {
            const bool last_half = chalf == cur.nh - 1;
            const bool mine_now = (mine >> c_ord) & 1ull;
            const bool active = wave * 32 < cur.len;  // wave-uniform; a wave without queries still stages
            // this wave's part of the stage's tiles has landed (the wait at the end of the previous trip) -> everybody's has,
            // and every wave is done with the other buffer (a bare s_barrier: __syncthreads() would put s_waitcnt vmcnt(0)
            // in front of it and wait for the ctx stores as well)
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // the stage to issue under this one: this item's other half, or the next item's first
            const Item &d_it = last_half ? nxt : cur;
            const int d_half = last_half ? 0 : chalf + 1;
            if (chalf == 0) {
#pragma unroll
                for (int t = 0; t < D / 32; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
#pragma unroll
                for (int hh = 0; hh < HP; ++hh) {
                    m_run[hh] = -1e30f;
                    l_run[hh] = 0.0f;
                }
            }
            const int len = cur.len;
            const int full_blocks = len / 32;  // key blocks without padding keys
            const int kb0 = chalf * (kAttnStage / 32);
            const int nkb = mine_now && active && !no_keys ? min(cur.sb / 32 - kb0, kAttnStage / 32) : 0;
            const char *kt = smem + cbuf * SB;
            // one block of 32 keys; TAIL: the block holds padding keys (only an item's last block can), masked in the scores
            // and in the V fragments -- a separate instance, so that the common one is one straight basic block
            auto key_block = [&](const int kbl, auto tail_tag) __attribute__((always_inline)) {
                constexpr bool TAIL = decltype(tail_tag)::value;
                const int kb = kb0 + kbl;
                const int rem = len - kb * 32 - 8 * h;  // this lane's keys 16 (r>>3) + (r&7) < rem are real
#pragma unroll
                for (int hh = 0; hh < HP; ++hh) {
                    // S^T tile: A-row j = (r&3) + 8*(r>>2) + 4h holds key kb*32 + pi(j) = kb*32 + 16*(r>>3) + 8h + (r&7); col = query l31
                    f32x16 sc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
#pragma unroll
                    for (int s = hh * (DH / 16); s < (hh + 1) * (DH / 16); ++s) {
                        const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(kt + (k_frag ^ (uint32_t)(s * 32)) + kbl * 32 * RB);
                        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sc, 0, 0, 0);
                    }
                    if (TAIL) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[r] = 16 * (r >> 3) + (r & 7) < rem ? sc[r] : -1e30f;  // reference: additive -10000 mask == exclusion in f32
                    }
                    if (SAFE) {
                        float bm = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
                        for (int r = 3; r < 15; r += 2) bm = fmaxf(fmaxf(bm, sc[r]), sc[r + 1]);
                        bm = fmaxf(bm, sc[15]);
                        bm = fmaxf(bm, __shfl_xor(bm, 32));
                        if (__builtin_amdgcn_ballot_w64(bm > m_run[hh]) != 0) {  // some query's max grew: rescale (rare later on)
                            const float m_new = fmaxf(m_run[hh], bm);
                            const float alpha = __builtin_amdgcn_exp2f(m_run[hh] - m_new);
                            l_run[hh] *= alpha;
                            m_run[hh] = m_new;
#pragma unroll
                            for (int t = hh * (DH / 32); t < (hh + 1) * (DH / 32); ++t)
#pragma unroll
                                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
                        }
                    }
                    typedef __attribute__((ext_vector_type(2))) float f32x2;
                    const f32x2 mm = {m_run[hh], m_run[hh]};
                    f32x2 ps2 = {0.0f, 0.0f};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 t = {sc[r], sc[r + 1]};
                        if (SAFE) t -= mm;
                        const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                        sc[r] = e[0];
                        sc[r + 1] = e[1];
                        ps2 += e;
                    }
                    l_run[hh] += ps2[0] + ps2[1];
                    // O^T += V^T P^T: k16 step s uses this lane's p[8s .. 8s+7] = keys kb*32 + 16s + 8h + (0..7)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        bf16x8 pf;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pf[e] = (__bf16)sc[8 * s + e];
#pragma unroll
                        for (int t = hh * (DH / 32); t < (hh + 1) * (DH / 32); ++t) {
                            bf16x8 vf = *reinterpret_cast<const bf16x8 *>(kt + (v_frag ^ (uint32_t)((kbl & 3) * 64 + s * 32)) + (kbl >> 2) * 256 +
                                                                          t * 32 * 512);
                            if (TAIL) {  // 0 * (another token's value, possibly not finite) must stay 0
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    if (16 * s + e >= rem) vf[e] = (__bf16)0.0f;
                            }
                            o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[t], 0, 0, 0);
                        }
                    }
                }
            };
            const int nfull = max(0, min(full_blocks - kb0, nkb));
            static_assert(NP + D / 16 <= kAttnStage / 32, "one DMA piece / q load per key block");
#pragma unroll
            for (int kbl = 0; kbl < kAttnStage / 32; ++kbl) {
                // the next stage's pieces, then (an item's first stage: the ctx stores load its last one) the next item's q
                if (kbl < NP) issue_piece(d_it, d_half, cbuf ^ 1, kbl);
                else if (kbl - NP < D / 16) {
                    if (chalf == 0) qn[kbl - NP < D / 16 ? kbl - NP : 0] = load_q1(nxt, kbl - NP);
                }
                if (kbl < nfull) key_block(kbl, std::false_type{});
            }
            for (int kbl = nfull; kbl < nkb; ++kbl) key_block(kbl, std::true_type{});
            if (last_half) {
                // a row sum outside (1e-30, 1e30): an exp2 (or P * v) may have overflowed, or the whole row underflowed
                // -> redo with the running maximum (the comparison is false for NaN as well)
                float inv[HP];
                bool bad = false;
#pragma unroll
                for (int hh = 0; hh < HP; ++hh) {
                    const float l_row = l_run[hh] + __shfl_xor(l_run[hh], 32);
                    bad = bad || __builtin_amdgcn_ballot_w64(!(l_row > 1.0e-30f && l_row < 1.0e30f)) != 0;
                    inv[hh] = 1.0f / l_row;
                }
                bad = bad && !SAFE && mode == 0 && active;
                if (bad) my_redo |= 1ull << c_ord;
                // next item's q (before the ctx stores: the wait for qn must not cover them)
#pragma unroll
                for (int s = 0; s < D / 16; ++s) qf[s] = qn[s];
                // O^T layout: col = query l31, row = dv (r&3) + 8*(r>>2) + 4h (+32t): 4 consecutive dv per register group.  The two
                // halves of the wave trade groups (v_permlane32_swap: lanes 32-63 of one register <-> lanes 0-31 of another), after
                // which lane (l31, h) holds dv 16 rgp + 8h .. + 7 of its row: 16-byte stores, half as many (the issue of the
                // 8-byte ones -- 32 lines touched per instruction -- cost 19 us of a 246 us launch at d = 64)
                asm volatile("" : "+v"(lane));
                const int vo = (wave * 32 + (lane & 31)) * pitch + (lane >> 5) * 16;
                const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(ctx + (size_t)cur.tok0 * hidden + cur.grp * D), 0, mine_now && !bad && c_live ? (uint32_t)((len - 1) * pitch + RB) : 0u,
                    0x00020000);
#pragma unroll
                for (int t = 0; t < D / 32; ++t)
#pragma unroll
                    for (int rgp = 0; rgp < 2; ++rgp) {
                        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                        u32x2 pa, pb;  // groups 2 rgp, 2 rgp + 1
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                            const float sc_ = inv[t / (DH / 32)];
                            const bf16x2 a2 = {(__bf16)(o[t][rgp * 8 + 2 * e] * sc_), (__bf16)(o[t][rgp * 8 + 2 * e + 1] * sc_)};
                            const bf16x2 b2 = {(__bf16)(o[t][rgp * 8 + 4 + 2 * e] * sc_), (__bf16)(o[t][rgp * 8 + 4 + 2 * e + 1] * sc_)};
                            pa[e] = __builtin_bit_cast(unsigned int, a2);
                            pb[e] = __builtin_bit_cast(unsigned int, b2);
                        }
                        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pa[0], pb[0], false, false);
                        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pa[1], pb[1], false, false);
                        const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_c, vo + (t * 32 + 16 * rgp) * 2, 0, 0);
                    }
                // the next stage's tiles (issued before this stage's key loop) have landed; the stores stay in flight
                __builtin_amdgcn_s_waitcnt(0x0F70 | NS);  // vmcnt(NS)
                // on to the next item; the one after it was fetched an item ago, the one after that is fetched now
                cur = nxt;
                c_ord = x_ord;
                nxt = nn;
                x_ord = nn_ord;
                nn_ord = pop();
                nn = fetch(nn_ord);
                chalf = 0;
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                chalf = 1;
            }
            asm volatile("" ::: "memory");
            cbuf ^= 1;
        }
    };

    const uint64_t all = n_my >= 64 ? ~0ull : (1ull << n_my) - 1ull;
    if (mode != 1) pass(std::false_type{}, all, all);
    else my_redo = all;  // MEMEX_HIP_DEBUG attn_safe=1: everything through the running-maximum loop
    if (mode >= 2) return;
    // items some wave of this workgroup has to redo (rare): staged again by everybody, computed by the waves that asked
    if (my_redo != 0ull && (tid & 63) == 0) __hip_atomic_fetch_or(wg_redo, my_redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const uint64_t redo = *wg_redo;
    const uint64_t redo_u = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(redo >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)redo);
    pass(std::true_type{}, redo_u, my_redo);
}

// ---------------------------------------------------------------------------------------------
// attention_short_kernel: passes whose longest sequence has <= 128 tokens (head dim 32) -- the window of the reference's default
// model.  attention_kernel above is built around 512-query sequences: 16 waves per item, keys streamed through LDS in
// 256-key stages behind a barrier; a 128-token sequence fills 4 of its waves and an item costs ~1.5-2 us of stage latency
// whatever its length (134 us per 131k rows at 128 tokens, 208 at 64, against 164 at 512).  Here an item (sequence, head) is
// one workgroup of up to four waves (32 queries each) in a plain grid: the item's K rows and V^T rows (8 KiB each at 128
// tokens) are copied to LDS once, behind ONE barrier, and every wave reads its fragments from there block by block and runs
// to its ctx stores without another synchronisation (SHARE = false: each wave fetches all of them itself in the MFMA's operand
// layouts, no LDS and no barrier at all -- four times the global loads and 144 VGPRs, slower wherever it was measured); at 109
// VGPRs a SIMD holds four such waves of different items, which is what hides the latency.  The ARITHMETIC is
// attention_kernel's, instruction for instruction (same MFMA sequences, no softmax shift on the fast path, the same row-sum
// range check with the same running-maximum redo, the same summation order and store conversion): bit-identical results
// (tests/test_encoder_gpu.py::test_short_sequence_passes_pair_heads).
// ---------------------------------------------------------------------------------------------
template <int NB, bool SHARE>  // NB: key blocks of 32 an item can have: 2 (sequences of <= 64 tokens) or 4; SHARE: K / V^T through LDS
__global__ __launch_bounds__(256) void attention_short_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ k,
                                                              const bf16_t *__restrict__ vt, int ldvt,
                                                              const AttnItem *__restrict__ plan, int hidden,
                                                              bf16_t *__restrict__ ctx, int mode) {
    constexpr int D = 32, RB = D * 2;
    const AttnItem it = plan[blockIdx.x];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int len = it.len;
    if (!SHARE && wave * 32 >= len) return;  // (no barrier in this form: waves without queries simply leave)
    const int l31 = lane & 31, h = lane >> 5;
    const int pitch = hidden * 2;
    const uint32_t nrec = (uint32_t)((len - 1) * pitch + RB);  // rows >= len of q / k read as zero, their ctx stores are dropped
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void *)(q + (size_t)it.tok0 * hidden + it.group * D), 0, nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void *)(k + (size_t)it.tok0 * hidden + it.group * D), 0, nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void *)(vt + (size_t)it.group * D * ldvt + it.tok0), 0,
                                                                           (uint32_t)(((size_t)D * ldvt - it.tok0) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void *)(ctx + (size_t)it.tok0 * hidden + it.group * D), 0, nrec, 0x00020000);
    const int vo_q = (wave * 32 + l31) * pitch + h * 16;
    bf16x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_q, vo_q + s * 32, 0, 0));
    const int pr = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);  // pi(l31): A-row j of a score tile holds key pi(j)
    const int vo_k = pr * pitch + h * 16;
    const int vo_v = (l31 * ldvt + 8 * h) * 2;
    const int nkb = (len + 31) / 32, full_blocks = len / 32;
    // every K and V^T fragment of the item is requested before the first MFMA: one memory round trip per wave instead of one
    // per key block (block by block, a 128-token pass ran 7 % behind the staged kernel with head pairs)
    bf16x8 kfr[NB][2], vfr[NB][2];
    constexpr int KP = 80, VP = NB * 64 + 16, NT = NB * 64;  // LDS pitches of a K row / a V^T row; NT threads = NB waves
    __shared__ __attribute__((aligned(16))) char sk[SHARE ? NB * 32 * KP + 32 * VP : 16];
    char *sv = sk + NB * 32 * KP;
    if (SHARE) {
        // the item's K rows (64 B each, pitch 80 B in LDS) and V^T rows (NB * 64 B each, pitch + 16 B): one cooperative copy,
        // ONE barrier, then every wave reads its fragments from LDS where it needs them (the padded pitches keep 16 lanes'
        // 16-byte reads on distinct banks) -- a quarter of the global loads of the form below, which fetches a head's K / V^T
        // once per wave, and no fragment held in registers across the key blocks
        u32x4 kg[2], vg[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * NT;                       // K: chunk c = row * 4 + part; V^T: chunk c = row * (NB * 4) + part
            kg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_k, (c >> 2) * pitch + (c & 3) * 16, 0, 0);
            const int vr = c / (NB * 4), vc = c % (NB * 4);
            vg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_v, (vr * ldvt + vc * 8) * 2, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * NT;
            *reinterpret_cast<u32x4 *>(sk + (c >> 2) * KP + (c & 3) * 16) = kg[i];
            const int vr = c / (NB * 4), vc = c % (NB * 4);
            *reinterpret_cast<u32x4 *>(sv + vr * VP + vc * 16) = vg[i];
        }
        __syncthreads();
        if (wave * 32 >= len) return;
    } else {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
        if (kb < nkb) {
#pragma unroll
            for (int s = 0; s < 2; ++s) kfr[kb][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_k, vo_k + kb * 32 * pitch + s * 32, 0, 0));
        }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
        if (kb < nkb) {
#pragma unroll
            for (int s = 0; s < 2; ++s) vfr[kb][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_v, vo_v + (kb * 32 + 16 * s) * 2, 0, 0));
        }
    }

    auto run = [&](auto safe_tag) __attribute__((always_inline)) -> bool {  // -> true: the row sums left the fast path's range
        constexpr bool SAFE = decltype(safe_tag)::value;
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.0f;
        float m_run = -1e30f, l_run = 0.0f;
        auto key_block = [&](const int kb, auto tail_tag) __attribute__((always_inline)) {
            constexpr bool TAIL = decltype(tail_tag)::value;
            const int rem = len - kb * 32 - 8 * h;  // this lane's keys 16 (r>>3) + (r&7) < rem are real
            bf16x8 kf[2], vf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if constexpr (SHARE) {
                    kf[s] = *reinterpret_cast<const bf16x8 *>(sk + (kb * 32 + pr) * KP + h * 16 + s * 32);
                    vf[s] = *reinterpret_cast<const bf16x8 *>(sv + l31 * VP + (kb * 32 + 16 * s + 8 * h) * 2);
                } else {
                    kf[s] = kfr[kb][s];
                    vf[s] = vfr[kb][s];
                }
            }
            f32x16 sc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 2; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], sc, 0, 0, 0);
            if (TAIL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[r] = 16 * (r >> 3) + (r & 7) < rem ? sc[r] : -1e30f;
            }
            if (SAFE) {
                float bm = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) bm = fmaxf(fmaxf(bm, sc[r]), sc[r + 1]);
                bm = fmaxf(bm, sc[15]);
                bm = fmaxf(bm, __shfl_xor(bm, 32));
                if (__builtin_amdgcn_ballot_w64(bm > m_run) != 0) {
                    const float m_new = fmaxf(m_run, bm);
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                    l_run *= alpha;
                    m_run = m_new;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] *= alpha;
                }
            }
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            const f32x2 mm = {m_run, m_run};
            f32x2 ps2 = {0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 t = {sc[r], sc[r + 1]};
                if (SAFE) t -= mm;
                const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                sc[r] = e[0];
                sc[r + 1] = e[1];
                ps2 += e;
            }
            l_run += ps2[0] + ps2[1];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (__bf16)sc[8 * s + e];
                bf16x8 v = vf[s];
                if (TAIL) {  // 0 * (another token's value, possibly not finite) must stay 0
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (16 * s + e >= rem) v[e] = (__bf16)0.0f;
                }
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, pf, o, 0, 0, 0);
            }
        };
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {  // (unrolled: the fragments are register arrays)
            if (kb < full_blocks) key_block(kb, std::false_type{});
            else if (kb < nkb) key_block(kb, std::true_type{});
        }
        const float l_row = l_run + __shfl_xor(l_run, 32);
        const bool bad = __builtin_amdgcn_ballot_w64(!(l_row > 1.0e-30f && l_row < 1.0e30f)) != 0;
        if (!SAFE && bad) return true;
        const float inv = 1.0f / l_row;
#pragma unroll
        for (int rgp = 0; rgp < 2; ++rgp) {  // (attention_kernel's store: the wave halves trade groups, 16-byte stores)
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
            u32x2 pa, pb;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                const bf16x2 a2 = {(__bf16)(o[rgp * 8 + 2 * e] * inv), (__bf16)(o[rgp * 8 + 2 * e + 1] * inv)};
                const bf16x2 b2 = {(__bf16)(o[rgp * 8 + 4 + 2 * e] * inv), (__bf16)(o[rgp * 8 + 4 + 2 * e + 1] * inv)};
                pa[e] = __builtin_bit_cast(unsigned int, a2);
                pb[e] = __builtin_bit_cast(unsigned int, b2);
            }
            const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pa[0], pb[0], false, false);
            const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pa[1], pb[1], false, false);
            const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_c, vo_q + (16 * rgp) * 2, 0, 0);
        }
        return false;
    };
    if (mode == 1) {
        (void)run(std::true_type{});
    } else if (run(std::false_type{})) {
        (void)run(std::true_type{});
    }
}

static size_t attn_lds(int d) { return (size_t)2 * 2 * kAttnStage * d * 2 + 16; }

// head groups of a stage: one head, or (d = 32) two adjacent heads -- q / k / ctx rows in 128-byte pieces instead of 64, half as
// many work items, each with the same fixed cost (a DMA stage, its barrier and wait).  At 512-token sequences the launch is
// bound by its exp2 and the paired form keeps one more register than it has (166 against 173 us per 131k tokens): one head.
// Short sequences are bound by the per-item cost instead -- 134 us per 131k tokens at 128-token sequences, 208 at 64, against
// 164 at 512 (profiles/r5_encoder_short_windows.txt) -- and pairs win: +0.8 % per pass at 256 tokens, +3.2 % at 128, +4.5 % at
// ragged U[32, 128], +7 % at 64 (profiles/r5_attention_pairs_short_windows.txt).  So the pass decides: pairs when its longest
// sequence has <= 256 tokens.  MEMEX_HIP_DEBUG attn_pair=1 / 0: always / never (tests, A/B).  Same arithmetic per head either way.
constexpr int kAttnPairMaxLen = 256;
// ... and attention_short_kernel (above) takes every pass whose longest sequence has <= 128 tokens (same box, full passes,
// profiles/r5_attention_short_kernel_ab.txt): 126.6k -> 130.4k sequences/s at 128 tokens, 170.7k -> 181.9k at U[32, 128],
// 449k -> 499k at 64 -- against the head pairs, which had been +3 .. +7 % on the one-head items themselves -- and one 16-token
// query 0.255 -> 0.240 ms (all-MiniLM-L6-v2), 0.486 -> 0.453 (L12): a plain grid of one workgroup per item beats sixteen-wave
// workgroups walking a list at every size.  (Its first two forms did not: fragments fetched per wave from global memory,
// 117.8k at 128 tokens; all fragments read from LDS up front and held in registers, 144 VGPRs, 125.5k.)
// MEMEX_HIP_DEBUG attn_short=0 / 1: never / as the rule says (tests, A/B); attn_short_lds=0: the per-wave global form.
constexpr int kAttnShortMaxLen = 128;
enum { ATTN_ONE = 0, ATTN_PAIR = 1, ATTN_SHORT = 2 };
static int attn_form(int heads, int d_head, int max_len, int B) {
    if (d_head != 32) return ATTN_ONE;
    if (max_len <= kAttnShortMaxLen && debug_flag("attn_short", 1) != 0) return ATTN_SHORT;  // (read per pass: tests switch these inside one process)
    if (heads % 2) return ATTN_ONE;
    const bool many = (long)B * heads >= 1024;  // a pass too small to fill the chip keeps more, smaller items
    const int env = debug_flag("attn_pair", -1);
    return (env >= 0 ? env == 1 : (many && max_len <= kAttnPairMaxLen)) ? ATTN_PAIR : ATTN_ONE;
}
int attention_groups(int heads, int d_head, int max_len, int B) { return attn_form(heads, d_head, max_len, B) == ATTN_PAIR ? heads / 2 : heads; }

hipError_t launch_attention(hipStream_t s, const bf16_t *q, const bf16_t *k, const bf16_t *vt, int ldvt, const void *plan, int B,
                            int heads, int d_head, int hidden, int max_len, bf16_t *ctx) {
    if (B < 1 || B > 1024 || (d_head != 32 && d_head != 64) || heads * d_head != hidden) return hipErrorInvalidValue;
    if ((size_t)d_head * 2 * ldvt * 2 > 0xFFFFFFF0ull || (size_t)kAttnQ * hidden * 2 > 0x7FFFFFFFull) return hipErrorInvalidValue;  // 32-bit buffer offsets
    static const int n_cu = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        return hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }();
    const int form = attn_form(heads, d_head, max_len, B);
    const bool pair = form == ATTN_PAIR;
    const int n_items = B * attention_groups(heads, d_head, max_len, B);
    const size_t lds = attn_lds(pair ? 64 : d_head);
    // MEMEX_HIP_DEBUG attn_safe=1: running-maximum loop only (tests compare it with the default fast path); 2 .. 6: measurement modes
    // (no key loop / no loads / no stores: wrong results)
    const int mode = [] {
        const int m = debug_flag("attn_safe", 0);
        return m >= 0 && m <= 6 ? m : 0;
    }();
    // a persistent grid: one 16-wave workgroup per CU (LDS: one at d = 64, register file: one at d = 32), each walks up to 64
    // items (its redo masks are 64 bits)
    const int G = n_cu;
    const AttnItem *items = reinterpret_cast<const AttnItem *>(plan);
    if (form == ATTN_SHORT && mode <= 1) {  // (the measurement modes 2 .. 6 belong to the staged kernel)
        const bool share = debug_flag("attn_short_lds", 1) != 0;  // K / V^T of an item through LDS (default) or per wave from global memory (0)
        if (max_len <= 64 && share) hipLaunchKernelGGL((attention_short_kernel<2, true>), dim3(n_items), dim3(128), 0, s, q, k, vt, ldvt, items, hidden, ctx, mode);
        else if (max_len <= 64) hipLaunchKernelGGL((attention_short_kernel<2, false>), dim3(n_items), dim3(128), 0, s, q, k, vt, ldvt, items, hidden, ctx, mode);
        else if (share) hipLaunchKernelGGL((attention_short_kernel<4, true>), dim3(n_items), dim3(256), 0, s, q, k, vt, ldvt, items, hidden, ctx, mode);
        else hipLaunchKernelGGL((attention_short_kernel<4, false>), dim3(n_items), dim3(256), 0, s, q, k, vt, ldvt, items, hidden, ctx, mode);
        return hipGetLastError();
    }
    for (int off = 0; off < n_items; off += 64 * G) {
        const int n = n_items - off < 64 * G ? n_items - off : 64 * G;
        const dim3 grid(n < G ? n : G);
        if (d_head == 32 && pair)
            hipLaunchKernelGGL((attention_kernel<32, 2>), grid, dim3(kAttnWaves * 64), lds, s, q, k, vt, ldvt, items + off, n, hidden, ctx, mode);
        else if (d_head == 32)
            hipLaunchKernelGGL((attention_kernel<32, 1>), grid, dim3(kAttnWaves * 64), lds, s, q, k, vt, ldvt, items + off, n, hidden, ctx, mode);
        else
            hipLaunchKernelGGL((attention_kernel<64, 1>), grid, dim3(kAttnWaves * 64), lds, s, q, k, vt, ldvt, items + off, n, hidden, ctx, mode);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// K5: pooling + L2 normalise
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void pool_kernel(const T *__restrict__ x, const int32_t *__restrict__ cu,
                                                    const int32_t *__restrict__ lens, int hidden, int pooling_cls,
                                                    int normalize, float *__restrict__ out) {
    // 1024 threads = hidden/8 column chunks (16-byte loads) x token stripes; partial sums meet in LDS
    __shared__ float s_part[8192];  // [stripe][hidden] partial sums (deterministic reduction order)
    __shared__ float s_red[16];
    const int b = blockIdx.x;
    const int tok0 = cu[b];
    const int len = pooling_cls ? 1 : lens[b];
    const int tid = threadIdx.x;
    const int nch = hidden / 8;              // column chunks (48 or 96)
    const int stripes = 1024 / nch;          // token stripes
    const int ch = tid % nch, st = tid / nch;
    if (st < stripes) {
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = st; t < len; t += stripes) {
            if constexpr (std::is_same<T, float>::value) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(x + (size_t)(tok0 + t) * hidden + ch * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(x + (size_t)(tok0 + t) * hidden + ch * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] += v0[e];
                    a[4 + e] += v1[e];
                }
            } else {
                const bf16x8 v = *reinterpret_cast<const bf16x8 *>(x + (size_t)(tok0 + t) * hidden + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += (float)v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s_part[st * hidden + ch * 8 + e] = a[e];
    }
    __syncthreads();
    const float denom = pooling_cls ? 1.0f : fmaxf((float)len, 1e-9f);  // sum(h*m) / clamp(sum(m), 1e-9)
    float v = 0.0f, ss = 0.0f;
    if (tid < hidden) {
        for (int i = 0; i < stripes; ++i) v += s_part[i * hidden + tid];
        v = v / denom;
        ss = v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((tid & 63) == 0) s_red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += s_red[i];
    const float sc = normalize ? 1.0f / fmaxf(sqrtf(tot), 1e-12f) : 1.0f;
    if (tid < hidden) out[(size_t)b * hidden + tid] = v * sc;
}

hipError_t launch_pool(hipStream_t s, const bf16_t *x, const float *xf, const int32_t *cu, const int32_t *lens, int B, int hidden,
                       int pooling_cls, int normalize, float *out) {
    if (hidden > 1024 || hidden % 8) return hipErrorInvalidValue;
    if (xf) hipLaunchKernelGGL(pool_kernel<float>, dim3(B), dim3(1024), 0, s, xf, cu, lens, hidden, pooling_cls, normalize, out);
    else hipLaunchKernelGGL(pool_kernel<bf16_t>, dim3(B), dim3(1024), 0, s, x, cu, lens, hidden, pooling_cls, normalize, out);
    return hipGetLastError();
}

hipError_t encoder_kernels_setup() {
    hipError_t e;
    if ((e = gemm_attr<EPI_BIAS, 2, 2, 2, 32, 4>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_BIAS_GELU, 2, 2, 2, 32, 4>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_QKV, 2, 2, 2, 32, 4>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_VT, 2, 2, 2, 32, 4>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_F32, 2, 2, 2, 32, 4>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_GELU_SPLIT, 2, 2, 2, 32, 4>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_F32_H, 2, 2, 2, 32, 4>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_GELU_SPLIT_H, 2, 2, 2, 32, 4>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_BIAS, 2, 4, 4, 32, 3>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_BIAS_GELU, 2, 4, 4, 32, 3>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_QKV, 2, 4, 4, 32, 3>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_VT, 2, 4, 4, 32, 3>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_F32, 2, 4, 4, 32, 3>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_BIAS_RES_LN, 2, 4, 2, 32, 4>()) != hipSuccess) return e;
    if ((e = gemm_attr<EPI_BIAS_RES_LN, 1, 8, 2, 32, 3>()) != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_kernel<32, 1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds(32));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_kernel<32, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds(64));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_kernel<64, 1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_lds(64));
}

}  // namespace mx
