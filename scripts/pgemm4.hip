// scripts/pgemm4.hip -- EXPERIMENT (round 6): pgemm_kernel's contract at the 1-wave-per-SIMD, 512-register design
// point.  C[M,N] = A[M,K] W[N,K]^T (+ epilogue), bf16 MFMA, f32 accumulate; same operands, k order and epilogue arithmetic as
// pgemm_kernel / gemm_kernel (memex_amd/csrc/encoder_pgemm.hip; the projections and MLP GEMMs behind `model.encode`, reference
// lib/libmemex/src/llm/embedding.rs:109) -- outputs must be bit-identical (scripts/gemm_ubench.hip compares every element).
//   * ONE persistent workgroup of FOUR waves per CU (one per SIMD, amdgpu_waves_per_eu(1,1): 512 registers per lane), 256 x 256
//     output tile, wave tile 128 x 128: 256 accumulator registers (AGPRs), 16 MFMAs per 8 fragment reads (pgemm_kernel: 8 per 6);
//   * a k-tile is 64 deep, staged as four 16-KiB units (A-lo, B-lo, B-hi, A-hi: the halves of a wave's rows / columns) in a ring
//     of two k-tiles; the wave walks the 2 x 2 quadrants of its tile in the order (lo,lo) (lo,hi) (hi,hi) (hi,lo), 16 MFMAs each,
//     and reads ONE unit's fragments per phase, one phase ahead of their use (no partner wave: the LDS latency is covered by the
//     wave's own MFMAs); four fragment sets A0 A1 B0 B1 rotate with a period of two k-tiles;
//   * ONE sync point per phase: lgkmcnt(0) (the unit read last phase is free), vmcnt(24) (the unit read this phase has landed:
//     every unit is staged 7 phases before it is read, 6 groups of 4 DMA pieces are younger), s_barrier, then the freed unit is
//     re-staged for k-tile g+2 -- the ring is a FIFO of 8 units with 6-7 in flight;
//   * the 256 accumulators live in AGPRs OWNED BY INLINE ASSEMBLY (a[16 (4 i + j) .. +15] = block (i, j)): left to hipcc's allocator the
//     kernel spills 110-180 registers inside the k-tile loop (every reload is a vmcnt wait that drains the DMA ring) -- the same
//     finding as tail2_kernel's in round 3; the compiler only ever sees the fragment VGPRs.  The first k-step of a tile multiplies
//     into C = 0 (no zeroing pass);
//   * epilogue as in pgemm_kernel (per wave through 4 KiB of private LDS, no barrier).
// Build / run: scripts/gemm_ubench.hip includes this file (-DMX_PGEMM4_ABLATE bits: 1 = no DMA in the loop, 2 = no epilogue, 4 = no
// global stores in the epilogue, 8 = no bias loads, 32 = every tile stages tile (0, 0)'s operands).
#include <cstdlib>
#include <type_traits>

#include "encoder_kernels.h"
#include "mx_gelu.h"

#ifndef MX_PGEMM4_ABLATE
#define MX_PGEMM4_ABLATE 0
#endif

namespace mx {
namespace p4 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kPT = 256;
constexpr int kPK = 64;
constexpr int kUnit = 128 * 128;
constexpr int kBuf = 4 * kUnit;
constexpr int kRingBytes = 2 * kBuf;
constexpr int kScratch = 4096;
constexpr int kLds = kRingBytes + 4 * kScratch;  // 144 KiB
constexpr uint32_t kU_ALO = 0, kU_BLO = kUnit, kU_BHI = 2 * kUnit, kU_AHI = 3 * kUnit;

#define MX_P4_DMA(rsrc, ldsoff, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lds_void *)(smem + (ldsoff)), 16, (voff), (soff), 0, 0)

#define MX_P4_MFMA(BASE, A, B) \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(A), "v"(B), "n"(BASE), "n"((BASE) + 15))
#define MX_P4_MFMA_Z(BASE, A, B) \
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(A), "v"(B), "n"(BASE), "n"((BASE) + 15))
#define MX_P4_C10_(a, b, c, d, e, f, g, h, i, j) "a" #a, "a" #b, "a" #c, "a" #d, "a" #e, "a" #f, "a" #g, "a" #h, "a" #i, "a" #j
#define MX_P4_C32(o) MX_P4_C10_(o##0, o##1, o##2, o##3, o##4, o##5, o##6, o##7, o##8, o##9)

template <int BASE>
__device__ __forceinline__ void acc_rd4(float (&d)[4]) {
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                 : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3])
                 : "n"(BASE), "n"(BASE + 1), "n"(BASE + 2), "n"(BASE + 3));
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void pgemm4_kernel(const GemmParams p, const int skew) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool FM = (EPI == EPI_VT);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // ---- this workgroup's tiles (pgemm_kernel's dealing: XCD x owns the m-tiles x, x+8, ...)
    const int n_tiles = p.n / kPT, m_tiles = p.m / kPT;
    const int xcd = blockIdx.x & 7, G = gridDim.x >> 3;
    const int cnt_x = (m_tiles - xcd + 7) >> 3;
    const int total_e = cnt_x * n_tiles;
    const int e0 = blockIdx.x >> 3;
    const int my_tiles = e0 < total_e ? (total_e - e0 + G - 1) / G : 0;
    if (my_tiles == 0) return;
    // start-up skew (measurement: do the workgroups' output bursts hurt because they come at the same time?): workgroup b starts
    // ((b >> 3) & 31) * skew ticks of 10 ns late
    if (skew > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), d = (unsigned long long)(((blockIdx.x >> 3) & 31) * skew);
        while (__builtin_amdgcn_s_memrealtime() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
    const int nk = p.k / kPK;  // even (pgemm4_supported)
    const int total_kt = my_tiles * nk;
    const uint32_t bytesA = (uint32_t)((size_t)p.m * p.lda * 2), bytesW = (uint32_t)((size_t)p.w_rows * p.k * 2);

    // ---- DMA pieces of this wave: pieces 4w .. 4w+3 of every unit (8 rows x 128 B each).  Lane l -> unit row u = 8 piece + (l >> 3),
    // physical chunk l & 7 = logical chunk c ^ ((u >> 1) & 7).  Unit row u of A-lo / B-lo = tile row / column (u >> 6) * 128 + (u & 63)
    // (+ 64 for A-hi / B-hi).  Weights are K-blocked [K/32][w_rows][32].
    uint32_t vA[4], vW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((u >> 1) & 7);
        const int r = (u >> 6) * 128 + (u & 63);
        vA[i] = (uint32_t)(r * p.lda * 2 + c * 16);
        vW[i] = (uint32_t)((((c >> 2) * p.w_rows) + r) * 64 + (c & 3) * 16);
    }
    const uint32_t a_hi_off = (uint32_t)(64 * p.lda * 2);
    const uint32_t w_hi_off = 64u * 64u;
    const uint32_t dstp = (uint32_t)wave * 4096u;

    // ---- staging cursor: the k-tile whose units are being issued
    int s_e = e0, s_kt = 0, s_idx = 0;
    uint32_t s_offA = 0, s_offW = 0;
    const uint32_t stepW = (uint32_t)(2 * p.w_rows * 64);
    auto cursor_tile = [&]() __attribute__((always_inline)) {
#if MX_PGEMM4_ABLATE & 32  // every tile stages the operands of tile (0, 0) of its XCD: the whole DMA stream hits in L2
        const int mq = 0, nt = 0;
#else
        const int mq = s_e / n_tiles, nt = s_e - mq * n_tiles;
#endif
        s_offA = (uint32_t)((xcd + 8 * mq) * kPT * p.lda * 2);
        s_offW = (uint32_t)((p.w_row0 + nt * kPT) * 64);
    };
    auto cursor_next = [&]() __attribute__((always_inline)) {
        ++s_idx;
        if (++s_kt == nk) {
            s_kt = 0;
            s_e += G;
            cursor_tile();
        } else {
            s_offA += kPK * 2;
            s_offW += stepW;
        }
    };
    cursor_tile();
    auto dmaA = [&](uint32_t lds_unit, uint32_t soff) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, s_idx < total_kt ? bytesA : 0u, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) MX_P4_DMA(rs, lds_unit + dstp + 1024u * i, vA[i], soff);
    };
    auto dmaW = [&](uint32_t lds_unit, uint32_t soff) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, s_idx < total_kt ? bytesW : 0u, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i) MX_P4_DMA(rs, lds_unit + dstp + 1024u * i, vW[i], soff);
    };

    // ---- fragment read offsets inside a unit (they carry the buffer bit of the k-tile being READ)
    uint32_t a_o[4], b_o[4];
    {
        const uint32_t t = (uint32_t)(h ^ ((l31 >> 1) & 7));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t sw = ((uint32_t)(ks << 5)) ^ (t << 4);
            a_o[ks] = (uint32_t)((wr * 64 + l31) * 128) + sw;
            b_o[ks] = (uint32_t)((wc * 64 + l31) * 128) + sw;
        }
    }

    // ---- the accumulators: a[16 (4 i + j) .. + 15] = block (i: 32-row m block, j: 32-column n block), owned by the assembly statements
    // below; this statement is what makes the kernel descriptor allocate a0 .. a255
    asm volatile("" ::: MX_P4_C32(), MX_P4_C32(1), MX_P4_C32(2), MX_P4_C32(3), MX_P4_C32(4), MX_P4_C32(5), MX_P4_C32(6), MX_P4_C32(7), MX_P4_C32(8), MX_P4_C32(9),
                 MX_P4_C32(10), MX_P4_C32(11), MX_P4_C32(12), MX_P4_C32(13), MX_P4_C32(14), MX_P4_C32(15), MX_P4_C32(16), MX_P4_C32(17), MX_P4_C32(18), MX_P4_C32(19),
                 MX_P4_C32(20), MX_P4_C32(21), MX_P4_C32(22), MX_P4_C32(23), MX_P4_C32(24), "a250", "a251", "a252", "a253", "a254", "a255");
    bf16x8 A0[2][4], A1[2][4], Ba[2][4], Bb[2][4];

    // ---- epilogue of one finished tile: pgemm_kernel's, over two 64-column halves of the wave's 128 columns
    char *sc = smem + kRingBytes + wave * kScratch;
    auto epilogue = [&](int e_done) __attribute__((always_inline)) {
#if MX_PGEMM4_ABLATE & 2
        return;
#endif
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results are readable
        const int mq = e_done / n_tiles, nt = e_done - mq * n_tiles;
        const int m0 = (xcd + 8 * mq) * kPT, n0 = nt * kPT;
        // (an opaque copy of the lane id: what the epilogue derives from it is computed HERE, not hoisted out of the k-tile loop
        // into registers that stay live across it)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int lane = lane_e, l31 = lane_e & 31, h = lane_e >> 5;
        static_for<0, 2>([&](auto jh_) {
            constexpr int jh = decltype(jh_)::value;
            if constexpr (FM) {
                static_for<0, 2>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    const int ncol = n0 + wc * 128 + jh * 64 + j * 32;
                    const float b = p.bias[ncol + l31];
                    static_for<0, 2>([&](auto ip_) {
                        constexpr int ip = decltype(ip_)::value;
                        static_for<0, 2>([&](auto ii_) {
                            constexpr int ii = decltype(ii_)::value;
                            static_for<0, 4>([&](auto rg_) {
                                constexpr int rg = decltype(rg_)::value;
                                float v[4];
                                acc_rd4<16 * (4 * (ip * 2 + ii) + jh * 2 + j) + rg * 4>(v);
                                bf16x4 pk;
#pragma unroll
                                for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(v[e] + b);
                                *reinterpret_cast<bf16x4 *>(sc + l31 * 128 + (((ii * 4 + rg) ^ (l31 & 7)) << 4) + h * 8) = pk;
                            });
                        });
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const int row = ps * 8 + (lane >> 3), pc = lane & 7, lc = pc ^ (row & 7);
                            const u32x4 v = *reinterpret_cast<const u32x4 *>(sc + row * 128 + pc * 16);
                            *reinterpret_cast<u32x4 *>(p.out_vt + (size_t)(ncol + row) * p.ldvt + m0 + wr * 128 + ip * 64 + lc * 8) = v;
                        }
                    });
                });
            } else {
                bf16_t *dst = p.out;
                int ncol0 = n0 + wc * 128 + jh * 64;
                float oscale = 1.0f;
                if (EPI == EPI_QKV) {
                    const int part = ncol0 / p.hidden;
                    dst = part == 0 ? p.out : p.out_k;
                    ncol0 -= part * p.hidden;
                    oscale = part == 0 ? p.qscale : 1.0f;
                }
                f32x4 b4[2][4];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
#if MX_PGEMM4_ABLATE & 8
                        b4[j][rg] = f32x4{0.01f, 0.01f, 0.01f, 0.01f};
#else
                        b4[j][rg] = *reinterpret_cast<const f32x4 *>(p.bias + n0 + wc * 128 + jh * 64 + j * 32 + 8 * rg + 4 * h);
#endif
                static_for<0, 4>([&](auto i_) {
                    constexpr int i = decltype(i_)::value;
                    static_for<0, 2>([&](auto j_) {
                        constexpr int j = decltype(j_)::value;
                        static_for<0, 4>([&](auto rg_) {
                            constexpr int rg = decltype(rg_)::value;
                            float v[4];
                            acc_rd4<16 * (4 * i + jh * 2 + j) + rg * 4>(v);
                            bf16x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                gelu_f32x2 t = {v[e] + b4[j][rg][e], v[e + 1] + b4[j][rg][e + 1]};
                                if (EPI == EPI_BIAS_GELU) t = gelu_erf2(t);
                                pk[e] = (__bf16)(t[0] * oscale);
                                pk[e + 1] = (__bf16)(t[1] * oscale);
                            }
                            *reinterpret_cast<bf16x4 *>(sc + l31 * 128 + (((j * 4 + rg) ^ (l31 & 7)) << 4) + h * 8) = pk;
                        });
                    });
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = ps * 8 + (lane >> 3), pc = lane & 7, lc = pc ^ (row & 7);
                        if (EPI == EPI_BIAS_RES) {
                            const size_t grow = (size_t)(m0 + wr * 128 + i * 32 + row);
                            const bf16x8 o = *reinterpret_cast<const bf16x8 *>(sc + row * 128 + pc * 16);
                            const bf16x8 rs = *reinterpret_cast<const bf16x8 *>(p.res + grow * p.ldres + ncol0 + lc * 8);
                            bf16x8 y;
#pragma unroll
                            for (int e = 0; e < 8; ++e) y[e] = (__bf16)((float)o[e] + (float)rs[e]);
                            *reinterpret_cast<bf16x8 *>(dst + grow * p.ldo + ncol0 + lc * 8) = y;
                        } else {
                            const u32x4 v = *reinterpret_cast<const u32x4 *>(sc + row * 128 + pc * 16);
#if MX_PGEMM4_ABLATE & 4
                            asm volatile("" ::"v"(v));
#else
                            *reinterpret_cast<u32x4 *>(dst + (size_t)(m0 + wr * 128 + i * 32 + row) * p.ldo + ncol0 + lc * 8) = v;
#endif
                        }
                    }
                });
            }
        });
    };

    // sync point: last phase's fragment reads are back (their unit is free), this phase's unit has landed: all but the N youngest
    // vector-memory operations are complete.  N = 24 (every unit is staged 7 phases before it is read: 6 groups of 4 DMA pieces are
    // younger); behind a tile boundary the epilogue's 32 stores sit among the young operations for seven sync points: N = 56 there,
    // so that the stores stay in flight under the next tile's first phases (loads and stores return in issue order on gfx9:
    // "at most N outstanding" then still means "everything older than the N youngest has returned")
    auto sync = [&](auto n_) __attribute__((always_inline)) {
        constexpr int N = decltype(n_)::value;
        __builtin_amdgcn_sched_barrier(0);
#if MX_PGEMM4_ABLATE & 1
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
#else
        __builtin_amdgcn_s_waitcnt(((N & 0x30) << 10) | 0x70 | (N & 0xF));  // vmcnt(N) lgkmcnt(0)
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // one phase: the quadrant (I0, J0) of the wave tile from the fragment sets A, B (16 MFMAs, in k-step order), with the 8 fragment
    // reads of the NEXT phase (unit `unit` through the offsets `o`, + 4096 for the second block) dealt out one per MFMA pair
    // MFMA operand roles as in pgemm_kernel: the weight fragment is the A operand, swapped for the feature-major V projection
#define MX_P4_MMA(Z, BASE, W, X)                                      \
    do {                                                              \
        if constexpr (FM) {                                           \
            if constexpr (Z) MX_P4_MFMA_Z(BASE, X, W);                \
            else MX_P4_MFMA(BASE, X, W);                              \
        } else {                                                      \
            if constexpr (Z) MX_P4_MFMA_Z(BASE, W, X);                \
            else MX_P4_MFMA(BASE, W, X);                              \
        }                                                             \
    } while (0)
    auto phase = [&](auto zero_, auto i0_, auto j0_, const bf16x8 (&A)[2][4], const bf16x8 (&B)[2][4], bf16x8 (&R)[2][4], uint32_t unit,
                     const uint32_t (&o)[4]) __attribute__((always_inline)) {
        constexpr bool ZERO = decltype(zero_)::value;
        constexpr int I0 = decltype(i0_)::value, J0 = decltype(j0_)::value;
        static_for<0, 4>([&](auto ks_) {
            constexpr int ks = decltype(ks_)::value;
            static_for<0, 2>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                // (read n = 2 ks + i of the next phase's 8: block n >> 2, k-step n & 3)
                constexpr int n = 2 * ks + i;
                R[n >> 2][n & 3] = *reinterpret_cast<const bf16x8 *>(smem + unit + o[n & 3] + (n >> 2) * 4096);
                MX_P4_MMA(ZERO && ks == 0, 16 * (4 * (I0 + i) + J0), B[0][ks], A[i][ks]);
                MX_P4_MMA(ZERO && ks == 0, 16 * (4 * (I0 + i) + J0 + 1), B[1][ks], A[i][ks]);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    using T_ = std::true_type;
    using F_ = std::false_type;

    // ---- prologue: all of k-tiles 0 and 1, fragments of A-lo(0) / B-lo(0), then A-lo(2)
    {
        dmaA(kU_ALO, s_offA);
        dmaW(kU_BLO, s_offW);
        dmaW(kU_BHI, s_offW + w_hi_off);
        dmaA(kU_AHI, s_offA + a_hi_off);
        cursor_next();
        dmaA(kBuf + kU_ALO, s_offA);
        dmaW(kBuf + kU_BLO, s_offW);
        dmaW(kBuf + kU_BHI, s_offW + w_hi_off);
        dmaA(kBuf + kU_AHI, s_offA + a_hi_off);
        cursor_next();
        __builtin_amdgcn_s_waitcnt(0x4078);  // vmcnt(24): A-lo(0), B-lo(0) have landed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                A0[i][ks] = *reinterpret_cast<const bf16x8 *>(smem + kU_ALO + a_o[ks] + i * 4096);
                Ba[i][ks] = *reinterpret_cast<const bf16x8 *>(smem + kU_BLO + b_o[ks] + i * 4096);
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        dmaA(kU_ALO, s_offA);  // A-lo(2)
    }

    uint32_t Bcur = 0;  // LDS offset of the buffer of the k-tile being multiplied
    // one k-tile; B0 holds its B-lo fragments on entry, B1 is free; on exit B1 holds B-lo of the next k-tile
    // n0_ .. n3_: the sync points' N
    auto ktile = [&](auto zero_, auto n0_, auto n1_, auto n2_, auto n3_, bf16x8 (&B0)[2][4], bf16x8 (&B1)[2][4]) __attribute__((always_inline)) {
        const uint32_t Both = Bcur ^ (uint32_t)kBuf;
        // ---- phase 0: (A-lo, B-lo); read B-hi; B-lo of this k-tile is free -> stage B-lo(g+2)
        sync(n0_);
#if !(MX_PGEMM4_ABLATE & 1)
        dmaW(Bcur + kU_BLO, s_offW);
#endif
        phase(zero_, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, A0, B0, B1, kU_BHI, b_o);
        // ---- phase 1: (A-lo, B-hi); read A-hi; stage B-hi(g+2)
        sync(n1_);
#if !(MX_PGEMM4_ABLATE & 1)
        dmaW(Bcur + kU_BHI, s_offW + w_hi_off);
#endif
        phase(zero_, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, A0, B1, A1, kU_AHI, a_o);
        // ---- phase 2: (A-hi, B-hi); read A-lo(g+1); stage A-hi(g+2)
        sync(n2_);
#if !(MX_PGEMM4_ABLATE & 1)
        dmaA(Bcur + kU_AHI, s_offA + a_hi_off);
#endif
        cursor_next();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {  // the reads move on to the next k-tile's buffer
            a_o[ks] ^= (uint32_t)kBuf;
            b_o[ks] ^= (uint32_t)kBuf;
        }
        phase(zero_, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, A1, B1, A0, kU_ALO, a_o);
        // ---- phase 3: (A-hi, B-lo); read B-lo(g+1) into B1; A-lo(g+1) is free -> stage A-lo(g+3)
        sync(n3_);
#if !(MX_PGEMM4_ABLATE & 1)
        dmaA(Both + kU_ALO, s_offA);
#endif
        phase(zero_, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, A1, B0, B1, kU_BLO, b_o);
        Bcur = Both;
    };

    int c_e = e0;
    using N24 = std::integral_constant<int, 24>;
#if MX_PGEMM4_ABLATE & 64  // (A/B: the stores waited for at the first sync point behind them)
    using N56 = std::integral_constant<int, 24>;
#else
    using N56 = std::integral_constant<int, 24 + 32>;
#endif
#pragma unroll 1
    for (int t = 0; t < my_tiles; ++t) {
        // the tile's first k-tile multiplies into C = 0; behind an epilogue its stores are among the young operations
        if (t == 0) {
            ktile(T_{}, N24{}, N24{}, N24{}, N24{}, Ba, Bb);
            ktile(F_{}, N24{}, N24{}, N24{}, N24{}, Bb, Ba);
        } else {
            ktile(T_{}, N56{}, N56{}, N56{}, N56{}, Ba, Bb);
            ktile(F_{}, N56{}, N56{}, N56{}, N24{}, Bb, Ba);
        }
#pragma unroll 1
        for (int kk = 2; kk < nk; kk += 2) {
            ktile(F_{}, N24{}, N24{}, N24{}, N24{}, Ba, Bb);
            ktile(F_{}, N24{}, N24{}, N24{}, N24{}, Bb, Ba);
        }
        __builtin_amdgcn_sched_barrier(0);
        epilogue(c_e);
        c_e += G;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dead DMA operations must not outlive the workgroup's LDS
}

template <int EPI>
hipError_t pgemm4_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&pgemm4_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
}

}  // namespace p4

inline bool pgemm4_supported(int epi, const GemmParams &p) {
    if (epi != EPI_BIAS && epi != EPI_BIAS_GELU && epi != EPI_QKV && epi != EPI_VT && epi != EPI_BIAS_RES) return false;
    if (p.m % p4::kPT || p.n % p4::kPT || p.k % (2 * p4::kPK) || p.k < 2 * p4::kPK) return false;
    if ((size_t)p.m * p.lda * 2 >= (1ull << 32) || (size_t)p.w_rows * p.k * 2 >= (1ull << 32)) return false;
    if (epi == EPI_QKV && (p.hidden % 64 || p.n != 2 * p.hidden)) return false;
    return true;
}

inline hipError_t launch_pgemm4(hipStream_t s, int epi, const GemmParams &p, int cus) {
    static const int skew = [] { const char *e = getenv("P4_SKEW"); return e ? atoi(e) : 0; }();
    if (!pgemm4_supported(epi, p)) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e;
        if ((e = p4::pgemm4_attr<EPI_BIAS>()) != hipSuccess) return e;
        if ((e = p4::pgemm4_attr<EPI_BIAS_GELU>()) != hipSuccess) return e;
        if ((e = p4::pgemm4_attr<EPI_QKV>()) != hipSuccess) return e;
        if ((e = p4::pgemm4_attr<EPI_VT>()) != hipSuccess) return e;
        if ((e = p4::pgemm4_attr<EPI_BIAS_RES>()) != hipSuccess) return e;
        attr_done = true;
    }
    switch (epi) {
        case EPI_BIAS: hipLaunchKernelGGL((p4::pgemm4_kernel<EPI_BIAS>), dim3(cus), dim3(256), p4::kLds, s, p, skew); break;
        case EPI_BIAS_GELU: hipLaunchKernelGGL((p4::pgemm4_kernel<EPI_BIAS_GELU>), dim3(cus), dim3(256), p4::kLds, s, p, skew); break;
        case EPI_QKV: hipLaunchKernelGGL((p4::pgemm4_kernel<EPI_QKV>), dim3(cus), dim3(256), p4::kLds, s, p, skew); break;
        case EPI_VT: hipLaunchKernelGGL((p4::pgemm4_kernel<EPI_VT>), dim3(cus), dim3(256), p4::kLds, s, p, skew); break;
        case EPI_BIAS_RES: hipLaunchKernelGGL((p4::pgemm4_kernel<EPI_BIAS_RES>), dim3(cus), dim3(256), p4::kLds, s, p, skew); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mx
