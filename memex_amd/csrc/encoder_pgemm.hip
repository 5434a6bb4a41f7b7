// encoder_pgemm.hip -- the large-pass GEMM of the sentence encoder: C[M,N] = A[M,K] W[N,K]^T (+ epilogue), bf16
// MFMA, f32 accumulate.  Same contract, operands, k order and epilogue arithmetic as gemm_kernel
// (encoder_kernels.hip; the projections and MLP GEMMs of the BERT layer that replaces rust-bert's
// `model.encode`, reference lib/libmemex/src/llm/embedding.rs:109) -- bit-identical outputs -- with a different
// schedule, for passes of >= 32768 rows where gemm_kernel's barrier-per-k-tile loop leaves the matrix pipe ~1/3
// busy (every wave reads its fragments right behind the barrier and waits for them, all waves in lockstep):
//   * ONE persistent 512-thread workgroup per CU walks its list of 256 x 256 output tiles; the LDS-DMA stream
//     (global -> LDS, 1 KiB per wave-instruction) never stops at a tile boundary: while tile t's last k-tiles are
//     multiplied and its epilogue runs, the first k-tiles of tile t+1 are already landing;
//   * a k-tile is 64 deep (full 128-byte lines of the activation rows) and is staged as four 16 KiB UNITS named
//     after the phase that reads them (A-lo, B-lo, B-hi, A-hi: the 2 x 2 quadrants of a wave's 128 x 64 output
//     tile); a unit is re-staged for k-tile g+2 one phase after its last read, so 3 units (48 KiB) are always in
//     flight and ONE counted s_waitcnt vmcnt(6) per k-tile is the only DMA wait;
//   * the two wave rows (waves 0-3 / 4-7: one wave of each on every SIMD) run half a phase apart ("ping-pong"):
//     while one group issues its 8 MFMAs of a phase, the other reads the next phase's fragments, issues its two DMA
//     pieces and waits for its LDS reads -- the matrix pipe sees one wave's MFMAs after the other's, the LDS
//     latency of a wave is covered by its partner's MFMAs, s_barrier is the metronome (2 per phase);
//   * fragment reads are conflict-free ds_read_b128 from unpadded 128-byte rows (16-byte chunk c of row r at
//     c ^ ((r >> 1) & 7), applied on the DMA's source side); 8 MFMAs per 6 fragment reads;
//   * epilogue per wave through 4 KiB of private LDS (no barrier): bias / GELU / q scale -> bf16 -> 128-byte row
//     segments -> 16-byte coalesced stores of full lines.
// EPI_BIAS_RES (the two Add & LayerNorm GEMMs of a hidden-768 layer on large passes): out = bf16(bf16(acc + bias) +
// residual), the LayerNorm itself runs as ln_rows_kernel afterwards (encoder_kernels.hip) -- a 256-column tile does not
// see whole rows, and the 64 x 768 tiles that do (gemm_kernel's EPI_BIAS_RES_LN) run at 0.21-0.26 of the MFMA peak
// against 0.41-0.52 here; the extra 2 x 201 MB of HBM traffic per LayerNorm cost less than that.
// MFMA operand roles as in gemm_kernel: the weight fragment is the A operand (D^T: a lane owns one output row and 4
// consecutive columns per register group), swapped for the feature-major V projection.
#include <cstdlib>
#include <type_traits>

#include "encoder_kernels.h"
#include "mx_gelu.h"

// Ablation switch for scripts/gemm_ubench.hip only (0 = production kernel); bits:
//   1 = no DMA inside the loop (the ring keeps the prologue's tiles), 2 = no epilogue (accumulators kept alive),
//   4 = no stagger between the wave rows, 8 = s_setprio 1 around the MFMA sections (measured slower: 306 against 281 us
//   on the QK projection of a hidden-768 layer), 16 = no bias loads in the epilogue (a constant)
// (bit 32: MFMA shape probe, 16x16x32 on the same operand traffic)
#ifndef MX_PGEMM_ABLATE
#define MX_PGEMM_ABLATE 0
#endif

namespace mx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((address_space(3))) void lds_void;

namespace {

constexpr int kPT = 256;                      // tile rows = tile columns
constexpr int kPK = 64;                       // k-tile depth
constexpr int kUnit = 128 * 128;              // 16 KiB: 128 rows x 128 B
constexpr int kBuf = 4 * kUnit;               // one k-tile: A-lo | B-lo | B-hi | A-hi
constexpr int kRingBytes = 2 * kBuf;          // 128 KiB
constexpr int kScratch = 4096;                // per-wave epilogue tile: 32 rows x 128 B
constexpr int kPLds = kRingBytes + 8 * kScratch;  // 160 KiB: the whole LDS of a CU
constexpr uint32_t kU_ALO = 0, kU_BLO = kUnit, kU_BHI = 2 * kUnit, kU_AHI = 3 * kUnit;

#define MX_PG_DMA(rsrc, ldsoff, voff, soff, imm) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lds_void *)(smem + (ldsoff)), 16, (voff), (soff), (imm), 0)

}  // namespace

template <int EPI_>
__global__ __launch_bounds__(512, 2) void pgemm_kernel(const GemmParams p, const int skew) {
    // the fp16 variants (EPI_F32_H, EPI_GELU_SPLIT_H: the mixed mode's MLP, two fp16 products per product) are their bf16
    // namesakes with v_mfma_f32_32x32x16_f16 and a two-block fp16 image of the GELU output
    constexpr bool F16 = EPI_ == EPI_F32_H || EPI_ == EPI_GELU_SPLIT_H;
    constexpr int EPI = EPI_ == EPI_F32_H ? (int)EPI_F32 : EPI_ == EPI_GELU_SPLIT_H ? (int)EPI_GELU_SPLIT : EPI_;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool FM = (EPI == EPI_VT);  // feature-major output, swapped MFMA roles
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;

    // ---- this workgroup's tiles.  Workgroup b runs on XCD b % 8 (observed dispatch rule, speed only): XCD x owns the
    // m-tiles x, x+8, ...; its entries e = (m-tile index, n-tile) are dealt round-robin to its G workgroups, so the G
    // tiles in flight on an XCD share 2-3 activation m-tiles and the weights in that XCD's L2.
    // (EPI_F32 also takes n = 256 t + 128: the last column tile is half valid -- its weight rows past w_rows alias into the next
    // k-block or fall outside the descriptor, and the two wave columns that own them skip their epilogue)
    const int n_tiles = (p.n + kPT - 1) / kPT, m_tiles = p.m / kPT;
    const int xcd = blockIdx.x & 7, G = gridDim.x >> 3;
    const int cnt_x = (m_tiles - xcd + 7) >> 3;
    const int total_e = cnt_x * n_tiles;
    const int e0 = blockIdx.x >> 3;
    const int my_tiles = e0 < total_e ? (total_e - e0 + G - 1) / G : 0;
    if (my_tiles == 0) return;
    const int nk = p.k / kPK;
    const int total_kt = my_tiles * nk;
    // Start-up skew (a measurement knob, 0 in production): every workgroup has the same work, so all CUs write their
    // 128 KiB output tiles at the same moments.  Spreading the starts (workgroup (b >> 3) & 15 starts that many units of
    // ~0.5 us late) buys nothing, it only adds the idle time (profiles/r4_pgemm_skew.txt); neither does a blocked tile
    // order (8 m-tiles x 4 n-tiles per window instead of row-major: the 1.67 GB the W1 launch fetches against 0.21 GB of
    // operands come out of the Infinity Cache at no measurable cost) nor leaving the output stores in flight across the
    // next k-tile's DMA wait (profiles/r4_pgemm_tile_order_and_wait.txt).
    for (int i = ((blockIdx.x >> 3) & 15) * skew; i > 0; --i) __builtin_amdgcn_s_sleep(16);

    // past the workgroup's last k-tile the descriptors get num_records = 0: the DMA operation still counts in vmcnt
    // (the waits stay constants) and touches no memory
    const uint32_t bytesA = (uint32_t)((size_t)p.m * p.lda * 2), bytesW = (uint32_t)((size_t)p.w_rows * p.k * 2);

    // ---- DMA pieces of this wave: pieces 2w, 2w+1 of every unit (8 rows x 128 B each).  Lane l -> unit row
    // u = 8 piece + (l >> 3), physical chunk l & 7 = logical chunk c ^ ((u >> 1) & 7).
    //   A-lo / A-hi: unit row u = tile row (u >> 6) * 128 + (u & 63) (+ 64): rows 0-63 / 64-127 of each wave row's half
    //   B-lo / B-hi: unit row u = tile column (u >> 5) * 64 + (u & 31) (+ 32): the first / second 32 columns of each wave
    // Weights are K-blocked [K/32][w_rows][32]: logical chunks 0-3 / 4-7 of a row are 64 B in k-blocks 2kt / 2kt+1.
    uint32_t vA[2], vW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = (wave * 2 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((u >> 1) & 7);
        vA[i] = (uint32_t)(((u >> 6) * 128 + (u & 63)) * p.lda * 2 + c * 16);
        vW[i] = (uint32_t)((((c >> 2) * p.w_rows) + (u >> 5) * 64 + (u & 31)) * 64 + (c & 3) * 16);
    }
    const uint32_t a_hi_off = (uint32_t)(64 * p.lda * 2);
    const uint32_t dstp = (uint32_t)wave * 2048u;  // LDS offset of this wave's first piece inside a unit

    // ---- staging cursor (runs two k-tiles ahead of the multiplication) and compute cursor
    int s_e = e0, s_kt = 0, s_idx = 0;  // entry, k-tile inside the tile, global k-tile index of the cursor
    uint32_t s_offA = 0, s_offW = 0;    // byte offsets of the cursor's k-tile in the activations / weights
    const uint32_t stepW = (uint32_t)(2 * p.w_rows * 64);
    auto cursor_tile = [&]() __attribute__((always_inline)) {
        const int mq = s_e / n_tiles, nt = s_e - mq * n_tiles;
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
    auto dmaA = [&](uint32_t lds_unit, bool live, uint32_t soff) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, live ? bytesA : 0u, 0x00020000);
        MX_PG_DMA(rs, lds_unit + dstp, vA[0], soff, 0);
        MX_PG_DMA(rs, lds_unit + dstp + 1024u, vA[1], soff, 0);
    };
    auto dmaW = [&](uint32_t lds_unit, bool live, uint32_t soff, int hi) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, live ? bytesW : 0u, 0x00020000);
        // (the second 32 columns sit 2048 B further; NOT the instruction's offset field: an LDS-DMA adds that to the LDS
        // address as well)
        const uint32_t so = soff + (hi ? 2048u : 0u);
        MX_PG_DMA(rs, lds_unit + dstp, vW[0], so, 0);
        MX_PG_DMA(rs, lds_unit + dstp + 1024u, vW[1], so, 0);
    };

    // ---- fragment read offsets inside a unit: row (wave's first row + 32 i + l31), k-step ks -> logical chunk
    // 2 ks + h -> physical (2 ks) ^ t, t = h ^ ((l31 >> 1) & 7).  They carry the k-tile's buffer (bit 16) and are
    // flipped after every k-tile.
    uint32_t a_o[4], b_o[4];
    {
        const uint32_t t = (uint32_t)(h ^ ((l31 >> 1) & 7));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t sw = ((uint32_t)(ks << 5)) ^ (t << 4);
            a_o[ks] = (uint32_t)((wr * 64 + l31) * 128) + sw;
            b_o[ks] = (uint32_t)((wc * 32 + l31) * 128) + sw;
        }
    }

    f32x16 acc[4][2];  // [i: 32-row m block][j: 32-column n block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    bf16x8 af[2][4], blo[4], bhi[4];  // activation fragments of the current row half [i][ks]; weight fragments [ks]

    // ---- epilogue of one finished tile (wave-private LDS scratch, no barrier)
    char *sc = smem + kRingBytes + wave * kScratch;
    auto epilogue = [&](int e_done) __attribute__((always_inline)) {
#if (MX_PGEMM_ABLATE & 2) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));
        return;
#endif
        const int mq = e_done / n_tiles, nt = e_done - mq * n_tiles;
        const int m0 = (xcd + 8 * mq) * kPT, n0 = nt * kPT;
        if ((EPI == EPI_F32 || EPI == EPI_VT) && n0 + wc * 64 >= p.n) {
            // wave-uniform: this wave's 64 columns lie past the matrix -- nothing to write (the accumulators are cleared below)
        } else if (FM) {
            // D[m][n]: lane owns column n = l31 of block j, 4 consecutive rows m per register group
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ncol = n0 + wc * 64 + j * 32;
#if MX_PGEMM_ABLATE & 16
                const float b = 0.01f;
#else
                const float b = p.bias[ncol + l31];
#endif
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {  // 64 m per pass: blocks 2ip, 2ip+1
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            bf16x4 pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(acc[ip * 2 + ii][j][rg * 4 + e] + b);
                            *reinterpret_cast<bf16x4 *>(sc + l31 * 128 + (((ii * 4 + rg) ^ (l31 & 7)) << 4) + h * 8) = pk;
                        }
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int row = ps * 8 + (lane >> 3), pc = lane & 7, lc = pc ^ (row & 7);
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(sc + row * 128 + pc * 16);
                        *reinterpret_cast<u32x4 *>(p.out_vt + (size_t)(ncol + row) * p.ldvt + m0 + wr * 128 + ip * 64 + lc * 8) = v;
                    }
                }
            }
        } else {
            bf16_t *dst = p.out;
            int ncol0 = n0 + wc * 64;  // first column of this wave in the output matrix
            float oscale = 1.0f;
            if (EPI == EPI_QKV) {
                const int part = ncol0 / p.hidden;  // 0 = q, 1 = k (wave-uniform: 64 divides hidden)
                dst = part == 0 ? p.out : p.out_k;
                ncol0 -= part * p.hidden;
                oscale = part == 0 ? p.qscale : 1.0f;
            }
            f32x4 b4[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#if MX_PGEMM_ABLATE & 16
                    b4[j][rg] = f32x4{0.01f, 0.01f, 0.01f, 0.01f};
#else
                    b4[j][rg] = *reinterpret_cast<const f32x4 *>(p.bias + n0 + wc * 64 + j * 32 + 8 * rg + 4 * h);
#endif
            if constexpr (EPI == EPI_F32) {
                // MX_PREC_BF16X3: bias + f32 out.  The wave's scratch holds 32 rows x 128 B = one 32-column block in f32: two
                // passes per row block, 16-byte chunk (2 rg + h) of row l31 at chunk ^ (l31 & 7)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e] + b4[j][rg][e];
                            *reinterpret_cast<f32x4 *>(sc + l31 * 128 + (((rg * 2 + h) ^ (l31 & 7)) << 4)) = v;
                        }
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const int row = ps * 8 + (lane >> 3), pc = lane & 7, lc = pc ^ (row & 7);
                            const u32x4 v = *reinterpret_cast<const u32x4 *>(sc + row * 128 + pc * 16);
                            *reinterpret_cast<u32x4 *>(p.out_f32 + (size_t)(m0 + wr * 128 + i * 32 + row) * p.ldo + ncol0 + j * 32 + lc * 4) = v;
                        }
                    }
            } else if constexpr (EPI == EPI_GELU_SPLIT) {
                // MX_PREC_BF16X3: bias + erf GELU at f32 accuracy, split into the [hi | lo | hi] column blocks the next GEMM
                // multiplies (blocks p.n apart): the hi pass and the lo pass go through the same scratch tile
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            // (the bias is fetched where it is used: 32 registers of it next to the GELU's temporaries spill)
                            const f32x4 bb = *reinterpret_cast<const f32x4 *>(p.bias + n0 + wc * 64 + j * 32 + 8 * rg + 4 * h);
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                const gelu_f32x2 g = gelu_erf2_precise(gelu_f32x2{acc[i][j][rg * 4 + e] + bb[e], acc[i][j][rg * 4 + e + 1] + bb[e + 1]});
                                acc[i][j][rg * 4 + e] = g[0];
                                acc[i][j][rg * 4 + e + 1] = g[1];
                            }
                        }
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        if (F16 && half == 1 && p.single) break;  // MX_PREC_MIXED1: one fp16 value per element (uniform over the launch)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) {
                                if constexpr (F16) {
                                    f16x4 pk;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float g = acc[i][j][rg * 4 + e];
                                        const _Float16 hi = (_Float16)fminf(fmaxf(g, -65504.0f), 65504.0f);
                                        pk[e] = half == 0 ? hi : (_Float16)(g - (float)hi);
                                    }
                                    *reinterpret_cast<f16x4 *>(sc + l31 * 128 + (((j * 4 + rg) ^ (l31 & 7)) << 4) + h * 8) = pk;
                                } else {
                                    bf16x4 pk;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float g = acc[i][j][rg * 4 + e];
                                        const __bf16 hi = (__bf16)g;
                                        pk[e] = half == 0 ? hi : (__bf16)(g - (float)hi);
                                    }
                                    *reinterpret_cast<bf16x4 *>(sc + l31 * 128 + (((j * 4 + rg) ^ (l31 & 7)) << 4) + h * 8) = pk;
                                }
                            }
#pragma unroll
                        for (int ps = 0; ps < 4; ++ps) {
                            const int row = ps * 8 + (lane >> 3), pc = lane & 7, lc = pc ^ (row & 7);
                            const u32x4 v = *reinterpret_cast<const u32x4 *>(sc + row * 128 + pc * 16);
                            bf16_t *o = dst + (size_t)(m0 + wr * 128 + i * 32 + row) * p.ldo + ncol0 + lc * 8;
                            if (half == 0) {
                                *reinterpret_cast<u32x4 *>(o) = v;
                                if (!F16) *reinterpret_cast<u32x4 *>(o + 2 * p.n) = v;  // (the mixed mode's image is [hi | lo])
                            } else {
                                *reinterpret_cast<u32x4 *>(o + p.n) = v;
                            }
                        }
                    }
                }
            } else
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // D^T[n][m]: lane owns row m = l31 of block i, 4 consecutive columns n per register group
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        bf16x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            gelu_f32x2 t = {acc[i][j][rg * 4 + e] + b4[j][rg][e], acc[i][j][rg * 4 + e + 1] + b4[j][rg][e + 1]};
                            if (EPI == EPI_BIAS_GELU) t = gelu_erf2(t);
                            pk[e] = (__bf16)(t[0] * oscale);
                            pk[e + 1] = (__bf16)(t[1] * oscale);
                        }
                        *reinterpret_cast<bf16x4 *>(sc + l31 * 128 + (((j * 4 + rg) ^ (l31 & 7)) << 4) + h * 8) = pk;
                    }
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
                        *reinterpret_cast<u32x4 *>(dst + (size_t)(m0 + wr * 128 + i * 32 + row) * p.ldo + ncol0 + lc * 8) = v;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    };

    auto mma = [&](const bf16x8 &w, const bf16x8 &a, f32x16 &c) __attribute__((always_inline)) {
        if constexpr (F16) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
        else if (FM) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w, c, 0, 0, 0);
        else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, c, 0, 0, 0);
    };
#if MX_PGEMM_ABLATE & 32  /* scripts/gemm_ubench.hip: MFMA shape probe -- two k-steps' operand traffic on four v_mfma_f32_16x16x32_bf16 (values meaningless) */
    auto mma2 = [&](const bf16x8 &w0, const bf16x8 &w1, const bf16x8 &a0, const bf16x8 &a1, f32x16 &c) __attribute__((always_inline)) {
        typedef __attribute__((ext_vector_type(4))) float f32x4p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4p t = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16((q & 1) ? w1 : w0, (q >> 1) ? a1 : a0, t, 0, 0, 0);
            c[4 * q] = t[0], c[4 * q + 1] = t[1], c[4 * q + 2] = t[2], c[4 * q + 3] = t[3];
        }
    };
#endif
    auto section_end = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto reads_done = [&]() __attribute__((always_inline)) {  // this wave's fragment reads have returned -> their unit may be re-staged
        __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0) (the builtin: the wait-count pass sees it and adds none of its own)
        __builtin_amdgcn_sched_barrier(0);
    };
    auto prio = [&](int on) __attribute__((always_inline)) {
#if MX_PGEMM_ABLATE & 8
        if (on) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
#endif
    };

    // ---- prologue: units A-lo, B-lo, B-hi, A-hi of k-tile 0 and A-lo, B-lo, B-hi of k-tile 1 (A-hi(1) follows in
    // phase 0 of k-tile 0, as in the steady state)
    uint32_t offA_p1;  // activation offset of k-tile g+1 (its A-hi unit is staged in phase 0 of k-tile g)
    bool live_p1;
    {
        dmaA(kU_ALO, true, s_offA);
        dmaW(kU_BLO, true, s_offW, 0);
        dmaW(kU_BHI, true, s_offW, 1);
        dmaA(kU_AHI, true, s_offA + a_hi_off);
        cursor_next();  // k-tile 1
        live_p1 = s_idx < total_kt;
        offA_p1 = s_offA;
        dmaA(kBuf + kU_ALO, live_p1, s_offA);
        dmaW(kBuf + kU_BLO, live_p1, s_offW, 0);
        dmaW(kBuf + kU_BHI, live_p1, s_offW, 1);
        cursor_next();  // k-tile 2
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        section_end();
#if !(MX_PGEMM_ABLATE & 4)
        if (wr == 1) section_end();  // the second wave row runs half a phase behind from here on
#endif
    }

    // ---- the k-tile loop: phases (A-lo, B-lo) (A-lo, B-hi) (A-hi, B-hi) (A-hi, B-lo); per phase a LOAD section
    // (fragment reads of the phase, two DMA pieces, reads returned) and an MFMA section, a barrier behind each.
    // B0 / B1: LDS offset of this k-tile's buffer / the other one.
    int c_e = e0, c_kt = 0;   // compute cursor
    int e_done = -1;          // tile whose accumulators are complete and not yet written (-1: none)
    uint32_t B0 = 0, B1 = kBuf;
#pragma unroll 1
    for (int g = 0; g < total_kt; ++g) {
        const bool live_p2 = s_idx < total_kt;  // k-tile g+2 exists
        // ---------------- phase 0: load A-lo, B-lo; stage A-hi(g+1) ----------------
        // A finished tile is written HERE, behind the first barrier that both wave rows have passed since their last
        // MFMA section on it: the leading row (0) takes its load-section barrier first and reads its fragments after the
        // epilogue, the trailing row (1) just starts its load section with the epilogue -- both epilogues run at the
        // same time, the stagger between the rows stays what it was.
        const bool lead_epi = (MX_PGEMM_ABLATE & 4) ? false : (wr == 0 && e_done >= 0);
#if !(MX_PGEMM_ABLATE & 1)
        if (lead_epi) dmaA(B1 + kU_AHI, live_p1, offA_p1 + a_hi_off);
#endif
        if (lead_epi) section_end();
        if (e_done >= 0) {
            epilogue(e_done);
            e_done = -1;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) blo[ks] = *reinterpret_cast<const bf16x8 *>(smem + kU_BLO + b_o[ks]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8 *>(smem + kU_ALO + a_o[ks] + i * 4096);
#if !(MX_PGEMM_ABLATE & 1)
        if (!lead_epi) dmaA(B1 + kU_AHI, live_p1, offA_p1 + a_hi_off);
#endif
        reads_done();
        if (!lead_epi) section_end();
        prio(1);
#if MX_PGEMM_ABLATE & 32
#pragma unroll
        for (int kp = 0; kp < 4; kp += 2) {
            mma2(blo[kp], blo[kp + 1], af[0][kp], af[0][kp + 1], acc[0][0]);
            mma2(blo[kp], blo[kp + 1], af[1][kp], af[1][kp + 1], acc[1][0]);
        }
#else
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            mma(blo[ks], af[0][ks], acc[0][0]);
            mma(blo[ks], af[1][ks], acc[1][0]);
        }
#endif
        prio(0);
        section_end();
        // ---------------- phase 1: load B-hi; stage A-lo(g+2) ----------------
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bhi[ks] = *reinterpret_cast<const bf16x8 *>(smem + kU_BHI + b_o[ks]);
#if !(MX_PGEMM_ABLATE & 1)
        dmaA(B0 + kU_ALO, live_p2, s_offA);
#endif
        reads_done();
        section_end();
        prio(1);
#if MX_PGEMM_ABLATE & 32
#pragma unroll
        for (int kp = 0; kp < 4; kp += 2) {
            mma2(bhi[kp], bhi[kp + 1], af[0][kp], af[0][kp + 1], acc[0][1]);
            mma2(bhi[kp], bhi[kp + 1], af[1][kp], af[1][kp + 1], acc[1][1]);
        }
#else
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            mma(bhi[ks], af[0][ks], acc[0][1]);
            mma(bhi[ks], af[1][ks], acc[1][1]);
        }
#endif
        prio(0);
        section_end();
        // ---------------- phase 2: load A-hi; stage B-lo(g+2) ----------------
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) af[i][ks] = *reinterpret_cast<const bf16x8 *>(smem + kU_AHI + a_o[ks] + i * 4096);
#if !(MX_PGEMM_ABLATE & 1)
        dmaW(B0 + kU_BLO, live_p2, s_offW, 0);
#endif
        reads_done();
        section_end();
        prio(1);
#if MX_PGEMM_ABLATE & 32
#pragma unroll
        for (int kp = 0; kp < 4; kp += 2) {
            mma2(bhi[kp], bhi[kp + 1], af[0][kp], af[0][kp + 1], acc[2][1]);
            mma2(bhi[kp], bhi[kp + 1], af[1][kp], af[1][kp + 1], acc[3][1]);
        }
#else
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            mma(bhi[ks], af[0][ks], acc[2][1]);
            mma(bhi[ks], af[1][ks], acc[3][1]);
        }
#endif
        prio(0);
        section_end();
        // ---------------- phase 3: stage B-hi(g+2); everything of k-tile g+1 has landed ----------------
#if !(MX_PGEMM_ABLATE & 1)
        dmaW(B0 + kU_BHI, live_p2, s_offW, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#endif
        live_p1 = live_p2;
        offA_p1 = s_offA;
        cursor_next();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {  // the next k-tile lives in the other buffer
            a_o[ks] ^= (uint32_t)kBuf;
            b_o[ks] ^= (uint32_t)kBuf;
        }
        B0 ^= (uint32_t)kBuf;
        B1 ^= (uint32_t)kBuf;
        section_end();
        prio(1);
#if MX_PGEMM_ABLATE & 32
#pragma unroll
        for (int kp = 0; kp < 4; kp += 2) {
            mma2(blo[kp], blo[kp + 1], af[0][kp], af[0][kp + 1], acc[2][0]);
            mma2(blo[kp], blo[kp + 1], af[1][kp], af[1][kp + 1], acc[3][0]);
        }
#else
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            mma(blo[ks], af[0][ks], acc[2][0]);
            mma(blo[ks], af[1][ks], acc[3][0]);
        }
#endif
        prio(0);
        section_end();
        if (++c_kt == nk) {  // the tile is complete
            c_kt = 0;
            e_done = c_e;
            c_e += G;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dead DMA operations must not outlive the workgroup's LDS
#if !(MX_PGEMM_ABLATE & 4)
    if (wr == 0) section_end();  // the leading wave row waits for the trailing one's last MFMA section (barrier balance)
#endif
    if (e_done >= 0) epilogue(e_done);
}

namespace {

template <int EPI>
hipError_t pgemm_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&pgemm_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, kPLds);
}

int g_pgemm_cus = 0;
int g_pgemm_skew = 0;  // start-up skew, a measurement knob of scripts/gemm_ubench.hip: 0 .. 8 measured, 0 is the fastest (profiles/r4_pgemm_skew.txt)

template <int EPI>
hipError_t pgemm_go(hipStream_t s, const GemmParams &p) {
    hipLaunchKernelGGL((pgemm_kernel<EPI>), dim3(g_pgemm_cus), dim3(512), kPLds, s, p, g_pgemm_skew);
    return hipGetLastError();
}

}  // namespace

hipError_t pgemm_setup() {
    hipError_t e;
    if ((e = pgemm_attr<EPI_BIAS>()) != hipSuccess) return e;
    if ((e = pgemm_attr<EPI_BIAS_GELU>()) != hipSuccess) return e;
    if ((e = pgemm_attr<EPI_QKV>()) != hipSuccess) return e;
    if ((e = pgemm_attr<EPI_VT>()) != hipSuccess) return e;
    if ((e = pgemm_attr<EPI_BIAS_RES>()) != hipSuccess) return e;
    if ((e = pgemm_attr<EPI_F32>()) != hipSuccess) return e;
    if ((e = pgemm_attr<EPI_GELU_SPLIT>()) != hipSuccess) return e;
    if ((e = pgemm_attr<EPI_F32_H>()) != hipSuccess) return e;
    if ((e = pgemm_attr<EPI_GELU_SPLIT_H>()) != hipSuccess) return e;
    int dev = 0;
    hipDeviceProp_t prop;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
    g_pgemm_cus = prop.multiProcessorCount / 8 * 8;  // one workgroup per CU, a multiple of the XCD count
    if (g_pgemm_cus < 8) g_pgemm_cus = 0;  // pgemm is optional: pgemm_supported() then says no and gemm_kernel runs every pass
    return hipSuccess;
}

// shapes pgemm_kernel takes: whole 256 x 256 tiles, k-tiles of 64, 32-bit byte offsets, the q / k split on a wave edge
bool pgemm_supported(int epi, const GemmParams &p) {
    if (epi != EPI_BIAS && epi != EPI_BIAS_GELU && epi != EPI_QKV && epi != EPI_VT && epi != EPI_BIAS_RES && epi != EPI_F32 &&
        epi != EPI_GELU_SPLIT && epi != EPI_F32_H && epi != EPI_GELU_SPLIT_H)
        return false;
    // whole column tiles, or for the f32-output epilogue a last tile of 128 columns (MX_PREC_BF16X3 at hidden 384: N = 1152 / 384;
    // A/B against gemm_kernel: profiles/r5_precise_partial_tile_ab.txt; the same for EPI_VT is worth +0.3 %: r5_vt_partial_tile_ab.txt)
    const bool n_ok = p.n % kPT == 0 || ((epi == EPI_F32 || epi == EPI_F32_H) && p.n % kPT == kPT / 2 && p.n > kPT / 2);
    if (g_pgemm_cus < 8 || p.m % kPT || !n_ok || p.k % kPK || p.k < 2 * kPK) return false;
    if ((size_t)p.m * p.lda * 2 >= (1ull << 32) || (size_t)p.w_rows * p.k * 2 >= (1ull << 32)) return false;
    if (epi == EPI_QKV && (p.hidden % 64 || p.n != 2 * p.hidden)) return false;
    return true;
}

hipError_t launch_pgemm(hipStream_t s, int epi, const GemmParams &p) {
    if (!pgemm_supported(epi, p)) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_BIAS: return pgemm_go<EPI_BIAS>(s, p);
        case EPI_BIAS_GELU: return pgemm_go<EPI_BIAS_GELU>(s, p);
        case EPI_QKV: return pgemm_go<EPI_QKV>(s, p);
        case EPI_VT: return pgemm_go<EPI_VT>(s, p);
        case EPI_BIAS_RES: return pgemm_go<EPI_BIAS_RES>(s, p);
        case EPI_F32: return pgemm_go<EPI_F32>(s, p);
        case EPI_GELU_SPLIT: return pgemm_go<EPI_GELU_SPLIT>(s, p);
        case EPI_F32_H: return pgemm_go<EPI_F32_H>(s, p);
        case EPI_GELU_SPLIT_H: return pgemm_go<EPI_GELU_SPLIT_H>(s, p);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mx
