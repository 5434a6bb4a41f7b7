// encoder_tail.hip -- everything of a transformer layer after the attention core, as ONE kernel
// (hidden = 384 models: the MiniLM family memex runs):
//     x1  = LayerNorm1(x + Wo ctx + bo)                         attention output projection + Add&Norm
//     out = LayerNorm2(x1 + W2 gelu_erf(W1 x1 + b1) + b2)       MLP block + Add&Norm
// Part of the encoder forward that replaces rust-bert's `model.encode(&segments)` (reference
// lib/libmemex/src/llm/embedding.rs:109; BERT self-output / intermediate / output blocks, restated in
// oracle/bert_oracle.py).  Unfused this is three GEMMs whose [tokens, 384] and [tokens, ffn] intermediates
// go through HBM (x1: 2 x 100 MB, h: 2 x 403 MB per layer at 131k tokens) and whose epilogues are bound by
// that traffic.  Here a workgroup of 4 waves owns 64 token rows and nothing but ctx, x and out touches HBM.
// Same arithmetic, rounding points and k order as the three-GEMM path (bit-identical outputs;
// tail_kernel<false> is the MLP block alone, x1 given).
//   * the ctx / x1 tile (48 KiB) lives in LDS for the whole kernel: ctx is loaded once by LDS-DMA, the
//     out-projection reads it as its B operand, LayerNorm1 overwrites it with x1, which then is the B operand
//     of every G1 and the residual of LayerNorm2; the h tile (64 x 128 bf16) is double-buffered -> ONE
//     barrier per ffn chunk;
//   * the weights never touch LDS: they are stored as ONE STREAM PER WAVE in exactly the order the wave
//     consumes them (tail_stream_layout below; built once at upload): 1-KiB MFMA A-fragments (32 weight rows
//     x 16 k, lane-major), 72 for the out-projection, then 24 per G1 segment and 24 per G2 segment.  Every
//     wave loads its stream straight from L2 into a 12-deep register ring, 12 fragments ahead of use
//     (Wo + W1 + W2 = 2.7 MB per layer: L2-resident, and every workgroup walks them in the same order);
//   * the GELU epilogue E1 of chunk c+1 is spread over the k-steps of G2 of chunk c: its VALU work runs in
//     the shadow of the SAME wave's MFMAs instead of in a phase of its own (two identical workgroups on a CU
//     start together and stay in lockstep, so a VALU-only phase is not hidden by the neighbour);
//   * 80 KiB of LDS and <= 256 VGPRs per workgroup -> TWO independent workgroups per CU.
// Wave tiles: out-projection and G2 64 tokens x 96 features (6 MFMAs per k-step, 96 accumulator VGPRs), G1
// 64 x 32 (2 MFMAs per k-step).  LDS tiles are unpadded with an XOR swizzle of the 16-byte chunk index by
// (row & 15): fragment reads (ds_read_b128) and the E1 stores (ds_write_b64) are conflict-free.
// The kernel is power-bound, not issue-bound: every variant of it measured so far runs at the 1400 W cap
// (profiles/r2_power_tail_*.log), the clock settling between 1.55 and 2.15 GHz depending on how dense the
// MFMA issue is -- what pays is energy per token (fewer operand bytes moved, fewer VALU instructions).
#include "encoder_kernels.h"
#include "mx_gelu.h"
#include "mx_layernorm.h"

// Ablation switch for scripts/tail_ubench.hip only (0 = production kernel); bits:
//   1 = no weight loads in the loop (ring keeps the prologue's fragments), 2 = weight loads re-read one
//   fixed 12 KiB (L1-resident), 4 = E1 without the GELU arithmetic, 8 = no x/h fragment reads in the loop
#ifndef MX_TAIL_ABLATE
#define MX_TAIL_ABLATE 0
#endif
// scripts/tail_ubench.hip only: lane 0 of wave 0 stamps the 100 MHz real-time counter at phase boundaries
#ifndef MX_TAIL_TRACE
#define MX_TAIL_TRACE 0
#endif
#if MX_TAIL_TRACE
#define MX_TRACE(i)                                                                                         \
    do {                                                                                                    \
        if (p.trace && tid == 0) p.trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();  \
    } while (0)
#else
#define MX_TRACE(i) do { } while (0)
#endif

namespace mx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

namespace {

constexpr int kHid = 384;
constexpr int kBM = 64;                         // token rows per workgroup
constexpr int kFC = 128;                        // ffn features per chunk
constexpr int kXBytes = kBM * kHid * 2;         // 49152: x tile, row pitch 768 B, swizzled
constexpr int kHBytes = kBM * kFC * 2;          // 16384: one h tile, row pitch 256 B, swizzled
constexpr int kLds = kXBytes + 2 * kHBytes;     // 81920 = 80 KiB -> two workgroups per CU
constexpr int kOutPitch = kHid * 2 + 16;        // 784 B: final bf16 tile, row-major
static_assert(kBM * kOutPitch <= kLds, "final tile must fit");
constexpr int kT1 = kHid / 16;                  // 24 G1 k-steps per chunk
constexpr int kT2 = kFC / 16;                   // 8 G2 k-steps per chunk
constexpr int kFPC = kT1 + 3 * kT2;             // 48 weight fragments per wave and chunk
constexpr int kRing = 12;                       // fragments in flight per wave
static_assert(kFPC % kRing == 0, "ring slots must line up across chunks");
constexpr int kMaxF = 1536;
constexpr int kTP = kHid / 16;                  // 24 out-projection k-steps (3 fragments each)

}  // namespace

template <bool PO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void tail_kernel(const TailParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * kBM;
    const int F = p.f;
    const int nch = F / kFC;
    const uint32_t s = (uint32_t)(l31 & 15), sh = s ^ (uint32_t)h;
#if MX_TAIL_TRACE
    if (p.trace && tid == 0)  // [7] = where this workgroup runs: XCC id << 16 | HW_ID (se / sh / cu / simd / wave)
        p.trace[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                              (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    MX_TRACE(0);
    if ((int)blockIdx.x < p.skew_hi && (((int)blockIdx.x >> p.skew_shift) & 1))
        for (int i = 0; i < p.skew_iters; ++i) __builtin_amdgcn_s_sleep(127);  // 127 x 64 clocks ~ 4 us
    MX_TRACE(1);

    // ---- the weight stream of this wave (see the header comment): fragment n at byte n * 1024 + lane * 16.
    // Buffer loads: the lane offset is ONE VGPR for the whole kernel, the stream position an SGPR.
    const int lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void *)p.wf, 0, (uint32_t)(4 * (3 * kTP + nch * kFPC)) * 1024u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void *)p.b1, 0, (uint32_t)F * 4, 0x00020000);
    // byte offset of the current segment (the MLP-only kernel skips the out-projection's 72 fragments)
    int spos = wn * ((3 * kTP + nch * kFPC) * 1024) + (PO ? 0 : 3 * kTP * 1024);
    auto load_frag = [&](int i) __attribute__((always_inline)) -> bf16x8 {  // i: fragment index relative to the segment
#if MX_TAIL_ABLATE & 2
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, lane16, (i % kRing) * 1024, 0));
#else
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, lane16, spos + i * 1024, 0));
#endif
    };

    // ---- LDS addressing.  Logical 16-byte chunk c of row r lives at physical chunk c ^ (r & 15).
    // B-operand fragment of k-step t: lane (l31, h) holds token row l31 (+32 ii), k = 16 t + 8 h .. +7,
    // i.e. logical chunk 2t + h -> physical ((2t) ^ s ^ h) within its group of 16.
    // The swizzle term is recomputed at every k-step (one v_xor + one v_add per 2..6 MFMAs) from an OPAQUE
    // copy of shs: left to itself the compiler hoists all 16 distinct fragment addresses out of the chunk
    // loop and keeps them in VGPRs, which this kernel does not have (acc 128 + ring 48 + fragments 16).
    const uint32_t x_row = (uint32_t)l31 * (kHid * 2);
    const uint32_t h_row = (uint32_t)kXBytes + (uint32_t)l31 * (kFC * 2);
    const uint32_t shs = sh << 4;
    auto swz = [&](int b) __attribute__((always_inline)) -> uint32_t {  // ((2b) ^ sh) << 4
        uint32_t o = shs;
        asm volatile("" : "+v"(o));
        return o ^ (uint32_t)(b << 5);
    };
    auto read_x = [&](int t, bf16x8 (&dst)[2]) __attribute__((always_inline)) {
        const char *a = smem + x_row + swz(t & 7) + (t >> 3) * 256;
        dst[0] = *reinterpret_cast<const bf16x8 *>(a);
        dst[1] = *reinterpret_cast<const bf16x8 *>(a + 32 * kHid * 2);
    };
    auto read_h = [&](uint32_t hb, int t2, bf16x8 (&dst)[2]) __attribute__((always_inline)) {
        const char *a = smem + hb + h_row + swz(t2);
        dst[0] = *reinterpret_cast<const bf16x8 *>(a);
        dst[1] = *reinterpret_cast<const bf16x8 *>(a + 32 * kFC * 2);
    };

    f32x16 acc1[2];     // h chunk: [ii: 32-row token block]
    f32x16 acc2[2][3];  // y: [ii][j: 32-feature block]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.0f;
    }

    // ---- prologue: the first 12 weight fragments, then the ctx (PO) / x tile by LDS-DMA (12 KiB-ops per wave)
    const bf16_t *tile_src = PO ? p.ctx : p.x;
    const int tile_ld = PO ? p.ldc : p.ldx;
    bf16x8 ring[kRing];
#pragma unroll
    for (int f = 0; f < kRing; ++f) ring[f] = load_frag(f);
#pragma unroll
    for (int o = 0; o < 12; ++o) {
        const int P = (wn * 12 + o) * 64 + lane;  // physical chunk position in the tile
        const int r = P / 48, pc = P % 48;
        const int c = pc ^ (r & 15);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(tile_src + (size_t)(m0 + r) * tile_ld + c * 8),
                                         (lds_void_t *)(smem + __builtin_amdgcn_readfirstlane((wn * 12 + o) * 1024)), 16, 0, 0);
    }
    // (the builtin, not inline asm: the compiler's own wait-count bookkeeping must see that nothing is
    // pending here, or it drains the fragment ring at the top of every chunk)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    MX_TRACE(2);

    bf16x8 bfr[2][2];  // activation fragments of one k-step, double-buffered: [buf][ii]
    float b1v;         // b1 of this wave's 32 features of the chunk in acc1: lane l holds feature l & 31
    auto load_b1 = [&](int chunk) __attribute__((always_inline)) {
        b1v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsb, l31 * 4, (chunk * kFC + wn * 32) * 4, 0));
    };

    // ---------------- G1 segment: acc1[64 x 32 of this wave] = x * W1[chunk]^T (24 k-steps) ----------------
    // entry: bfr[0] = x fragments of k-step 0.  exit: bfr[0] = h fragments of k-step 0 of `hb_next`
    auto g1_segment = [&](uint32_t hb_next, bool prefetch_h) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < kT1; ++t) {
            const int cur = t & 1, nxt = cur ^ 1;
            if (MX_TAIL_ABLATE & 8) {
            } else if (t + 1 < kT1) {
                read_x(t + 1, bfr[nxt]);
            } else if (prefetch_h) {
                read_h(hb_next, 0, bfr[nxt]);
            }
            acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[t % kRing], bfr[cur][0], acc1[0], 0, 0, 0);
            acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[t % kRing], bfr[cur][1], acc1[1], 0, 0, 0);
#if !(MX_TAIL_ABLATE & 1)
            ring[t % kRing] = load_frag(t + kRing);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        spos += kT1 * 1024;
    };

    // ---------------- E1 piece: 4 values of h = gelu(acc1 + b1) -> bf16 -> h tile `hb` ----------------
    // Lane owns token row l31 of each 32-row block and features 8 rg + 4 h + (0..3) of the wave's 32.
    auto e1_bias = [&](int rg, float (&b4)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            b4[e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((8 * rg + 4 * h + e) * 4, __builtin_bit_cast(int, b1v)));
    };
    // half = 0 / 1: values 0,1 / 2,3 of the piece into pk; the caller stores pk after half 1
    auto e1_half = [&](int rg, int ii, int half, const float (&b4)[4], bf16x4 &pk) __attribute__((always_inline)) {
        const int e = 2 * half;
#if MX_TAIL_ABLATE & 4
        const gelu_f32x2 g = gelu_f32x2{acc1[ii][rg * 4 + e] + b4[e], acc1[ii][rg * 4 + e + 1] + b4[e + 1]};
#else
        const gelu_f32x2 g = gelu_erf2(gelu_f32x2{acc1[ii][rg * 4 + e] + b4[e], acc1[ii][rg * 4 + e + 1] + b4[e + 1]});
#endif
        pk[e] = (__bf16)g[0];
        pk[e + 1] = (__bf16)g[1];
        acc1[ii][rg * 4 + e] = 0.0f;
        acc1[ii][rg * 4 + e + 1] = 0.0f;
    };
    auto e1_store = [&](uint32_t hb, int rg, int ii, const bf16x4 &pk) __attribute__((always_inline)) {
        const uint32_t pc = (((uint32_t)(wn * 4 + rg)) ^ s) << 4;
        *reinterpret_cast<bf16x4 *>(smem + hb + h_row + ii * (32 * kFC * 2) + pc + h * 8) = pk;
    };
    auto e1_piece = [&](uint32_t hb, int rg, int ii, const float (&b4)[4]) __attribute__((always_inline)) {
        bf16x4 pk;
        e1_half(rg, ii, 0, b4, pk);
        e1_half(rg, ii, 1, b4, pk);
        e1_store(hb, rg, ii, pk);
    };

    // ---------------- G2 segment: y[64 x 96 of this wave] += h(hb_r) * W2[:, chunk]^T (8 k-steps), with the
    // E1 of the NEXT chunk (acc1 -> h tile hb_w) spread over its k-steps: the GELU runs in the shadow of
    // this wave's own MFMAs.  entry: bfr[0] = h fragments of k-step 0.  exit: bfr[0] = x fragments of k-step 0
    auto g2_segment = [&](uint32_t hb_r, uint32_t hb_w, bool with_e1) __attribute__((always_inline)) {
        float b4[4];
#pragma unroll
        for (int t2 = 0; t2 < kT2; ++t2) {
            const int cur = t2 & 1, nxt = cur ^ 1;
            if (MX_TAIL_ABLATE & 8) {
            } else if (t2 + 1 < kT2) {
                read_h(hb_r, t2 + 1, bfr[nxt]);
            } else {
                read_x(0, bfr[nxt]);
            }
            if (with_e1 && !(t2 & 1)) e1_bias(t2 >> 1, b4);
            bf16x4 pk;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int i = 3 * t2 + j;
                acc2[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % kRing], bfr[cur][0], acc2[0][j], 0, 0, 0);
                acc2[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % kRing], bfr[cur][1], acc2[1][j], 0, 0, 0);
#if !(MX_TAIL_ABLATE & 1)
                ring[i % kRing] = load_frag(i + kRing);
#endif
                if (with_e1) {  // one pair of GELUs behind each of the first two MFMA pairs, the store behind the third
                    if (j < 2) e1_half(t2 >> 1, t2 & 1, j, b4, pk);
                    else e1_store(hb_w, t2 >> 1, t2 & 1, pk);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        spos += 3 * kT2 * 1024;
    };

    // ---------------- Add & LayerNorm of a 64 x 384 accumulator tile (used twice) ----------------
    // pass 1: acc2 + bias -> bf16 staging tile [m][n] (row pitch 784 B) at LDS offset 0 (over the x tile);
    // pass 2: 4 threads per row, 12 chunks of 8 features each, interleaved chunk-wise: + residual rs[],
    // mean / variance over the 384 features, gamma / beta -> emit(c, bf16x8).  Same arithmetic as
    // gemm_kernel's EPI_BIAS_RES_LN epilogue.
    // (row / part are re-derived from an opaque copy of tid at each use: anything computed from them up
    // front would sit in VGPRs through the main loop, which has none to spare)
    constexpr int TPR = 256 / kBM, CPT = kHid / TPR / 8;  // 4 threads per row, 12 chunks per thread
    int ln_row = 0, ln_prt = 0;
    auto ln_ids = [&]() __attribute__((always_inline)) {
        int t = tid;
        asm volatile("" : "+v"(t));
        ln_row = t / TPR;
        ln_prt = t % TPR;
    };
    auto stage_acc2 = [&](const float *bias) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int nloc = wn * 96 + j * 32 + 8 * rg + 4 * h;
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + nloc);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    bf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(acc2[ii][j][rg * 4 + e] + b4[e]);
                    *reinterpret_cast<bf16x4 *>(smem + (ii * 32 + l31) * kOutPitch + nloc * 2) = pk;
                }
            }
    };
    auto ln_rows = [&](const bf16x8 (&rs)[CPT], const float *gamma, const float *beta, auto emit) __attribute__((always_inline)) {
        float y[CPT * 8];
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int col = (c * TPR + ln_prt) * 8;
            const bf16x8 o = *reinterpret_cast<const bf16x8 *>(smem + ln_row * kOutPitch + col * 2);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[c * 8 + e] = (float)o[e] + (float)rs[c][e];
        }
        float mean, rstd;
        ln_row_stats<TPR, CPT * 8>(y, p.eps, mean, rstd);
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int col = (c * TPR + ln_prt) * 8;
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
            emit(c, o);
        }
    };
    // this thread's 12 chunks of row ln_row in the swizzled x tile (logical chunk c * 4 + prt)
    auto x_tile_chunk = [&](int c) __attribute__((always_inline)) -> char * {
        return smem + ln_row * (kHid * 2) + ((((uint32_t)(c * TPR + ln_prt)) ^ (uint32_t)(ln_row & 15)) << 4);
    };

    if (PO) {
        // ---------------- out-projection: acc2[64 x 96 of this wave] = ctx * Wo^T (24 k-steps x 3 fragments),
        // then x1 = LayerNorm1(acc2 + bo + x) -> the x tile (over ctx)
        read_x(0, bfr[0]);
#pragma unroll
        for (int t = 0; t < kTP; ++t) {
            const int cur = t & 1, nxt = cur ^ 1;
            if (t + 1 < kTP) read_x(t + 1, bfr[nxt]);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int i = 3 * t + j;
                acc2[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % kRing], bfr[cur][0], acc2[0][j], 0, 0, 0);
                acc2[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % kRing], bfr[cur][1], acc2[1][j], 0, 0, 0);
                ring[i % kRing] = load_frag(i + kRing);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        spos += 3 * kTP * 1024;
        MX_TRACE(3);
        ln_ids();
        bf16x8 rs[CPT];  // the residual rows (layer input), straight from HBM, in flight during pass 1
#pragma unroll
        for (int c = 0; c < CPT; ++c)
            rs[c] = *reinterpret_cast<const bf16x8 *>(p.x + (size_t)(m0 + ln_row) * p.ldx + (c * TPR + ln_prt) * 8);
        __builtin_amdgcn_s_barrier();  // every wave is done with the ctx tile
        stage_acc2(p.bo);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.0f;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bf16x8 x1r[CPT];
        ln_rows(rs, p.ln1g, p.ln1b, [&](int c, const bf16x8 &o) __attribute__((always_inline)) { x1r[c] = o; });
        if (p.stop_after_ln1) {  // small passes (encoder_small.hip): x1 leaves here, the MLP runs split over the ffn dimension
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                *reinterpret_cast<bf16x8 *>(p.out + (size_t)(m0 + ln_row) * p.ldo + (c * TPR + ln_prt) * 8) = x1r[c];
            return;
        }
        __builtin_amdgcn_s_barrier();  // staging tile consumed: its space becomes the x1 tile
#pragma unroll
        for (int c = 0; c < CPT; ++c) *reinterpret_cast<bf16x8 *>(x_tile_chunk(c)) = x1r[c];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- stream order (= the order of p.wf): G1(0) | G1(1) | G2(0) G1(2) | G2(1) G1(3) | ... | G2(nch-2) | G2(nch-1)
    MX_TRACE(4);
    load_b1(0);
    read_x(0, bfr[0]);
    g1_segment(0u, false);
    {   // chunk 0: E1 with nothing to hide under
        float b4[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            e1_bias(rg, b4);
            e1_piece(0u, rg, 0, b4);
            e1_piece(0u, rg, 1, b4);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // h tile 0 complete
    load_b1(1);
    read_x(0, bfr[0]);
    g1_segment(0u, true);

#pragma unroll 1
    for (int chunk = 1; chunk + 1 < nch; ++chunk) {
        // G2(chunk - 1) reads h[(chunk-1) & 1]; E1(chunk) fills h[chunk & 1] from acc1 (= G1(chunk))
        const uint32_t hb_w = (uint32_t)(chunk & 1) * kHBytes, hb_r = hb_w ^ (uint32_t)kHBytes;
        g2_segment(hb_r, hb_w, true);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the only barrier of the chunk: h[chunk & 1] complete, h[(chunk-1) & 1] free
        load_b1(chunk + 1);
        g1_segment(hb_w, true);
    }
    {   // last chunk: its E1 under G2(nch - 2), then G2(nch - 1) with nothing left to hide
        const uint32_t hb_w = (uint32_t)((nch - 1) & 1) * kHBytes, hb_r = hb_w ^ (uint32_t)kHBytes;
        g2_segment(hb_r, hb_w, true);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        read_h(hb_w, 0, bfr[0]);
        g2_segment(hb_w, 0u, false);
    }
    // ---- LayerNorm2: out = LN(acc2 + b2 + x1).  The residual comes out of the x tile before the staging tile
    // overwrites it.
    __builtin_amdgcn_s_barrier();  // every wave is done with the x and h tiles
    MX_TRACE(5);
    {
        ln_ids();
        bf16x8 rs[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) rs[c] = *reinterpret_cast<const bf16x8 *>(x_tile_chunk(c));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage_acc2(p.b2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ln_rows(rs, p.gamma, p.beta, [&](int c, const bf16x8 &o) __attribute__((always_inline)) {
            *reinterpret_cast<bf16x8 *>(p.out + (size_t)(m0 + ln_row) * p.ldo + (c * TPR + ln_prt) * 8) = o;
        });
    }
    MX_TRACE(6);
}

// Host side of the layout.  For wave wn the stream is the segment list
//     PO | G1(0) | G1(1) | G2(0) G1(2) | G2(1) G1(3) | ... | G2(nch-2) | G2(nch-1)
// PO: 24 k-steps x 3 row groups of Wo (fragment 3 t + j: rows wn*96 + j*32 .. +31, k = 16 t ..); a G1(c)
// segment is the 24 k-steps of W1 rows c*128 + wn*32 .. +31; a G2(c) segment is 8 k-steps x 3 row groups
// (fragment 3 t2 + j: W2 rows wn*96 + j*32 .. +31, k = c*128 + 16 t2 ..).  Inside a fragment lane (h, r)
// holds row r, k offset 8h .. 8h+7.  out: tail_stream_elems(F) bf16.
size_t tail_stream_elems(int F) { return (size_t)4 * (3 * kTP + (F / kFC) * kFPC) * 512; }

void tail_stream_layout(const float *wo, const float *w1, const float *w2, int F, uint16_t *out, uint16_t (*to_bf16)(float)) {
    const int nch = F / kFC;
    size_t o = 0;
    for (int wn = 0; wn < 4; ++wn) {
        auto frag = [&](const float *w, size_t ld, int row0, int k0) {
            for (int ln = 0; ln < 64; ++ln)
                for (int e = 0; e < 8; ++e) out[o++] = to_bf16(w[(size_t)(row0 + (ln & 31)) * ld + k0 + 8 * (ln >> 5) + e]);
        };
        auto g1 = [&](int c) {
            for (int t = 0; t < kT1; ++t) frag(w1, kHid, c * kFC + wn * 32, 16 * t);
        };
        auto g2 = [&](int c) {
            for (int t2 = 0; t2 < kT2; ++t2)
                for (int j = 0; j < 3; ++j) frag(w2, F, wn * 96 + j * 32, c * kFC + 16 * t2);
        };
        for (int t = 0; t < kTP; ++t)
            for (int j = 0; j < 3; ++j) frag(wo, kHid, wn * 96 + j * 32, 16 * t);
        g1(0);
        for (int c = 0; c < nch; ++c) {
            if (c >= 1) g2(c - 1);
            if (c + 1 < nch) g1(c + 1);
        }
        g2(nch - 1);
    }
}

hipError_t tail_setup() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tail_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&tail_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
}

bool tail_supported(int hidden, int ffn) { return hidden == kHid && ffn >= 2 * kFC && ffn % kFC == 0 && ffn <= kMaxF; }

// p.ctx != nullptr: the whole tail (out-projection + LN1 + MLP + LN2), p.x = layer input;
// p.ctx == nullptr: the MLP block alone, p.x = x1
hipError_t launch_tail(hipStream_t s, const TailParams &p) {
    if (p.m % kBM || p.f % kFC || p.f < 2 * kFC || p.f > kMaxF || !p.wf) return hipErrorInvalidValue;
    if (p.ctx) hipLaunchKernelGGL(tail_kernel<true>, dim3(p.m / kBM), dim3(256), kLds, s, p);
    else hipLaunchKernelGGL(tail_kernel<false>, dim3(p.m / kBM), dim3(256), kLds, s, p);
    return hipGetLastError();
}

}  // namespace mx
