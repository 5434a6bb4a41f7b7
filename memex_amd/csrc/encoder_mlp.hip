// encoder_mlp.hip -- the transformer MLP block as ONE kernel (hidden = 384 models: MiniLM family):
//     out = LayerNorm(x + W2 * gelu_erf(W1 * x + b1) + b2)
// Part of the encoder forward that replaces rust-bert's `model.encode(&segments)` (reference
// lib/libmemex/src/llm/embedding.rs:109; BERT intermediate/output blocks, restated in
// oracle/bert_oracle.py).  Unfused this is two GEMMs with a [tokens, ffn] bf16 intermediate that is
// written to and read back from HBM (2 x 403 MB per layer at 131k tokens) -- both GEMMs are then
// bound by that traffic and by their epilogues, not by MFMA.  Here the intermediate never leaves the
// CU: a workgroup owns 128 token rows and walks the ffn dimension in chunks of 128 features:
//     G1  h[128 x 128] = x[128 x 384] * W1[chunk]^T          (12 k-tiles of 32, f32 accumulate)
//     E1  h = gelu(h + b1) -> bf16 tile in LDS (row pitch 272 B: conflict-free A-fragment reads)
//     G2  y[128 x 384] += h * W2[:, chunk]^T                  (4 k-tiles of 32; y stays in registers)
// and finishes with bias + residual + LayerNorm on the 128 x 384 tile.
//
// 512 threads = 8 waves as 2 (m) x 4 (n).  Wave tiles: G1 64 x 32 (2 MFMA tiles), G2 64 x 96 (6 tiles,
// 96 accumulator VGPRs).  Operands are staged by LDS-DMA (global_load_lds_dwordx4) as 64-byte rows
// with the same source-side chunk swizzle as gemm_kernel (encoder_kernels.hip): a G1 stage is 128 x
// rows + 128 W1 rows (16 KiB), a G2 stage is the 384 W2 rows of one k-block (24 KiB, contiguous in
// the K-blocked weight layout).  5-slot ring of 24 KiB + the h tile (34 KiB) = 154 KiB of LDS: one
// workgroup per CU.  MFMA: v_mfma_f32_32x32x16_bf16 with the weight fragment as A and the activation
// fragment as B (D^T: a lane owns one token row and 4 consecutive features -> 8-byte LDS stores).
#include "encoder_kernels.h"

// Ablation switch for scripts/mlp_ubench.hip only (0 = production kernel); bits:
//   1 = no DMA (compute on whatever LDS holds), 2 = E1 without the GELU arithmetic, 4 = no MFMA,
//   8 = no fragment reads, 16 = no final epilogue (E2), 32 = no E1; 64 / 128 / 256 = the x / W1 / W2
//   DMA pieces re-read one fixed KiB (cache-resident) instead of their stream
#ifndef MX_MLP_ABLATE
#define MX_MLP_ABLATE 0
#endif

namespace mx {

#ifdef MX_MLP_TRACE
// scripts/mlp_ubench.hip only: per-segment s_memtime totals of workgroup 0 / wave 0, kept in scalar
// registers during the kernel (a segment = the code between two MX_TRACE points, named by its END tag)
__device__ unsigned long long g_mlp_trace[32];
#define MX_TRACE(tag)                                                    \
    do {                                                                 \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
        tr_seg[tag] += now_ - tr_last;                                   \
        tr_cnt[tag] += 1;                                                \
        tr_last = now_;                                                  \
    } while (0)
#else
#define MX_TRACE(tag) do { } while (0)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

namespace {

constexpr int kHid = 384;
constexpr int kBM = 128;                       // token rows per workgroup
constexpr int kFC = 128;                       // ffn features per chunk
constexpr int kS = 5;                          // ring slots
constexpr int kSlot = kHid * 64;               // 24 KiB: the larger (G2) stage
constexpr int kHtPitch = kFC * 2 + 16;         // 272 B
constexpr int kHtOff = kS * kSlot;             // 122880
constexpr int kB1Off = kHtOff + kBM * kHtPitch;   // 157696: b1 in LDS (a global load in E1 would queue behind the DMA stream)
constexpr int kMaxF = 1536;
constexpr int kMlpLds = kB1Off + kMaxF * 4;       // 163840 = all 160 KiB
constexpr int kOutPitch = kHid * 2 + 16;       // 784 B: final bf16 tile, row-major
static_assert(kBM * kOutPitch <= kMlpLds, "final tile must fit the ring + h tile");
constexpr int kG1 = kHid / 32;                 // 12 G1 stages per chunk
constexpr int kG2 = kFC / 32;                  // 4 G2 stages per chunk
constexpr int kSPC = kG1 + kG2;                // 16 stages per chunk

__device__ __forceinline__ float gelu_erf_mlp(float x) {  // same arithmetic as gemm_kernel's epilogue
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    const float erfv = x < 0.0f ? -e : e;
    return 0.5f * x * (1.0f + erfv);
}

constexpr int stage_ops(int i) { return (i % kSPC) < kG1 ? 2 : 3; }  // DMA ops per wave of stage-in-chunk i

// n is a compile-time constant after unrolling: exactly one of the three waits survives
__device__ __forceinline__ void wait_vm(int n) {
    if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

}  // namespace

__global__ __launch_bounds__(512) void mlp_kernel(const MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * kBM;
    const int F = p.f;
    const int nch = F / kFC;
    const int total = nch * kSPC;  // stages
#ifdef MX_MLP_TRACE
    unsigned long long tr_seg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_last = __builtin_amdgcn_s_memtime();
    unsigned int tr_cnt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // ---- LDS-DMA sources.  A piece is 16 stage rows x 64 B; lane l -> row 16P + (l>>2), physical 16-B
    // chunk l&3, which holds logical chunk (l&3) ^ ((row>>2)&3).  All piece bases are multiples of 16
    // rows, so the swizzle term only depends on the lane.
    const int rowp = lane >> 2;
    const int csw = (lane & 3) ^ ((rowp >> 2) & 3);
    // G1: wave w moves x rows 16w.. and W1 rows 16w.. of the chunk
    const char *src_x = reinterpret_cast<const char *>(p.x + (size_t)(m0 + 16 * wave + rowp) * p.ldx + csw * 8);
    const char *src_w1 = reinterpret_cast<const char *>(p.w1 + (size_t)(16 * wave + rowp) * 32 + csw * 8);
    // G2: wave w moves W2 rows 16(w + 8j).., j = 0..2, of the k-block
    const char *src_w2 = reinterpret_cast<const char *>(p.w2 + (size_t)(16 * wave + rowp) * 32 + csw * 8);
    const size_t w1_kstep = (size_t)F * 64;      // bytes between k-tiles of W1 ([H/32][F][32])
    const size_t w2_block = (size_t)kHid * 64;   // bytes per k-block of W2 ([F/32][384][32])

    auto issue = [&](int chunk, int i, uint32_t slot) __attribute__((always_inline)) {  // i: compile-time at call sites
#if MX_MLP_ABLATE & 1
        return;
#endif
        if (i < kG1) {
            __builtin_amdgcn_global_load_lds((gbl_void_t *)((MX_MLP_ABLATE & 64) ? src_x : src_x + (size_t)i * 64),
                                             (lds_void_t *)(smem + __builtin_amdgcn_readfirstlane(slot + wave * 1024)), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void_t *)((MX_MLP_ABLATE & 128) ? src_w1 : src_w1 + (size_t)i * w1_kstep + (size_t)chunk * (kFC * 64)),
                                             (lds_void_t *)(smem + __builtin_amdgcn_readfirstlane(slot + (8 + wave) * 1024)), 16, 0, 0);
        } else {
            const char *b = (MX_MLP_ABLATE & 256) ? src_w2 : src_w2 + (size_t)(chunk * kG2 + (i - kG1)) * w2_block;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_amdgcn_global_load_lds((gbl_void_t *)(b + (size_t)j * (8 * 1024)),
                                                 (lds_void_t *)(smem + __builtin_amdgcn_readfirstlane(slot + (wave + 8 * j) * 1024)), 16, 0, 0);
        }
    };

    // ---- fragment addressing (64-B stage rows, swizzled chunks; h tile rows of 272 B)
    const uint32_t sw0 = (uint32_t)((h ^ ((l31 >> 2) & 3)) << 4), sw1 = sw0 ^ 32u;
    const uint32_t g1_a = (uint32_t)(wm * 64 + l31) * 64;          // x rows (B operand)
    const uint32_t g1_w = (uint32_t)(kBM + wn * 32 + l31) * 64;    // W1 rows (A operand)
    const uint32_t g2_a = (uint32_t)kHtOff + (uint32_t)(wm * 64 + l31) * kHtPitch + (uint32_t)h * 16;
    const uint32_t g2_w = (uint32_t)(wn * 96 + l31) * 64;          // W2 rows (A operand)

    f32x16 acc1[2];     // h chunk: [i: 32-row m block]
    f32x16 acc2[2][3];  // y: [i][j: 32-col n block]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.0f;
    }

    // ---- software pipeline.  Fragment registers are double-buffered (fb[2][6]):
    //   G1 (wave tile 64 x 32: only 2 MFMAs per k-step, less than one LDS latency) is pipelined a whole
    //   STAGE ahead: at the start of stage i the wave publishes stage i+1 (counted wait, barrier, DMA
    //   refill) and reads all 6 of its fragments between the 4 MFMAs of stage i;
    //   G2 (6 MFMAs per k-step) is pipelined one k-step ahead, the publish of the next stage sits in
    //   front of its second k-step.
    // The only drain is the G1 -> G2 hand-over (E1 needs the finished accumulators and G2's first
    // fragments come from the tile E1 writes).
    bf16x8 fb[2][6];  // G1: [3*ks + {W1, x rows 0-31, x rows 32-63}]; G2: [W2 x3, h rows x2] of one k-step
    uint32_t slot = 0;  // byte offset of the ring slot of the stage being multiplied
    auto next_slot = [](uint32_t sl) { return sl + kSlot == (uint32_t)(kS * kSlot) ? 0u : sl + kSlot; };
    auto prev_slot = [](uint32_t sl) { return sl == 0u ? (uint32_t)((kS - 1) * kSlot) : sl - kSlot; };
    auto read_g1 = [&](uint32_t sl, int buf, int which) __attribute__((always_inline)) {  // which: 0..5
#if MX_MLP_ABLATE & 8
        return;
#endif
        const uint32_t sw = which < 3 ? sw0 : sw1;
        const int part = which % 3;
        fb[buf][which] = *reinterpret_cast<const bf16x8 *>(smem + sl + (part == 0 ? g1_w : g1_a + (part - 1) * (32 * 64)) + sw);
    };
    auto read_g2 = [&](int i, int ks, uint32_t sl, int buf, int which) __attribute__((always_inline)) {  // which: 0..4
#if MX_MLP_ABLATE & 8
        return;
#endif
        const uint32_t sw = ks == 0 ? sw0 : sw1;
        const int kt2 = i - kG1;
        if (which < 3) fb[buf][which] = *reinterpret_cast<const bf16x8 *>(smem + sl + g2_w + which * (32 * 64) + sw);
        else fb[buf][which] = *reinterpret_cast<const bf16x8 *>(smem + g2_a + (which - 3) * (32 * kHtPitch) + (kt2 * 32 + ks * 16) * 2);
    };
    // publish stage g+1: counted wait, barrier, refill the slot of stage g-1 with stage g+4
    auto publish_next = [&](int chunk, int i, int g) __attribute__((always_inline)) {
        if (g + 1 >= total) return;
        MX_TRACE(1);
        // issued so far: stages <= g+3; stage g+1 has landed once at most stages g+2, g+3 are outstanding
        if (g + 3 < total) wait_vm(stage_ops(i + 2) + stage_ops(i + 3));
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MX_TRACE(2);
        __builtin_amdgcn_s_barrier();
        MX_TRACE(3);
        if (g + 4 < total) {
            const int i4 = (i + 4) % kSPC, c4 = chunk + (i + 4) / kSPC;
            issue(c4, i4, prev_slot(slot));
        }
        MX_TRACE(4);
    };

    // b1 -> LDS: loads first (older than the DMA prologue, so their wait does not drain it), stores after
    float b1v[kMaxF / 512];
#pragma unroll
    for (int c = 0; c < kMaxF / 512; ++c) b1v[c] = (c * 512 + tid < F) ? p.b1[c * 512 + tid] : 0.0f;
    // ---- prologue: stages 0 .. 3 in flight, stage 0's fragments in registers
#pragma unroll
    for (int i = 0; i < kS - 1; ++i) issue(0, i, (uint32_t)i * kSlot);
#pragma unroll
    for (int c = 0; c < kMaxF / 512; ++c) reinterpret_cast<float *>(smem + kB1Off)[c * 512 + tid] = b1v[c];
    wait_vm(stage_ops(1) + stage_ops(2) + stage_ops(3));
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int w = 0; w < 6; ++w) read_g1(0u, 0, w);

#pragma unroll 1
    for (int chunk = 0; chunk < nch; ++chunk) {
        const int g0 = chunk * kSPC;
#pragma unroll
        for (int i = 0; i < kSPC; ++i) {
            if (i < kG1) {
                // ---------------- G1 stage i: fragments in fb[i & 1] ----------------
                const int cur = i & 1, nxt = cur ^ 1;
                const bool last = i == kG1 - 1;  // next stage is G2's first: no prefetch (drain)
                uint32_t nslot = slot;
                if (!last) {
                    publish_next(chunk, i, g0 + i);
                    nslot = next_slot(slot);
                }
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    const int ks = mm >> 1, ii = mm & 1;
                    if (!last) {  // the 6 fragment reads of stage i+1, spread 2-2-1-1 in front of the 4 MFMAs
                        if (mm == 0) { read_g1(nslot, nxt, 0); read_g1(nslot, nxt, 1); }
                        if (mm == 1) { read_g1(nslot, nxt, 2); read_g1(nslot, nxt, 3); }
                        if (mm == 2) read_g1(nslot, nxt, 4);
                        if (mm == 3) read_g1(nslot, nxt, 5);
                    }
#if MX_MLP_ABLATE & 4
                    asm volatile("" ::"v"(fb[cur][3 * ks]), "v"(fb[cur][3 * ks + 1 + ii]));
#else
                    acc1[ii] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][3 * ks], fb[cur][3 * ks + 1 + ii], acc1[ii], 0, 0, 0);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
                slot = next_slot(slot);
                MX_TRACE(5);
                if (last) {
#if !(MX_MLP_ABLATE & 32)
                    // ---- E1: h = gelu(acc1 + b1) -> bf16 -> h tile.  Lane owns token row (l31) of each
                    // 32-row block and features 8*rg + 4*h + (0..3) of the wave's 32.
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int nloc = wn * 32 + 8 * rg + 4 * h;
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(smem + kB1Off + (chunk * kFC + nloc) * 4);
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            bf16x4 pk;
#pragma unroll
#if MX_MLP_ABLATE & 2
                            for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(acc1[ii][rg * 4 + e] + b4[e]);
#else
                            for (int e = 0; e < 4; ++e) pk[e] = (__bf16)gelu_erf_mlp(acc1[ii][rg * 4 + e] + b4[e]);
#endif
                            *reinterpret_cast<bf16x4 *>(smem + kHtOff + (wm * 64 + ii * 32 + l31) * kHtPitch + nloc * 2) = pk;
                        }
                    }
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc1[ii][r] = 0.0f;
#endif
                    MX_TRACE(6);
                    // publish stage kG1 (G2's first) and the h tile: all h-tile writes done (lgkmcnt),
                    // then the same wait/barrier/refill as everywhere else; `slot` already points at it
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    {
                        const int g = g0 + i;
                        if (g + 3 < total) wait_vm(stage_ops(i + 2) + stage_ops(i + 3));
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        if (g + 4 < total) issue(chunk + (i + 4) / kSPC, (i + 4) % kSPC, prev_slot(prev_slot(slot)));
                    }
#pragma unroll
                    for (int w = 0; w < 5; ++w) read_g2(kG1, 0, slot, 0, w);
                    MX_TRACE(7);
                }
            } else {
                // ---------------- G2 stage i: k-step ks in fb[ks] ----------------
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int cur = ks, nxt = ks ^ 1;
                    uint32_t nslot = slot;
                    if (ks == 1) {
                        publish_next(chunk, i, g0 + i);
                        nslot = next_slot(slot);
                    }
                    const bool to_g1 = ks == 1 && i == kSPC - 1;  // next: stage 0 of the next chunk (6 fragments)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const int q = ii * 3 + j;
                            if (to_g1) read_g1(nslot, nxt, q);
                            else if (q < 5) read_g2(ks == 0 ? i : i + 1, ks ^ 1, nslot, nxt, q);
#if MX_MLP_ABLATE & 4
                            asm volatile("" ::"v"(fb[cur][j]), "v"(fb[cur][3 + ii]));
#else
                            acc2[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][j], fb[cur][3 + ii], acc2[ii][j], 0, 0, 0);
#endif
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    if (ks == 1) slot = next_slot(slot);
                    MX_TRACE(8 + ks);
                }
            }
        }
    }
#ifdef MX_MLP_TRACE
    if (blockIdx.x == 0 && tid == 0)
        for (int t = 0; t < 10; ++t) {
            g_mlp_trace[t] = tr_seg[t];
            g_mlp_trace[16 + t] = tr_cnt[t];
        }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // ring and h tile are dead: their space becomes the 128 x 384 output tile

#if MX_MLP_ABLATE & 16
    if (p.eps > 1e30f)
#endif
    {
    // ---- E2 pass 1: y + b2 -> bf16 tile [m][n]
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int nloc = wn * 96 + j * 32 + 8 * rg + 4 * h;
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(p.b2 + nloc);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(acc2[ii][j][rg * 4 + e] + b4[e]);
                *reinterpret_cast<bf16x4 *>(smem + (wm * 64 + ii * 32 + l31) * kOutPitch + nloc * 2) = pk;
            }
        }
    __syncthreads();
    // ---- E2 pass 2: + residual, LayerNorm over the 384 features of a row, coalesced 16-byte stores.
    // 4 threads per row, 12 chunks of 8 features each, interleaved chunk-wise.
    {
        constexpr int TPR = 512 / kBM, CPT = kHid / TPR / 8;
        const int row = tid / TPR, prt = tid % TPR;
        float y[CPT * 8];
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int col = (c * TPR + prt) * 8;
            const bf16x8 o = *reinterpret_cast<const bf16x8 *>(smem + row * kOutPitch + col * 2);
            const bf16x8 rs = *reinterpret_cast<const bf16x8 *>(p.x + (size_t)(m0 + row) * p.ldx + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                y[c * 8 + e] = (float)o[e] + (float)rs[e];
                sum += y[c * 8 + e];
            }
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)kHid;
        float sq = 0.0f;
#pragma unroll
        for (int e = 0; e < CPT * 8; ++e) {
            const float dlt = y[e] - mean;
            sq += dlt * dlt;
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) sq += __shfl_xor(sq, o);
        const float rstd = 1.0f / sqrtf(sq / (float)kHid + p.eps);
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int col = (c * TPR + prt) * 8;
            const f32x4 g0v = *reinterpret_cast<const f32x4 *>(p.gamma + col);
            const f32x4 g1v = *reinterpret_cast<const f32x4 *>(p.gamma + col + 4);
            const f32x4 b0v = *reinterpret_cast<const f32x4 *>(p.beta + col);
            const f32x4 b1v = *reinterpret_cast<const f32x4 *>(p.beta + col + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (__bf16)((y[c * 8 + e] - mean) * rstd * g0v[e] + b0v[e]);
                o[4 + e] = (__bf16)((y[c * 8 + 4 + e] - mean) * rstd * g1v[e] + b1v[e]);
            }
            *reinterpret_cast<bf16x8 *>(p.out + (size_t)(m0 + row) * p.ldo + col) = o;
        }
    }
    }
}

hipError_t mlp_setup() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               kMlpLds);
}

bool mlp_supported(int hidden, int ffn) { return hidden == kHid && ffn >= kFC && ffn % kFC == 0 && ffn <= kMaxF; }

hipError_t launch_mlp(hipStream_t s, const MlpParams &p) {
    if (p.m % kBM || p.f % kFC || p.f < kFC || p.f > kMaxF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mlp_kernel, dim3(p.m / kBM), dim3(512), kMlpLds, s, p);
    return hipGetLastError();
}

}  // namespace mx
