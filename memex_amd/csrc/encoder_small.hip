// encoder_small.hip -- the small-pass layer of the hidden-384 encoder: query-time embedding.
//
// `encode_single` (reference lib/libmemex/src/llm/embedding.rs:146-151, called per search request at
// lib/api/src/endpoints/collections/handlers.rs:61-81) embeds ONE short text.  The kernels built for ingest passes of
// 100k tokens are wrong-sized for it: a 16-token query fills one 64-row tile, and tail_kernel then streams the layer's
// 2.65 MB of Wo / W1 / W2 through ONE compute unit (48 us of a 71 us layer, 6-12 layers); the two projection GEMMs take
// 8.8 us each for a dozen barrier-separated k-tiles.  Passes of <= kSmallRows packed rows therefore run
//     sp_qkv_kernel        q | k | v^T = x Wqkv^T: one WAVE per 64 rows x 32 output features, weight fragments straight from
//                          L2 in the GEMMs' K-blocked layout (24 loads issued up front, no barrier in the k loop)
//     attention_kernel     unchanged
//     sp_out_ln_kernel     out-projection + Add&Norm: twelve waves per row tile, a wave per 32 features
//     sp_ffn_kernel        the MLP split over the ffn dimension: workgroup (row tile, chunk c of 128 features) computes
//                          h_c = gelu(x1 W1_c^T + b1_c) and the PARTIAL product h_c W2[:, c]^T (64 x 384, f32) -- the layer's
//                          weights stream through ffn/128 = 12 compute units instead of one
//     sp_reduce_ln_kernel  sum of the partials (fixed order) + b2 -> bf16 -> + x1 -> LayerNorm2
// Same operands, MFMA shape, k order within a product and rounding points as the large-pass kernels; what differs is the
// order in which the MLP's 12 chunk products are summed (f32), so a row's embedding agrees with the large-pass result to
// 1 - cos ~ 1e-7, not bit for bit (tests/test_encoder_gpu.py::test_small_pass_matches_large_pass).
#include "encoder_kernels.h"
#include "mx_gelu.h"
#include "mx_layernorm.h"

namespace mx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

namespace {

constexpr int kHid = 384;
constexpr int kBM = 64;                    // token rows per workgroup
constexpr int kFC = 128;                   // ffn features per chunk (tail_kernel's chunk)
constexpr int kXBytes = kBM * kHid * 2;    // 48 KiB: x tile, row pitch 768 B, 16-byte chunk c of row r at c ^ (r & 15)
constexpr int kHBytes = kBM * kFC * 2;     // 16 KiB: h tile, row pitch 256 B, same swizzle
constexpr int kT1 = kHid / 16;             // 24 k-steps over the hidden dimension
constexpr int kT2 = kFC / 16;              // 8 k-steps over a chunk
constexpr int kTP = kHid / 16;             // fragments x 3 of the out-projection segment of tail_kernel's weight stream
constexpr int kFPC = kT1 + 3 * kT2;        // 48 fragments per wave and chunk in that stream
constexpr int kMaxCh = 12;                 // ffn <= 1536 (tail_supported)

// the x tile (64 rows x 384 bf16 of `src`, row pitch ld elements) -> LDS offset 0 by LDS-DMA: 48 KiB-operations dealt to
// NW waves
template <int NW = 4>
__device__ __forceinline__ void stage_x_tile(char *smem, const bf16_t *src, int ld, int m0, int wn, int lane) {
    constexpr int OPS = 48 / NW;
#pragma unroll
    for (int o = 0; o < OPS; ++o) {
        const int P = (wn * OPS + o) * 64 + lane;  // physical chunk position in the tile
        const int r = P / 48, pc = P % 48;
        const int c = pc ^ (r & 15);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(src + (size_t)(m0 + r) * ld + c * 8),
                                         (lds_void_t *)(smem + __builtin_amdgcn_readfirstlane((wn * OPS + o) * 1024)), 16, 0, 0);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// q | k | v^T = x Wqkv^T + b: grid (row tiles, 3H / 128), a wave per 32 output features
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_qkv_kernel(const bf16_t *__restrict__ x, const bf16_t *__restrict__ w, const float *__restrict__ bias,
                                                      int w_rows, float qscale, bf16_t *__restrict__ q, bf16_t *__restrict__ k,
                                                      bf16_t *__restrict__ vt, int ldvt, int rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * kBM;
    const int n0 = blockIdx.y * 128 + wn * 32;  // this wave's output features
    // rows >= `rows` of the pass are padding (a 16-token query fills a quarter of its tile): the tile's second 32-row block
    // is multiplied only when it holds tokens.  What the skipped rows keep is never read: attention masks by length, pooling
    // reads tokens, the LayerNorms are row-wise.
    const bool two = m0 + 32 < rows;
    // weight fragments of all 24 k-steps: K-blocked [K/32][w_rows][32]: element (n, k) at ((k >> 5) * w_rows + n) * 32 + (k & 31)
    bf16x8 wf[kT1];
#pragma unroll
    for (int t = 0; t < kT1; ++t)
        wf[t] = *reinterpret_cast<const bf16x8 *>(w + ((size_t)(t >> 1) * w_rows + n0 + l31) * 32 + (t & 1) * 16 + 8 * h);
    stage_x_tile(smem, x, kHid, m0, wn, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): fragments and this wave's part of the tile
    __builtin_amdgcn_s_barrier();
    const uint32_t x_row = (uint32_t)l31 * (kHid * 2);
    const uint32_t shs = ((uint32_t)(l31 & 15) ^ (uint32_t)h) << 4;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
    for (int t = 0; t < kT1; ++t) {
        const char *a = smem + x_row + (shs ^ (uint32_t)((t & 7) << 5)) + (t >> 3) * 256;
        const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(a);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(a + 32 * kHid * 2);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t], b0, acc[0], 0, 0, 0);
        if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t], b1, acc[1], 0, 0, 0);
    }
    // D^T: lane owns token row l31 (+ 32 ii) and features n0 + 8 rg + 4 h + (0..3) per register group
    const int part = n0 / kHid;  // 0 = q, 1 = k, 2 = v (wave-uniform)
    const float oscale = part == 0 ? qscale : 1.0f;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int nloc = n0 + 8 * rg + 4 * h;
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + nloc);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            if (ii == 1 && !two) break;
            const int row = m0 + ii * 32 + l31;
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (__bf16)((acc[ii][rg * 4 + e] + b4[e]) * oscale);
            if (part < 2) {
                *reinterpret_cast<bf16x4 *>((part == 0 ? q : k) + (size_t)row * kHid + (nloc - part * kHid)) = pk;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) vt[(size_t)(nloc - 2 * kHid + e) * ldvt + row] = pk[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// x1 = LayerNorm1(x + ctx Wo^T + bo): one workgroup of TWELVE waves per row tile, a wave per 32 output features with all 24
// weight fragments in flight at once (tail_kernel's 4 waves x 96 features walk 72 fragments each through a 12-deep ring: six
// round trips to the Infinity Cache, 12 us for a tile), then tail_kernel's Add & LayerNorm (same staging tile and helpers;
// 8 threads per row instead of 4: twelve waves share one register file).
// ---------------------------------------------------------------------------------------------
constexpr int kOutPitch = kHid * 2 + 16;  // bf16 staging tile, row-major (tail_kernel's)

__global__ __launch_bounds__(768) void sp_out_ln_kernel(const bf16_t *__restrict__ ctx, const bf16_t *__restrict__ xres, const bf16_t *__restrict__ wo,
                                                         const float *__restrict__ bo, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float eps, bf16_t *__restrict__ x1, int rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0 .. 11
    const bool two = (int)blockIdx.x * kBM + 32 < rows;       // (see sp_qkv_kernel)
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * kBM;
    const int n0 = wn * 32;
    bf16x8 wf[kT1];
#pragma unroll
    for (int t = 0; t < kT1; ++t)
        wf[t] = *reinterpret_cast<const bf16x8 *>(wo + ((size_t)(t >> 1) * kHid + n0 + l31) * 32 + (t & 1) * 16 + 8 * h);
    stage_x_tile<12>(smem, ctx, kHid, m0, wn, lane);
    // the LayerNorm threads: 8 per row, 6 chunks of 8 features each, interleaved chunk-wise
    constexpr int TPR = 8, CPT = kHid / TPR / 8;
    const int ln_row = tid / TPR, ln_prt = tid % TPR;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    const uint32_t x_row = (uint32_t)l31 * (kHid * 2);
    const uint32_t shs = ((uint32_t)(l31 & 15) ^ (uint32_t)h) << 4;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
    for (int t = 0; t < kT1; ++t) {
        const char *a = smem + x_row + (shs ^ (uint32_t)((t & 7) << 5)) + (t >> 3) * 256;
        const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(a);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(a + 32 * kHid * 2);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t], b0, acc[0], 0, 0, 0);
        if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t], b1, acc[1], 0, 0, 0);
    }
    // (the residual loads start here, in the registers the fragments have left: 12 waves share a register file of 512)
    bf16x8 rs[CPT];
    if (tid < kBM * TPR) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) rs[c] = *reinterpret_cast<const bf16x8 *>(xres + (size_t)(m0 + ln_row) * kHid + (c * TPR + ln_prt) * 8);
    }
    __builtin_amdgcn_s_barrier();  // every wave is done with the ctx tile (its LDS reads fed MFMAs already issued): staging tile
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int nloc = n0 + 8 * rg + 4 * h;
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bo + nloc);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(acc[ii][rg * 4 + e] + b4[e]);
            *reinterpret_cast<bf16x4 *>(smem + (ii * 32 + l31) * kOutPitch + nloc * 2) = pk;
        }
    }
    __syncthreads();
    if (tid >= kBM * TPR) return;
    float y[CPT * 8];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int col = (c * TPR + ln_prt) * 8;
        const bf16x8 o = *reinterpret_cast<const bf16x8 *>(smem + ln_row * kOutPitch + col * 2);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[c * 8 + e] = (float)o[e] + (float)rs[c][e];
    }
    float mean, rstd;
    ln_row_stats<TPR, CPT * 8>(y, eps, mean, rstd);
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
        *reinterpret_cast<bf16x8 *>(x1 + (size_t)(m0 + ln_row) * kHid + col) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// MLP chunk: part[c] = gelu(x1 W1_c^T + b1_c) W2[:, c]^T, grid (row tiles, ffn / 128).  Weight fragments come from
// tail_kernel's per-wave streams (encoder_tail.hip::tail_stream_layout): wave wn's G1(c) segment = 24 fragments of W1 rows
// c*128 + wn*32 .. +31, its G2(c) segment = 8 k-steps x 3 row groups of W2 rows wn*96 + j*32 .. +31, k = c*128 + 16 t2.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_ffn_kernel(const bf16_t *__restrict__ x1, const bf16_t *__restrict__ wfs, const float *__restrict__ b1,
                                                      int nch, int m_pad, float *__restrict__ part, int rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * kBM, c = blockIdx.y;
    const bool two = m0 + 32 < rows;  // (see sp_qkv_kernel)
    // fragment positions in the wave's stream: PO (72) | G1(0) | G1(1) | G2(0) G1(2) | ... | G2(nch-2) | G2(nch-1)
    const int f_g1 = c == 0 ? 3 * kTP : c == 1 ? 3 * kTP + kT1 : 3 * kTP + 2 * kT1 + kFPC * (c - 2) + 3 * kT2;
    const int f_g2 = c + 1 < nch ? 3 * kTP + 2 * kT1 + kFPC * c : 3 * kTP + 2 * kT1 + kFPC * (nch - 2) + 3 * kT2;
    const char *ws = reinterpret_cast<const char *>(wfs) + (size_t)wn * (3 * kTP + nch * kFPC) * 1024 + lane * 16;
    // all 48 fragments of the chunk are issued before anything else: the kernel is one latency chain (weights from the
    // Infinity Cache: six layers' worth does not stay in a 4 MiB L2), so the second segment must not wait for the first's use
    bf16x8 g1[kT1], ring[kT1];
#pragma unroll
    for (int t = 0; t < kT1; ++t) g1[t] = *reinterpret_cast<const bf16x8 *>(ws + (size_t)(f_g1 + t) * 1024);
#pragma unroll
    for (int t = 0; t < kT1; ++t) ring[t] = *reinterpret_cast<const bf16x8 *>(ws + (size_t)(f_g2 + t) * 1024);
    const float b1v = b1[c * kFC + wn * 32 + l31];  // lane l holds the bias of feature l & 31 of the wave's 32
    stage_x_tile(smem, x1, kHid, m0, wn, lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the tile's DMA is the youngest operation, everything has landed
    __builtin_amdgcn_s_barrier();
    const uint32_t s = (uint32_t)(l31 & 15);
    const uint32_t shs = (s ^ (uint32_t)h) << 4;
    const uint32_t x_row = (uint32_t)l31 * (kHid * 2);
    const uint32_t h_row = (uint32_t)kXBytes + (uint32_t)l31 * (kFC * 2);
    f32x16 acc1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][r] = 0.0f;
#pragma unroll
    for (int t = 0; t < kT1; ++t) {
        const char *a = smem + x_row + (shs ^ (uint32_t)((t & 7) << 5)) + (t >> 3) * 256;
        const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(a);
        const bf16x8 bq = *reinterpret_cast<const bf16x8 *>(a + 32 * kHid * 2);
        acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[t], b0, acc1[0], 0, 0, 0);
        if (two) acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g1[t], bq, acc1[1], 0, 0, 0);
    }
    // E1: h = gelu(acc1 + b1) -> bf16 -> h tile (lane: token row l31 (+ 32 ii), features 8 rg + 4 h + (0..3) of the wave's 32)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        float b4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            b4[e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((8 * rg + 4 * h + e) * 4, __builtin_bit_cast(int, b1v)));
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const gelu_f32x2 g = gelu_erf2(gelu_f32x2{acc1[ii][rg * 4 + e] + b4[e], acc1[ii][rg * 4 + e + 1] + b4[e + 1]});
                pk[e] = (__bf16)g[0];
                pk[e + 1] = (__bf16)g[1];
            }
            const uint32_t pc = (((uint32_t)(wn * 4 + rg)) ^ s) << 4;
            *reinterpret_cast<bf16x4 *>(smem + h_row + ii * (32 * kFC * 2) + pc + h * 8) = pk;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();        // h tile complete
    f32x16 acc2[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.0f;
#pragma unroll
    for (int t2 = 0; t2 < kT2; ++t2) {
        const char *a = smem + h_row + (shs ^ (uint32_t)(t2 << 5));
        const bf16x8 b0 = *reinterpret_cast<const bf16x8 *>(a);
        const bf16x8 bq = *reinterpret_cast<const bf16x8 *>(a + 32 * kFC * 2);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            acc2[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[3 * t2 + j], b0, acc2[0][j], 0, 0, 0);
            if (two) acc2[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[3 * t2 + j], bq, acc2[1][j], 0, 0, 0);
        }
    }
    float *po = part + ((size_t)c * m_pad + m0) * kHid;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int nloc = wn * 96 + j * 32 + 8 * rg + 4 * h;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc2[ii][j][rg * 4 + e];
                *reinterpret_cast<f32x4 *>(po + (size_t)(ii * 32 + l31) * kHid + nloc) = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// out[r] = LayerNorm2(bf16(sum_c part[c][r] + b2) + x1[r]): 16 lanes per row, 24 values per lane
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_reduce_ln_kernel(const float *__restrict__ part, int nch, int m_pad, const float *__restrict__ b2,
                                                            const bf16_t *__restrict__ x1, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps, int rows, bf16_t *__restrict__ out) {
    constexpr int TPR = 16;
    const int l = threadIdx.x % TPR;
    const int row = (int)(blockIdx.x * (256 / TPR) + threadIdx.x / TPR);
    if (row >= rows) return;
    float y[24];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        const int col = (cc * TPR + l) * 8;
        // (all chunks' loads in flight together -- a dependent loop pays the L2 latency nch times -- then summed in chunk order)
        f32x4 pa[kMaxCh], pb[kMaxCh];
#pragma unroll
        for (int c = 0; c < kMaxCh; ++c) {
            const float *p = part + ((size_t)(c < nch ? c : 0) * m_pad + row) * kHid + col;
            pa[c] = *reinterpret_cast<const f32x4 *>(p);
            pb[c] = *reinterpret_cast<const f32x4 *>(p + 4);
        }
        f32x4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = s0;
#pragma unroll
        for (int c = 0; c < kMaxCh; ++c)
            if (c < nch) {
                s0 += pa[c];
                s1 += pb[c];
            }
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(b2 + col);
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(b2 + col + 4);
        const bf16x8 rs = *reinterpret_cast<const bf16x8 *>(x1 + (size_t)row * kHid + col);
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
        *reinterpret_cast<bf16x8 *>(out + (size_t)row * kHid + col) = o;
    }
}

hipError_t small_setup() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sp_qkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kXBytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sp_out_ln_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kBM * kOutPitch);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&sp_ffn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kXBytes + kHBytes);
}

hipError_t launch_sp_out_ln(hipStream_t s, const bf16_t *ctx, const bf16_t *xres, const bf16_t *wo, const float *bo, const float *gamma,
                            const float *beta, float eps, int m, int rows, bf16_t *x1) {
    if (m % kBM || rows < 1 || rows > m) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sp_out_ln_kernel, dim3((rows + kBM - 1) / kBM), dim3(768), kBM * kOutPitch, s, ctx, xres, wo, bo, gamma, beta, eps, x1, rows);
    return hipGetLastError();
}

hipError_t launch_sp_qkv(hipStream_t s, const bf16_t *x, const bf16_t *wqkv, const float *bqkv, int m, int rows, float qscale, bf16_t *q,
                         bf16_t *k, bf16_t *vt, int ldvt) {
    if (m % kBM || rows < 1 || rows > m) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sp_qkv_kernel, dim3((rows + kBM - 1) / kBM, 3 * kHid / 128), dim3(256), kXBytes, s, x, wqkv, bqkv, 3 * kHid, qscale, q, k, vt,
                       ldvt, rows);
    return hipGetLastError();
}

hipError_t launch_sp_ffn(hipStream_t s, const bf16_t *x1, const bf16_t *wf, const float *b1, int f, int m, int rows, float *part) {
    if (m % kBM || f % kFC || f < 2 * kFC || rows < 1 || rows > m) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sp_ffn_kernel, dim3((rows + kBM - 1) / kBM, f / kFC), dim3(256), kXBytes + kHBytes, s, x1, wf, b1, f / kFC, m, part, rows);
    return hipGetLastError();
}

hipError_t launch_sp_reduce_ln(hipStream_t s, const float *part, int f, int m, int rows, const float *b2, const bf16_t *x1, const float *gamma,
                               const float *beta, float eps, bf16_t *out) {
    hipLaunchKernelGGL(sp_reduce_ln_kernel, dim3((rows + 15) / 16), dim3(256), 0, s, part, f / kFC, m, b2, x1, gamma, beta, eps, rows, out);
    return hipGetLastError();
}

}  // namespace mx
