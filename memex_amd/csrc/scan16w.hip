// scan16w.hip -- the streaming cosine scan over the bf16 filter copy for WIDE rows (768 < dim_pad <= 1536:
// bge-large / e5-large 1024-d, 1536-d API embeddings).
//
// scan16_kernel keeps the B fragments of 32 queries x dim_pad in the VGPRs of every wave: 192 of 256
// registers at 768 dims, nothing left above that.  Here the k-steps of a row are dealt to TWO waves: the
// 8 waves of the workgroup are 4 query groups (32 queries each, 128 queries per launch) x 2 slot parities.
// A row tile (32 rows x dim_pad) is KC 8-KiB slots of 128 dims, KC even (index.hip pads wide rows to a
// multiple of 256 dims): waves 0-3 multiply the ODD slots of every tile, waves 4-7 the EVEN ones, each
// against the 32 queries of its group (qfrag layout unchanged: the wave loads the k-steps of its parity).
// Because KC is even a wave's slots are j, j+2, j+4, ... across tile boundaries, so the fragment ring of
// scan16_kernel carries over with "two slots ahead" in place of "the next slot":
//   * DMA stream, ring of 16 slots with 15 in flight, one 1 KiB LDS-DMA per wave and slot -- unchanged;
//   * s_waitcnt vmcnt(12) + s_barrier per slot: slots <= j+2 have landed, slot j-1 is free for slot j+15;
//   * the wave whose parity slot j has: 8 MFMAs, each followed by the ds_read_b128 that refills the
//     fragment register it consumed (from slot j or slot j+2); the other wave only issues its DMA.
// Tile end: waves 4-7 finish their last slot (KC-2) one step before waves 0-3 finish theirs (KC-1); they
// park their 16 partial sums per lane in LDS (4 x ds_write_b128, lane-linear) before the barrier of
// step KC-1, and waves 0-3 add them after their own last MFMA and run the tile epilogue of scan16_kernel
// (same records, same lane positions: theta_kernel / finish_kernel do not know which kernel ran).
// MFMA density is half of scan16_kernel's (one of the two waves of a SIMD multiplies at any time), which
// at 32 B/clk/CU of stream still outruns HBM: the launch is bound by the DMA stream.  256 queries are two
// launches (index.hip splits the batch).
#include <type_traits>

#include "index_kernels.h"

namespace mx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

#define MX_LDS_DMA16(rsrc, ldsptr, voff, soff, aux) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lds_void *)(ldsptr), 16, (voff), (soff), 0, (aux))

static_assert(kRing16 == 16, "waits below assume a 16-slot ring with 15 slots in flight");

template <int KC, int MODE>
__global__ __launch_bounds__(kScanThreads, 2) void scan16w_kernel(const ScanParams p) {
    static_assert(KC % 2 == 0 && KC >= 4, "slot parities need an even slot count per tile");
    constexpr int KL = KC / 2;           // slots per tile this wave multiplies
    constexpr int R = KL <= 4 ? 8 : KL == 5 ? 4 : 2;  // fragment ring: what the 256 VGPRs leave next to qf (KL = 6: 192 of them)
    constexpr bool DUAL = KL <= 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // smem: the slot ring | [4 groups][4][64 lanes] x 16 B partial sums

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave & 3;          // query group: queries 32*grp .. +31
    const int par = wave < 4 ? 1 : 0;  // slot parity this wave multiplies (waves 0-3: odd slots + the epilogue)
    const int m = lane & 31;

    // ---- register-resident query fragments of this wave's slots: local chunk c = slot 2c + par of a tile
    bf16x8 qf[KL * 8];
    {
        const bf16x8 *src = reinterpret_cast<const bf16x8 *>(p.qfrag) + (size_t)grp * (KC * 8) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KL * 8; ++i) qf[i] = src[(size_t)(((i >> 3) * 2 + par) * 8 + (i & 7)) * 64];
    }
    const float theta = MODE == 1 ? p.theta[grp * 32 + m] : 0.0f;

    const uint32_t grid = gridDim.x;
    const uint32_t stride = p.tile_stride;
    const uint32_t t0 = p.tile_begin + blockIdx.x * stride;
    const uint32_t tstep = grid * stride;
    const uint32_t nT = (t0 < p.tile_end) ? (p.tile_end - t0 + tstep - 1) / tstep : 0;
    const uint32_t tilebytes = p.ds * (kTileRows * 2);

    const uint32_t lane16 = (uint32_t)lane * 16u;

    // ---- DMA stream: exactly as scan16_kernel (dead past the last tile: num_records = 0)
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t is_ti = 0;
    auto open_tile = [&]() {
        const char *base = reinterpret_cast<const char *>(p.xh) + (size_t)(t0 + is_ti * tstep) * tilebytes;
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, is_ti < nT ? tilebytes : 0u, 0x00020000);
        ++is_ti;
    };
    auto issue = [&](int kci, uint32_t ring_pos) {
        if (kci == 0) open_tile();
        char *dst = smem + __builtin_amdgcn_readfirstlane(ring_pos * kSlot16Bytes + wave * 1024);
        MX_LDS_DMA16(rsrc, dst, lane16, kci * kSlot16Bytes + wave * 1024, 2 /* nt */);  // wave's KiB of the slot: scalar offset
    };

    // records go where scan16_kernel's wave `grp` would put them (only waves 0-3 write any); the addresses
    // are rebuilt from this one word where a record is stored (a few percent of the tiles): registers are
    // what this kernel is short of
    auto mylane = [&]() { return (uint32_t)(grp * 64 + (lane16 >> 4)) * gridDim.x + blockIdx.x; };
    uint32_t cnt = 0;  // records written; bit 31: a record did not fit
    float best = -INFINITY;

#pragma unroll
    for (int i = 0; i < kRing16 - 1; ++i) issue(i % KC, (uint32_t)i);

    bf16x8 a[R];
    asm volatile("s_waitcnt vmcnt(13)" ::: "memory");  // slots 0 and 1 landed
    __builtin_amdgcn_s_barrier();
    {
        const uint32_t fb = (uint32_t)par * kSlot16Bytes + lane16;
#pragma unroll
        for (int ks = 0; ks < R; ++ks) a[ks] = *reinterpret_cast<const bf16x8 *>(smem + fb + ks * 1024);
    }
    const uint32_t xoff = __builtin_amdgcn_readfirstlane(kRing16 * kSlot16Bytes + grp * 4096);  // + lane16 + i*1024: this lane's parked sums

    // one tile; PAR is the compile-time copy of `par`
    uint32_t rp = 0;
    auto tile = [&](auto par_tag, uint32_t ti) {
        constexpr int PAR = decltype(par_tag)::value;
        f32x16 acc, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f, acc1[r] = 0.0f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const uint32_t rp1 = (rp + 1) & (kRing16 - 1);
            const uint32_t rp2 = (rp + 2) & (kRing16 - 1);
            const uint32_t rpi = (rp + kRing16 - 1) & (kRing16 - 1);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if ((kc & 1) == PAR) {
                const int c = kc >> 1;  // local chunk
                const uint32_t fb0 = rp * kSlot16Bytes + lane16, fb2 = rp2 * kSlot16Bytes + lane16;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (DUAL && (ks & 1))
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % R], qf[c * 8 + ks], acc1, 0, 0, 0);
                    else
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % R], qf[c * 8 + ks], acc, 0, 0, 0);
                    a[ks % R] = *reinterpret_cast<const bf16x8 *>(smem + (ks + R < 8 ? fb0 : fb2) + ((ks + R) & 7) * 1024);
                    if (ks == 1) issue((kc + kRing16 - 1) % KC, rpi);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (PAR == 0 && kc == KC - 2) {
                    // even-slot waves: park the tile's partial sums for the odd-slot wave of the group
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = DUAL ? acc[4 * i + j] + acc1[4 * i + j] : acc[4 * i + j];
                        *reinterpret_cast<f32x4 *>(smem + xoff + lane16 + i * 1024) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // in LDS before the barrier of step KC-1
                }
            } else {
                issue((kc + kRing16 - 1) % KC, rpi);
            }
            rp = rp1;
        }
        if constexpr (PAR == 1) {
        // ---- tile epilogue (odd-slot waves): own sums + the parked ones; then as scan16_kernel
        if (DUAL) acc += acc1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 o = *reinterpret_cast<const f32x4 *>(smem + xoff + lane16 + i * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * i + j] += o[j];
        }
        float mx = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, acc[r]), acc[r + 1]);
        mx = fmaxf(mx, acc[15]);
        if (MODE == 0) {
            best = fmaxf(best, mx);
        } else if (__builtin_amdgcn_ballot_w64(mx >= theta) != 0) {
            if (mx >= theta) {
                if ((cnt & 0x7fffffffu) < (uint32_t)kRecCap) {
                    f32x4 *dst = reinterpret_cast<f32x4 *>(p.lane_rec + ((size_t)mylane() * kRecCap + (cnt & 0x7fffffffu)) * 16);
                    dst[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
                    dst[1] = f32x4{acc[4], acc[5], acc[6], acc[7]};
                    dst[2] = f32x4{acc[8], acc[9], acc[10], acc[11]};
                    dst[3] = f32x4{acc[12], acc[13], acc[14], acc[15]};
                    p.lane_tile[(size_t)mylane() * kRecCap + (cnt & 0x7fffffffu)] = t0 + ti * tstep;
                    ++cnt;
                } else {
                    cnt |= 0x80000000u;  // overflow flag
                }
            }
        }
        }
    };

    if (par) {
#pragma unroll 1
        for (uint32_t ti = 0; ti < nT; ++ti) tile(std::integral_constant<int, 1>{}, ti);
    } else {
#pragma unroll 1
        for (uint32_t ti = 0; ti < nT; ++ti) tile(std::integral_constant<int, 0>{}, ti);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dead DMA ops must not outlive the workgroup's LDS

    if (par) {
        if (MODE == 0) {
            p.lane_max[mylane()] = best;
        } else {
            p.lane_cnt[mylane()] = cnt & 0x7fffffffu;
            if (cnt >> 31) p.overflow[grp * 32 + m] = 1;
        }
    }
}

template <int KC, int MODE>
static hipError_t setup16w_one() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&scan16w_kernel<KC, MODE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kScan16WideLdsBytes);
}

hipError_t scan16w_setup() {
    hipError_t e;
#define MX_SETUP(KC)                                            \
    if ((e = setup16w_one<KC, 0>()) != hipSuccess) return e;    \
    if ((e = setup16w_one<KC, 1>()) != hipSuccess) return e;
    MX_SETUP(8) MX_SETUP(10) MX_SETUP(12)
#undef MX_SETUP
    return hipSuccess;
}

template <int KC>
static hipError_t launch16w_kc(hipStream_t s, bool collect, int nwg, const ScanParams &p) {
    if (collect)
        hipLaunchKernelGGL((scan16w_kernel<KC, 1>), dim3(nwg), dim3(kScanThreads), kScan16WideLdsBytes, s, p);
    else
        hipLaunchKernelGGL((scan16w_kernel<KC, 0>), dim3(nwg), dim3(kScanThreads), kScan16WideLdsBytes, s, p);
    return hipGetLastError();
}

hipError_t launch_scan16w(hipStream_t s, int kc, bool collect, int nwg, const ScanParams &p) {
    switch (kc) {
        case 8: return launch16w_kc<8>(s, collect, nwg, p);
        case 10: return launch16w_kc<10>(s, collect, nwg, p);
        case 12: return launch16w_kc<12>(s, collect, nwg, p);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mx
