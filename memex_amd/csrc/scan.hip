// scan.hip -- the HBM-streaming cosine scan (kernel K6 of SURVEY.md section 2.2).
//
// Replaces the graph walk of `self.hnsw.search(vec, limit, 32)`
// (reference lib/libmemex/src/storage/local.rs:76) with an exhaustive pass over the corpus.
//
// One persistent 512-thread workgroup per CU.  Data path per 32-row x 128-float "slot" (16 KiB):
//   1. buffer_load ... lds (LDS-DMA, nt policy, no VGPR staging) streams the f32 slot into an
//      8-deep LDS ring.  Wave w DMAs rows 4w..4w+3 and is the ONLY reader of those rows, so the
//      ring needs no barrier: a counted s_waitcnt vmcnt covers it.
//   2. each wave converts its 4 rows to bf16 once (2 ds_read_b128 + 4 v_cvt_pk_bf16_f32 +
//      1 ds_write_b128 per lane) into a padded, double-buffered bf16 tile shared by the workgroup.
//   3. after one s_barrier, all 8 waves read MFMA A-fragments (32 corpus rows x 16 k) from the bf16
//      tile with immediate-offset ds_read_b128 and run v_mfma_f32_32x32x16_bf16 against their own
//      32 queries, held as register-resident B-fragments for the whole launch.
// Steps 2 (slot j+1) and 3 (slot j) are independent and interleave.
// With A = corpus rows and B = queries every lane ends a tile holding 16 corpus-row scores of ONE
// query: scale by 1/|c|, take their maximum, compare with that query's pass threshold; a passing lane
// stores the 16 scores as one record in HBM.  No top-k bookkeeping, atomics or cross-lane traffic on
// the streaming path.
//
// Algorithmic bytes: rows * ds * 4 per launch (+ rows*4 for 1/|c|); MFMA work 2*256*rows*ds flop
// (512 MFMA cycles per SIMD per slot vs ~1250 cycles of HBM time per slot per CU: HBM-bound).
// Approximate scores are within kApproxErr of the exact cosine; exactness of the final answer is
// restored by index_kernels.hip (pool select + f64 rescoring).
#include "index_kernels.h"

// Ablation switch for scripts/scan_ubench.hip only (0 = production kernel):
//   1 = DMA + waits + LDS reads of the conversion pass, 2 = + bf16 conversion and tile writes,
//   3 = + fragment reads (no MFMA)
#ifndef MX_SCAN_ABLATE
#define MX_SCAN_ABLATE 0
#endif

namespace mx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

#define MX_LDS_DMA16(rsrc, ldsptr, voff, soff, aux) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lds_void *)(ldsptr), 16, (voff), (soff), 0, (aux))

constexpr int kTilePitch = kChunkFloats * 2 + 16;          // bf16 tile row pitch: 272 B (odd # of 16-B slots)
constexpr int kTileBytes = kTileRows * kTilePitch;         // 8704 B
constexpr uint32_t kTileOff = kNumSlots * kSlotBytes;      // two bf16 tiles after the ring
constexpr uint32_t kScaleOff = kTileOff + 2 * kTileBytes;  // per-tile 1/|c| ring
static_assert(kScaleOff + kScaleRing * kTileRows * 4 == kScanLdsBytes, "LDS layout");
static_assert(kPrefetch == kNumSlots && kNumSlots == 8, "waits below assume an 8-slot ring, all in flight");

template <int KC, int MODE>
__global__ __launch_bounds__(kScanThreads, 2) void scan_kernel(const ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31;  // MFMA: corpus row of the A fragment / query column of B and D
    const int h = lane >> 5;

    // ---- register-resident query fragments (B operand), loaded once per launch
    bf16x8 qf[KC * 8];
    {
        const bf16x8 *src = reinterpret_cast<const bf16x8 *>(p.qfrag) + (size_t)wave * (KC * 8) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KC * 8; ++i) qf[i] = src[(size_t)i * 64];
    }
    const float theta = MODE == 1 ? p.theta[wave * 32 + m] : 0.0f;

    // ---- tiles of this workgroup: tile_begin + (blockIdx + i*grid) * tile_stride
    const uint32_t grid = gridDim.x;
    const uint32_t t0 = p.tile_begin + blockIdx.x * p.tile_stride;
    const uint32_t tstep = grid * p.tile_stride;
    const uint32_t nT = (t0 < p.tile_end) ? (p.tile_end - t0 + tstep - 1) / tstep : 0;
    const uint32_t total = nT * KC;  // slots this workgroup consumes
    const uint32_t rowbytes = p.ds * 4;

    // ---- LDS-DMA lane constants.  Piece P (1 KiB) of a slot = rows 2P, 2P+1 x 512 B; wave w
    // issues pieces 2w, 2w+1 (rows 4w..4w+3).  The LDS image is lane-linear, so the layout the
    // conversion pass wants -- even 16-B chunks in the first 256 B of a row, odd chunks in the
    // second -- is produced by permuting the SOURCE address: physical chunk pc <- logical chunk
    // ((pc & 15) << 1) | (pc >> 4).
    const int pc = lane & 31;
    const uint32_t lchunk = (uint32_t)(((pc & 15) << 1) | (pc >> 4));
    const uint32_t voff0 = (uint32_t)(4 * wave + h) * rowbytes + (lchunk << 4);
    const uint32_t voff1 = voff0 + 2u * rowbytes;

    uint32_t nx = 0, nx_ti = 0, nx_kc = 0, nx_rp = 0;  // next slot to issue (all wave-uniform)
    auto issue_next = [&]() {
        if (nx >= total) return;
        const uint32_t t = t0 + nx_ti * tstep;
        const char *base = reinterpret_cast<const char *>(p.x) + (size_t)t * kTileRows * rowbytes;
        __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, kTileRows * rowbytes, 0x00020000);
        if (nx_kc == 0 && wave == 0) {
            // this tile's 32 x 1/|c| (128 B): one masked DMA, older than the tile's first slot
            __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(p.scale + (size_t)t * kTileRows), 0, kTileRows * 4, 0x00020000);
            if (lane < 8) MX_LDS_DMA16(srs, smem + kScaleOff + (nx_ti & (kScaleRing - 1)) * 128, lane * 16, 0, 0);
        }
        // readfirstlane: the masked scale DMA above makes control flow divergent, and hipcc would
        // otherwise carry these wave-uniform values in VGPRs and wrap each DMA in a waterfall loop
        char *dst = smem + __builtin_amdgcn_readfirstlane(nx_rp * kSlotBytes + wave * 2048);
        const int soff = __builtin_amdgcn_readfirstlane((int)(nx_kc * (kChunkFloats * 4)));
        MX_LDS_DMA16(rsrc, dst, voff0, soff, 2 /*nt*/);
        MX_LDS_DMA16(rsrc, dst + 1024, voff1, soff, 2 /*nt*/);
        ++nx;
        if (++nx_kc == KC) { nx_kc = 0; ++nx_ti; }
        if (++nx_rp == kNumSlots) nx_rp = 0;
    };

    // conversion pass of the slot at ring position rp into bf16 tile tb: this wave's own 4 rows.
    // lane -> row 4w + (lane>>4), floats [8*(lane&15), +8) = physical chunks (lane&15), 16+(lane&15)
    const uint32_t cv_src = (uint32_t)wave * 2048u + (uint32_t)(lane >> 4) * 512u + (uint32_t)(lane & 15) * 16u;
    const uint32_t cv_dst = kTileOff + (uint32_t)(4 * wave + (lane >> 4)) * kTilePitch + (uint32_t)(lane & 15) * 16u;
    auto convert = [&](uint32_t rp, uint32_t tb) {
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(smem + rp * kSlotBytes + cv_src);
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(smem + rp * kSlotBytes + cv_src + 256);
#if MX_SCAN_ABLATE == 1
        asm volatile("" ::"v"(lo), "v"(hi));
        (void)tb;
#else
        bf16x8 a;
        a[0] = (__bf16)lo[0]; a[1] = (__bf16)lo[1]; a[2] = (__bf16)lo[2]; a[3] = (__bf16)lo[3];
        a[4] = (__bf16)hi[0]; a[5] = (__bf16)hi[1]; a[6] = (__bf16)hi[2]; a[7] = (__bf16)hi[3];
        *reinterpret_cast<bf16x8 *>(smem + cv_dst + tb * kTileBytes) = a;
#endif
    };

    // A-fragment read base: row m, bf16 [8h, 8h+8) of k-step ks at + ks*32 (immediate offsets)
    const uint32_t frag_base = kTileOff + (uint32_t)m * kTilePitch + (uint32_t)h * 16u;

    // lane buffers are laid out [thread-in-workgroup][workgroup]: everything one query ever receives
    // (2 lanes x all workgroups) is contiguous for the gather in update_kernel
    const size_t mylane = (size_t)tid * gridDim.x + blockIdx.x;
    f32x4 *myrec = reinterpret_cast<f32x4 *>(p.lane_rec + mylane * (kRecCap * 16));
    uint32_t *mytile = p.lane_tile + mylane * kRecCap;
    uint32_t cnt = 0;
    uint32_t ovf = 0;
    float best = -INFINITY;  // MODE 0 (sample): running maximum of this lane's scores (scan16.hip)

    // ---- prologue: fill the ring, convert slot 0
#pragma unroll 1
    for (int i = 0; i < kNumSlots; ++i) issue_next();
    if (total > 0) {
        if (total >= (uint32_t)kNumSlots)
            asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // 7 newer slots x 2 ops may stay in flight
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        convert(0, 0);
    }

    uint32_t rp = 0;  // ring position of slot j
    uint32_t j = 0;
#pragma unroll 1
    for (uint32_t ti = 0; ti < nT; ++ti) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

#pragma unroll
        for (int kc = 0; kc < KC; ++kc, ++j) {
            // Issued so far: slots 0 .. j+7.  Slot j+1 (converted below) has landed once at most
            // the 6 newer slots (12 DMA ops) are outstanding; near the end just drain.
            // lgkmcnt(0): this wave's bf16 tile writes of the previous iteration are done before
            // it arrives at the barrier.
            if (j + (uint32_t)kNumSlots <= total)
                asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            // One barrier per slot: tile j&1 (written last iteration) is complete and visible;
            // tile (j+1)&1 (read last iteration) is free to overwrite.
            __builtin_amdgcn_s_barrier();
            const uint32_t rp1 = (rp + 1 == (uint32_t)kNumSlots) ? 0 : rp + 1;
            const bool more = j + 1 < total;

            // (a) every LDS read of this iteration up front: the slot's 8 A-fragments and the two
            //     f32 chunks of the conversion pass -> one LDS latency per slot
            const uint32_t fb = frag_base + (j & 1) * kTileBytes;
            bf16x8 a[8];
#if MX_SCAN_ABLATE == 0 || MX_SCAN_ABLATE == 3
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) a[ks] = *reinterpret_cast<const bf16x8 *>(smem + fb + ks * 32);
#endif
            f32x4 lo, hi;
            if (more) {
                lo = *reinterpret_cast<const f32x4 *>(smem + rp1 * kSlotBytes + cv_src);
                hi = *reinterpret_cast<const f32x4 *>(smem + rp1 * kSlotBytes + cv_src + 256);
            }
            __builtin_amdgcn_sched_barrier(0);
            // (b) DMA issue (scalar address math) under the LDS latency: slot j+8 -> ring
            //     position of slot j, which this wave finished converting last iteration
            issue_next();
            __builtin_amdgcn_sched_barrier(0);
            // (c) MFMAs; the VALU/LDS-write tail of the conversion pass sits between them
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#if MX_SCAN_ABLATE == 3
                asm volatile("" ::"v"(a[ks]));
#elif MX_SCAN_ABLATE == 0
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], qf[kc * 8 + ks], acc, 0, 0, 0);
#endif
                if (ks == 3 && more) {
#if MX_SCAN_ABLATE == 1
                    asm volatile("" ::"v"(lo), "v"(hi));
#else
                    bf16x8 c;
                    c[0] = (__bf16)lo[0]; c[1] = (__bf16)lo[1]; c[2] = (__bf16)lo[2]; c[3] = (__bf16)lo[3];
                    c[4] = (__bf16)hi[0]; c[5] = (__bf16)hi[1]; c[6] = (__bf16)hi[2]; c[7] = (__bf16)hi[3];
                    *reinterpret_cast<bf16x8 *>(smem + cv_dst + ((j + 1) & 1) * kTileBytes) = c;
#endif
                }
            }
            rp = rp1;
        }

        // ---- tile epilogue: lane holds query (wave*32 + m), rows (r&3) + 8*(r>>2) + 4*h
        const float *sc = reinterpret_cast<const float *>(smem + kScaleOff + (ti & (kScaleRing - 1)) * 128) + 4 * h;
        const f32x4 s0 = *reinterpret_cast<const f32x4 *>(sc);
        const f32x4 s1 = *reinterpret_cast<const f32x4 *>(sc + 8);
        const f32x4 s2 = *reinterpret_cast<const f32x4 *>(sc + 16);
        const f32x4 s3 = *reinterpret_cast<const f32x4 *>(sc + 24);
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x4 sv = (r >> 2) == 0 ? s0 : (r >> 2) == 1 ? s1 : (r >> 2) == 2 ? s2 : s3;
            v[r] = acc[r] * sv[r & 3];  // 1/|c| is 0 for a zero-norm row: score 0 (see scan16.hip)
        }
        float mx = fmaxf(fmaxf(v[0], v[1]), v[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, v[r]), v[r + 1]);
        mx = fmaxf(mx, v[15]);
        if (MODE == 0) {
            best = fmaxf(best, mx);
        } else if (__builtin_amdgcn_ballot_w64(mx >= theta) != 0) {
            if (mx >= theta) {  // one 64-byte record per passing lane and tile (scan16.hip)
                if (cnt < (uint32_t)kRecCap) {
                    f32x4 *dst = myrec + cnt * 4;
                    dst[0] = f32x4{v[0], v[1], v[2], v[3]};
                    dst[1] = f32x4{v[4], v[5], v[6], v[7]};
                    dst[2] = f32x4{v[8], v[9], v[10], v[11]};
                    dst[3] = f32x4{v[12], v[13], v[14], v[15]};
                    mytile[cnt] = t0 + ti * tstep;
                    ++cnt;
                } else {
                    ovf = 1;
                }
            }
        }
    }

    if (MODE == 0) {
        p.lane_max[(size_t)tid * gridDim.x + blockIdx.x] = best;
        return;
    }
    p.lane_cnt[(size_t)tid * gridDim.x + blockIdx.x] = cnt;
    if (ovf) p.overflow[wave * 32 + m] = 1;
}

template <int KC, int MODE>
static hipError_t setup_one() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_kernel<KC, MODE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kScanLdsBytes);
}

hipError_t scan_setup() {
    hipError_t e;
#define MX_SETUP(KC)                                         \
    if ((e = setup_one<KC, 0>()) != hipSuccess) return e;    \
    if ((e = setup_one<KC, 1>()) != hipSuccess) return e;
    MX_SETUP(1) MX_SETUP(2) MX_SETUP(3) MX_SETUP(4) MX_SETUP(5) MX_SETUP(6)
#undef MX_SETUP
    return hipSuccess;
}

template <int KC>
static hipError_t launch_kc(hipStream_t s, bool collect, int nwg, const ScanParams &p) {
    if (collect)
        hipLaunchKernelGGL((scan_kernel<KC, 1>), dim3(nwg), dim3(kScanThreads), kScanLdsBytes, s, p);
    else
        hipLaunchKernelGGL((scan_kernel<KC, 0>), dim3(nwg), dim3(kScanThreads), kScanLdsBytes, s, p);
    return hipGetLastError();
}

// MODE 1 (collect) is the launch that covers the corpus; MODE 0 is the short sample launch.  They are
// distinct symbols, so rocprofv3 --stats averages them separately.
hipError_t launch_scan(hipStream_t s, int kc, bool collect, int nwg, const ScanParams &p) {
    switch (kc) {
        case 1: return launch_kc<1>(s, collect, nwg, p);
        case 2: return launch_kc<2>(s, collect, nwg, p);
        case 3: return launch_kc<3>(s, collect, nwg, p);
        case 4: return launch_kc<4>(s, collect, nwg, p);
        case 5: return launch_kc<5>(s, collect, nwg, p);
        case 6: return launch_kc<6>(s, collect, nwg, p);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mx
