// scan16.hip -- the streaming cosine scan over the bf16 filter copy of the corpus.
//
// Same role as scan.hip (exhaustive replacement of `self.hnsw.search(vec, limit, 32)`, reference
// lib/libmemex/src/storage/local.rs:76) at half the HBM bytes: next to the f32 rows that the exact
// rescoring reads, the index keeps bf16(c_i / |c|) for every row, stored in MFMA A-fragment order
// (index_kernels.h, launch_shadow).  The accumulator of a row IS its approximate cosine:
//   approx = sum_f32 bf16(q_i/|q|) * bf16(c_i/|c|)
//   |approx - cos| <= (2^-7 + 2^-16) * |q^||c^|   two bf16 roundings (unit roundoff 2^-8) + Cauchy-Schwarz
//                     + ~1e-4                      f32 normalisation and accumulation over <= 768 terms
//                  <  kApproxErr = 0.0081          (the same certificate as scan.hip; index.hip relies on it)
// A zero-norm row is stored as zeros (score 0); its exact dist is 0 (DistCosine's else-branch), so the
// index keeps such rows in a short list that finish_kernel adds to every query's candidates.
//
// One persistent 512-thread workgroup per CU.  Per 8 KiB slot (32 rows x 128 dims):
//   1. each wave issues ONE 1 KiB LDS-DMA (16 B per lane, nt policy): wave w moves k-step w of the
//      slot.  HBM address and LDS address are both lane-linear -> perfectly coalesced, no staging, no
//      conversion pass.  15 slots (120 KiB per CU) are in flight; the stream never branches: past the
//      last tile the descriptor's num_records is 0 and the DMA touches no memory.
//   2. s_waitcnt vmcnt(13) + one s_barrier make the next slot visible to all waves.
//   3. every wave multiplies the slot against its own 32 queries (register-resident B fragments):
//      8 v_mfma_f32_32x32x16_bf16, each followed by ONE conflict-free ds_read_b128 that refills the
//      fragment register it just consumed with the fragment R k-steps ahead (register ring, R = 8
//      for dim_pad <= 512, R = 4 above: the B fragments of 768 dims take 192 of the 256 VGPRs), so
//      LDS reads, DMA issue and scalar bookkeeping all issue in the shadow of the MFMA pipe.
// Epilogue per 32-row tile, by MODE:
//   MODE 1 (collect): v_max3 tree over the lane's 16 scores, ONE compare with the query's pass
//                     threshold; a passing lane stores all 16 scores as a 64-byte record (no per-row
//                     code on the stream: finish_kernel picks the rows).
//   MODE 0 (sample):  keep only the lane's running maximum.  The k-th largest of a query's lane
//                     maxima (2 lanes per workgroup) is a certified lower bound of its k-th best
//                     approximate score: index.hip turns it into the pass threshold of the collect
//                     launch (theta_kernel), so no score of the sample is ever written to HBM.
// Tiles are dealt round-robin: workgroup b handles tiles tile_begin + (b + i*grid)*tile_stride.
// tile_stride > 1 spreads the sample evenly over the corpus (a contiguous head of the corpus can be
// unrepresentative: the first documents ingested).
//
// Measured (10M x 384, B = 256; profiles/r2_scan16_traffic.json, r2_power_scan16_*.log): the collect launch takes
// 1.73-1.78 ms = 4.3-4.4 TB/s at the 1400 W package power cap, shader clock 1.44-1.46 GHz, MFMA pipe 71 % busy
// in cycles (81 % at 768-d); the DMA stream alone runs at 7.1 TB/s and the MFMA + fragment-read half alone at
// 1.86 PFLOP/s, both at full clock and below the cap: together they need more than the cap allows, so what
// pays is energy per launch, not cycles.
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

namespace {
template <int N>
using ic16 = std::integral_constant<int, N>;
template <int B, int E, class F>
__device__ __forceinline__ void static_for16(F &&f) {
    if constexpr (B < E) {
        f(ic16<B>{});
        static_for16<B + 1, E>(f);
    }
}
// DMA operations a wave has issued after the one of slot j+1 when it waits for that slot: the slots lo .. hi (relative to the
// tile start), plus the one a_c operation of every tile that starts among them (as scan8_kernel's tile scales)
template <int KC>
constexpr int ops_after16(int lo, int hi) {
    int n = 0;
    for (int i = lo; i <= hi; ++i) n += 1 + (i % KC == 0 ? 1 : 0);
    return n;
}
static_assert(ops_after16<3>(2, 14) == 17 && ops_after16<6>(5, 17) == 15 && ops_after16<1>(2, 14) == 26, "");
}  // namespace

// Ablation switch for scripts/scan16_ubench.hip only (0 = production kernel):
//   1 = DMA + waits + barriers, 2 = + fragment reads (no MFMA), 4 = everything except the DMA
//   (compute on whatever LDS holds).  MX_SCAN16_CLOCK: lane_cnt receives the workgroup's s_memtime
//   cycle count instead of the candidate count (effective-clock probe).
#ifndef MX_SCAN16_ABLATE
#define MX_SCAN16_ABLATE 0
#endif

#ifndef MX_SCAN16_AUX
#define MX_SCAN16_AUX 2 /* nt */
#endif

template <int KC, int MODE>
__global__ __launch_bounds__(kScanThreads, 2) void scan16_kernel(const ScanParams p) {
    // Fragment ring: the A fragment of k-step f lives in ring[f % R] and is re-read R k-steps ahead
    // right after the MFMA that consumed it.  8 % R == 0, so the ring index of a k-step does not
    // depend on the slot and is a compile-time constant everywhere.
    constexpr int R = KC <= 4 ? 8 : 4;
    constexpr bool DUAL = KC <= 4;  // two accumulator chains (16 more VGPRs)
    extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef MX_SCAN16_CLOCK
    const uint64_t clk0 = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31;  // query column of B and D (lane >> 5 selects which 16 of the tile's 32 rows the lane scores)
    // a wave none of whose 32 query columns is wanted only keeps the DMA stream and the barriers going (ScanParams::wave_mask)
    // (not at 768 dims: the query fragments take 192 of the 256 VGPRs there and the branch costs the rest)
    const bool live = KC == 6 || ((p.wave_mask >> wave) & 1u) != 0;

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
    const uint32_t stride = p.tile_stride;
    const uint32_t t0 = p.tile_begin + blockIdx.x * stride;
    const uint32_t tstep = grid * stride;
    const uint32_t nT = (t0 < p.tile_end) ? (p.tile_end - t0 + tstep - 1) / tstep : 0;
    const uint32_t tilebytes = p.ds * (kTileRows * 2);

    const uint32_t voff = (uint32_t)wave * 1024u + (uint32_t)lane * 16u;  // this lane's 16 B of a slot
    const uint32_t lane16 = (uint32_t)lane * 16u;

    // ---- DMA stream.  Slots are issued strictly in order, 15 ahead of the slot being multiplied,
    // and ALWAYS: past the last tile the buffer descriptor has num_records = 0, so the load is
    // out of range and touches no memory.  That keeps the loop free of "is there more?" branches and
    // the vmcnt arithmetic uniform (exactly one DMA op per slot per wave, dead or alive).
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t is_ti = 0;  // tiles opened so far
    // Centred copy (ScanParams::amean): the tile's 32 values a_c travel in the same stream, one 128-byte operation issued right
    // before the tile's first slot -- they have landed when that slot has.  (A scalar load per tile exposes an HBM latency
    // per tile -- a tile is 1.4 us of work --, a vector load puts its own wait on the ring's vmcnt.)  Every wave issues it,
    // also for a plain copy (num_records = 0: no memory touched): same bytes, same LDS address, and the waits below stay
    // wave-uniform compile-time arithmetic.
    const bool centred = p.amean != nullptr;
    const uint32_t lane4 = (uint32_t)lane * 4u;
    auto open_tile = [&]() {
        const uint32_t tile = t0 + is_ti * tstep;
        const bool live_tile = is_ti < nT;
        __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)(p.amean + (size_t)tile * kTileRows), 0,
                                                                         live_tile && centred ? (uint32_t)(kTileRows * 4) : 0u, 0x00020000);
        char *adst = smem + __builtin_amdgcn_readfirstlane(kRing16 * kSlot16Bytes + (is_ti & (kMeanRing16 - 1)) * 256);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lds_void *)adst, 4, lane4, 0, 0, 0);
        const char *base = reinterpret_cast<const char *>(p.xh) + (size_t)tile * tilebytes;
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, live_tile ? tilebytes : 0u, 0x00020000);
        ++is_ti;
    };
    auto issue = [&](int kci, uint32_t ring_pos) {  // kci is a compile-time constant at every call site
#if MX_SCAN16_ABLATE != 4
        if (kci == 0) open_tile();
        char *dst = smem + __builtin_amdgcn_readfirstlane(ring_pos * kSlot16Bytes + wave * 1024);
        MX_LDS_DMA16(rsrc, dst, voff, kci * kSlot16Bytes, MX_SCAN16_AUX);
#endif
    };

    const size_t mylane = (size_t)tid * gridDim.x + blockIdx.x;
    f32x4 *myrec = reinterpret_cast<f32x4 *>(p.lane_rec + mylane * (kRecCap * 16));
    uint32_t *mytile = p.lane_tile + mylane * kRecCap;
    uint32_t cnt = 0;
    uint32_t ovf = 0;
    float best = -INFINITY;  // MODE 0: running maximum of this lane's scores

    // ---- prologue: slots 0 .. 14 in flight, fragments of k-steps 0 .. R-1 in the ring
#pragma unroll
    for (int i = 0; i < kRing16 - 1; ++i) issue(i % KC, (uint32_t)i);

    bf16x8 a[R];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ops_after16<KC>(1, kRing16 - 2)) : "memory");  // slot 0 (and its tile's a_c) landed
    __builtin_amdgcn_s_barrier();
#if MX_SCAN16_ABLATE != 1
#pragma unroll
    for (int ks = 0; ks < R; ++ks) a[ks] = *reinterpret_cast<const bf16x8 *>(smem + lane16 + ks * 1024);
#endif

    // a_q of this lane's query waits in LDS behind the rings (1 KiB; a register held across the loop is one more than the
    // 768-dim kernel has): written and read by the same wave, so program order is all the synchronisation it needs
    float *aq_lds = reinterpret_cast<float *>(smem + kRing16 * kSlot16Bytes + kMeanRing16 * 256) + wave * 32 + m;
    if (centred && lane < 32) *aq_lds = p.qmean[wave * 32 + m];

    uint32_t rp = 0;  // ring position of the slot being multiplied
#pragma unroll 1
    for (uint32_t ti = 0; ti < nT; ++ti) {
        f32x16 acc, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f, acc1[r] = 0.0f;

        static_for16<0, KC>([&](auto kct) __attribute__((always_inline)) {
            constexpr int kc = decltype(kct)::value;
            const uint32_t rp1 = (rp + 1) & (kRing16 - 1);
            const uint32_t rpi = (rp + kRing16 - 1) & (kRing16 - 1);  // ring position of slot j+15 = of slot j-1
            // Slot j+1 (the ring reads ahead into it) must have landed; slots 0 .. j+14 are issued ->
            // the 13 newer ones (and the a_c operations of the tiles that start among them) may be in flight.
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ops_after16<KC>(kc + 2, kc + kRing16 - 2)) : "memory");
            // One barrier per slot: every wave's piece of slot j+1 is in LDS, and every wave has
            // consumed (MFMA issued) its fragments of slot j-1, whose ring position is refilled below.
            __builtin_amdgcn_s_barrier();
            const uint32_t fb0 = rp * kSlot16Bytes + lane16, fb1 = rp1 * kSlot16Bytes + lane16;
            if (!live) {
                issue((kc + kRing16 - 1) % KC, rpi);
                rp = rp1;
                return;
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#if MX_SCAN16_ABLATE == 1 || MX_SCAN16_ABLATE == 2
                asm volatile("" ::"v"(a[ks % R]));
#else
#ifdef MX_SCAN16_I8_PROBE  /* scripts/scan16_ubench.hip: timing / power of the i8 instruction on the same stream (values meaningless) */
                {
                    typedef __attribute__((ext_vector_type(4))) int i32x4_;
                    typedef __attribute__((ext_vector_type(16))) int i32x16_;
                    f32x16 &dst = (DUAL && (ks & 1)) ? acc1 : acc;
                    dst = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4_, a[ks % R]), __builtin_bit_cast(i32x4_, qf[kc * 8 + ks]),
                                                                                       __builtin_bit_cast(i32x16_, dst), 0, 0, 0));
                }
#else
                if (DUAL && (ks & 1))
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % R], qf[kc * 8 + ks], acc1, 0, 0, 0);
                else
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % R], qf[kc * 8 + ks], acc, 0, 0, 0);
#endif
#endif
                // refill the register just consumed with the fragment R k-steps ahead (this slot or
                // the next one): the read issues while the MFMA pipe works on the instruction above
#if MX_SCAN16_ABLATE != 1
                a[ks % R] = *reinterpret_cast<const bf16x8 *>(smem + (ks + R < 8 ? fb0 : fb1) + ((ks + R) & 7) * 1024);
#endif
                if (ks == 1) issue((kc + kRing16 - 1) % KC, rpi);
                __builtin_amdgcn_sched_barrier(0);
            }
            rp = rp1;
        });

        if (!live) continue;
        // ---- tile epilogue: lane holds query (wave*32 + m), rows (r&3) + 8*(r>>2) + 4*h.
        // The copy holds c/|c|, so the accumulator already is the approximate cosine (a zero-norm row is
        // stored as zeros and scores 0: such rows reach finish_kernel through the index's zero-row list).
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = DUAL ? acc[r] + acc1[r] : acc[r];
        if (centred) {  // score = a_q a_c + r_q . r_c; this lane's rows: (r & 3) + 8 (r >> 2) + 4 h
            const float aq = *aq_lds;
            const f32x4 *am = reinterpret_cast<const f32x4 *>(smem + kRing16 * kSlot16Bytes + (ti & (kMeanRing16 - 1)) * 256) + (lane >> 5);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a4 = am[2 * j];  // rows 8 j + 4 h .. + 3
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = __builtin_fmaf(aq, a4[e], v[4 * j + e]);
            }
        }
        float mx = fmaxf(fmaxf(v[0], v[1]), v[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, v[r]), v[r + 1]);
        mx = fmaxf(mx, v[15]);
        if (MODE == 0) {
            // sample: only full tiles are sampled (index.hip), so every row is a real row
            best = fmaxf(best, mx);
        } else if (__builtin_amdgcn_ballot_w64(mx >= theta) != 0) {
            // a few percent of the tiles: the lanes that pass store their 16 scores as ONE record
            // (4 x 16 bytes + the tile index); which rows pass is sorted out by finish_kernel
            if (mx >= theta) {
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dead DMA ops must not outlive the workgroup's LDS

#ifdef MX_SCAN16_CLOCK
    cnt = (uint32_t)(__builtin_amdgcn_s_memtime() - clk0);
#endif
    if (MODE == 0) {
        p.lane_max[(size_t)tid * gridDim.x + blockIdx.x] = best;
    } else {
        p.lane_cnt[(size_t)tid * gridDim.x + blockIdx.x] = cnt;
        if (ovf) p.overflow[wave * 32 + m] = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// filter-copy construction: one workgroup per 32-row tile, one 16-B fragment per thread and step
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shadow_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                     int ds, uint32_t tile0, uint32_t tile1,
                                                     bf16x8 *__restrict__ xh, uint32_t *__restrict__ ec_max,
                                                     uint32_t src_tile0, uint64_t row_lo, uint64_t row_hi,
                                                     const float *__restrict__ mean, float *__restrict__ amean) {
    // per row: |bf16(c/|c|) - c/|c||^2 (the row's share of the scan's error bound) and |bf16(c/|c|)|^2
    __shared__ float s_r2[kTileRows], s_n2[kTileRows];
    __shared__ double s_a[kTileRows];  // centred form: a_c = (c/|c|) . mean, summed in f64 (the identity
                                       // c/|c| = a_c mean + r_c must hold to ~1e-7 for the scan's certificate)
    const uint32_t frags = (uint32_t)(ds / 16) * 64u;  // fragments per tile
    float worst = 0.0f;
    for (uint32_t t = tile0 + blockIdx.x; t < tile1; t += gridDim.x) {
        const float *xt = x + (size_t)(t - src_tile0) * kTileRows * ds;
        bf16x8 *ot = xh + (size_t)t * frags;
        if (threadIdx.x < kTileRows) s_r2[threadIdx.x] = 0.0f, s_n2[threadIdx.x] = 0.0f, s_a[threadIdx.x] = 0.0;
        __syncthreads();
        if (mean) {
            for (uint32_t f = threadIdx.x; f < frags; f += 256) {
                const uint32_t ks = f >> 6, l = f & 63, mm = l & 31, hh = l >> 5;
                const uint64_t grow = (uint64_t)t * kTileRows + mm;
                if (grow < row_lo || grow >= row_hi) continue;
                const f32x4 *src = reinterpret_cast<const f32x4 *>(xt + (size_t)mm * ds + ks * 16 + hh * 8);
                const f32x4 *mp = reinterpret_cast<const f32x4 *>(mean + ks * 16 + hh * 8);
                const float sc = scale[(size_t)(t - src_tile0) * kTileRows + mm];
                const f32x4 lo = src[0] * sc, hi = src[1] * sc, m0 = mp[0], m1 = mp[1];
                double d = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i) d += (double)lo[i] * (double)m0[i] + (double)hi[i] * (double)m1[i];
                atomicAdd(&s_a[mm], d);
            }
            __syncthreads();
            if (threadIdx.x < kTileRows) {
                const uint64_t grow = (uint64_t)t * kTileRows + threadIdx.x;
                if (grow >= row_lo && grow < row_hi) amean[grow] = (float)s_a[threadIdx.x];
            }
        }
        for (uint32_t f = threadIdx.x; f < frags; f += 256) {
            const uint32_t ks = f >> 6, l = f & 63, mm = l & 31, hh = l >> 5;
            const uint64_t grow = (uint64_t)t * kTileRows + mm;
            if (grow < row_lo || grow >= row_hi) continue;
            const f32x4 *src = reinterpret_cast<const f32x4 *>(xt + (size_t)mm * ds + ks * 16 + hh * 8);
            const float sc = scale[(size_t)(t - src_tile0) * kTileRows + mm];  // 1/|c|; 0 for a zero-norm row -> stored as zeros
            f32x4 lo = src[0] * sc, hi = src[1] * sc;
            if (mean) {  // r_c = c/|c| - a_c mean, with the a_c the scan will use (the f32 value just stored)
                const float ac = (float)s_a[mm];
                const f32x4 *mp = reinterpret_cast<const f32x4 *>(mean + ks * 16 + hh * 8);
                const f32x4 m0 = mp[0], m1 = mp[1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    lo[i] = __builtin_fmaf(-ac, m0[i], lo[i]);
                    hi[i] = __builtin_fmaf(-ac, m1[i], hi[i]);
                }
            }
            bf16x8 o;
            o[0] = (__bf16)lo[0]; o[1] = (__bf16)lo[1]; o[2] = (__bf16)lo[2]; o[3] = (__bf16)lo[3];
            o[4] = (__bf16)hi[0]; o[5] = (__bf16)hi[1]; o[6] = (__bf16)hi[2]; o[7] = (__bf16)hi[3];
            ot[f] = o;
            float r2 = 0.0f, n2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = (float)o[i] - lo[i], b = (float)o[4 + i] - hi[i];
                r2 += a * a + b * b;
                n2 += (float)o[i] * (float)o[i] + (float)o[4 + i] * (float)o[4 + i];
            }
            atomicAdd(&s_r2[mm], r2);
            atomicAdd(&s_n2[mm], n2);
        }
        __syncthreads();
        if (threadIdx.x < kTileRows) {
            // the stored row is not exactly unit: ||c^| - 1| enters the bound when the copy is the corpus (never centred)
            const float dev = !mean && s_n2[threadIdx.x] > 0.0f ? fabsf(sqrtf(s_n2[threadIdx.x]) - 1.0f) : 0.0f;
            worst = fmaxf(worst, fmaxf(sqrtf(s_r2[threadIdx.x]), dev));
        }
        __syncthreads();
    }
    if (threadIdx.x < kTileRows) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) worst = fmaxf(worst, __shfl_xor(worst, o));
        if (threadIdx.x == 0 && worst > 0.0f) atomicMax(ec_max, __float_as_uint(worst));
    }
}

hipError_t launch_shadow(hipStream_t s, const float *x, const float *scale, int ds, uint32_t tile0, uint32_t tile1,
                         void *xh, uint32_t *ec_max, uint32_t src_tile0, uint64_t row_lo, uint64_t row_hi, const float *mean,
                         float *amean) {
    if (tile1 <= tile0) return hipSuccess;
    if (mean && !amean) return hipErrorInvalidValue;
    const uint32_t blocks = tile1 - tile0 < 16384u ? tile1 - tile0 : 16384u;
    hipLaunchKernelGGL(shadow_kernel, dim3(blocks), dim3(256), 0, s, x, scale, ds, tile0, tile1,
                       reinterpret_cast<bf16x8 *>(xh), ec_max, src_tile0, row_lo, row_hi, mean, amean);
    return hipGetLastError();
}

template <int KC, int MODE>
static hipError_t setup16_one() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&scan16_kernel<KC, MODE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kScan16LdsBytes);
}

hipError_t scan16_setup() {
    hipError_t e;
#define MX_SETUP(KC)                                           \
    if ((e = setup16_one<KC, 0>()) != hipSuccess) return e;    \
    if ((e = setup16_one<KC, 1>()) != hipSuccess) return e;
    MX_SETUP(1) MX_SETUP(2) MX_SETUP(3) MX_SETUP(4) MX_SETUP(5) MX_SETUP(6)
#undef MX_SETUP
    return hipSuccess;
}

template <int KC>
static hipError_t launch16_kc(hipStream_t s, bool collect, int nwg, const ScanParams &p) {
    if (collect)
        hipLaunchKernelGGL((scan16_kernel<KC, 1>), dim3(nwg), dim3(kScanThreads), kScan16LdsBytes, s, p);
    else
        hipLaunchKernelGGL((scan16_kernel<KC, 0>), dim3(nwg), dim3(kScanThreads), kScan16LdsBytes, s, p);
    return hipGetLastError();
}

hipError_t launch_scan16(hipStream_t s, int kc, bool collect, int nwg, const ScanParams &p) {
    switch (kc) {
        case 1: return launch16_kc<1>(s, collect, nwg, p);
        case 2: return launch16_kc<2>(s, collect, nwg, p);
        case 3: return launch16_kc<3>(s, collect, nwg, p);
        case 4: return launch16_kc<4>(s, collect, nwg, p);
        case 5: return launch16_kc<5>(s, collect, nwg, p);
        case 6: return launch16_kc<6>(s, collect, nwg, p);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mx
