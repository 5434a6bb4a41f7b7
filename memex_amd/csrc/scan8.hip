// scan8.hip -- the streaming cosine scan over an 8-bit filter copy of the corpus (SURVEY.md section 8 f-4:
// "bf16 / 8-bit rows + fp32 rerank"), and the kernel that builds that copy.
//
// Same role and same pipeline as scan16.hip (exhaustive replacement of `self.hnsw.search(vec, limit, 32)`,
// reference lib/libmemex/src/storage/local.rs:76): the scan only has to bring every row of the exact top-k
// into finish_kernel, which rescores in f32 and decides in f64 on the f32 rows.  scan16_kernel is bound by
// the package power cap, not by cycles, so what pays is joules per row -- and an
// int8 row costs half the HBM bytes, half the LDS bytes and one v_mfma_i32_32x32x32_i8 where bf16 needs two
// v_mfma_f32_32x32x16_bf16: the same 8-KiB slot stream at the same rate (scripts/r3_i8probe.sh) carries
// twice the rows.
//
// Quantisation (shadow8_kernel).  Rows are normalised and rotated (mx_rotate.h: a fixed orthonormal map, so dot
// products are unchanged and the elements look Gaussian whatever the embedding model's spectrum is).  Per 32-row
// half tile h:  s_h = max |rotated element| / 127 over its rows,  c8 = rint(rotated / s_h) in [-127, 127];  a query
// likewise (same rotation) with its own step s_q.  The accumulator is
// an exact integer, so for a row of half tile h
//   score = s_q * s_h * sum q8_i c8_i,   |score - cos| <= qa + qb * e_h,   qa = Eq + slack, qb = 1 + Eq
// with the MEASURED residual norms e_h = 1.01 max_rows-of-h |c/|c| - s_h c8| + 1e-6 (kept per half tile next to
// s_h) and Eq = |q/|q| - s_q q8| (prep_queries_kernel): 0.016 for a typical half tile at 384 dims, 0.022-0.027 for
// the worst one of a 10M-row corpus, against 0.004 for bf16 -- a few hundred candidates per query for
// finish_kernel instead of a few dozen, no change to the answers.
//
// Layout.  A scan tile is 64 rows = two half tiles u = 0, 1; slot s of tile T (8 KiB, 128 dims) holds the
// eight 1-KiB A operands f = 2j + u of k-step j (32 dims) and half u: lane l -> 16 int8 = row
// 64T + 32u + (l & 31), dims 128s + 32j + 16(l >> 5) .. + 15 (A and B use the same k order, which is all
// the dot product needs).  tscale[T] = (s_h0, s_h1, e_h0, e_h1, 1/s_h0, 1/s_h1, -, -): 32 bytes per tile, fetched by one
// more DMA operation one tile ahead of the tile's first slot -- and a second one for the a_c of a CENTRED copy (the rows
// stored minus their component along the corpus mean direction; a_q a_c comes back as the accumulators' initial value:
// scan8_kernel<..., CEN = true>).  The kernel is scan16_kernel with two accumulators (one per
// half) fed alternately: one persistent 512-thread workgroup per CU, one 1 KiB LDS-DMA per wave and slot, 15
// slots in flight, one s_waitcnt vmcnt(N) + s_barrier per slot (N counts the operations issued after the slot
// being waited for: 13 slots plus the per-tile operations among them, a compile-time constant per position in the
// tile), 8 MFMAs per slot each followed by the ds_read_b128 that refills the fragment register it consumed.
// The query fragments take dim_pad/8 VGPRs (48 at 384 dims, 192 at 1536), so one launch serves 256 queries at
// every supported width -- 512 up to 512 dims, with two query groups per wave (QG = 2).  A wave none of whose
// queries is wanted (small batches, retry passes: ScanParams::wave_mask) skips its MFMAs.
// Tile epilogue, per half: integer max over the lane's 16 sums, one conversion, two multiplications (s_h, s_q),
// one FMA (theta - qb * e_h), one compare; a passing lane stores its 16 scores as floats -- the record format of
// scan16_kernel with the 32-row tile index 2T + u, so theta_kernel and finish_kernel do not know which scan ran.
#include <type_traits>

#include "index_kernels.h"
#include "mx_rotate.h"

namespace mx {

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

#define MX_LDS_DMA16(rsrc, ldsptr, voff, soff, aux) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (lds_void *)(ldsptr), 16, (voff), (soff), 0, (aux))

static_assert(kRing16 == 16, "waits below assume a 16-slot ring with 15 slots in flight");

// Ablation switch for scripts/scan8_ubench.hip only (0 = production kernel): 1 = tile scales are constants (no LDS read),
// 2 = every second fragment read dropped
#ifndef MX_SCAN8_ABLATE
#define MX_SCAN8_ABLATE 0
#endif

namespace {
template <int N>
using ic = std::integral_constant<int, N>;
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (B < E) {
        f(ic<B>{});
        static_for<B + 1, E>(f);
    }
}
// DMA operations a wave has issued after the one of slot j+1 when it waits for that slot at position kc of a
// tile: the 13 slots j+2 .. j+14, plus the TWO per-tile operations (scales; a_c of the centred copy, issued with
// num_records = 0 for a plain one: the kernel has one wait schedule) of every tile that starts among them
template <int KC>
constexpr int ops_after(int lo, int hi) {  // slots lo .. hi relative to the tile start
    int n = 0;
    for (int i = lo; i <= hi; ++i) n += 1 + (i % KC == 0 ? 2 : 0);
    return n;
}
// 384 dims (3 slots per tile): after slot j+1 come 13 slots and the per-tile operations of the tiles that start at
// relative slots 3, 6, 9, 12 (position 0 and 2) or 3 .. 15 (position 1); before the loop, after slot 0: 14 + 2 x 4.
static_assert(ops_after<3>(2, 14) == 21 && ops_after<3>(3, 15) == 23 && ops_after<3>(4, 16) == 21 && ops_after<3>(1, 14) == 22, "");
static_assert(ops_after<1>(2, 14) == 39 && ops_after<12>(2, 14) == 15 && ops_after<12>(13, 25) == 15 && ops_after<6>(5, 17) == 17, "");
}  // namespace

// QG = query groups (of 32) per wave: 1 = a pass of 256 queries (the kernel every batch size up to 256 runs);
// 2 = a pass of 512 (each fragment read feeds two MFMAs; wave w holds the "virtual waves" 2w and 2w+1 of
// theta_kernel's / finish_kernel's lane numbering; up to 512 dims)
// CEN = the centred copy (KC <= kMaxKC, one query group): see below
template <int KC, int MODE, int QG, bool CEN>
__global__ __launch_bounds__(kScanThreads, 2) void scan8_kernel(const ScanParams p) {
    static_assert(!CEN || (KC <= kMaxKC && QG == 1), "the centred form exists up to kMaxKC slots with one query group");
    constexpr int R = QG == 2 ? (KC <= 3 ? 8 : 4) : KC <= 8 ? 8 : KC <= 10 ? 4 : 2;  // fragment ring: what the 256 VGPRs leave next to qf
    extern __shared__ __attribute__((aligned(16))) char smem[];  // slot ring | per-tile ring [kScaleRing8] x kScale8Entry B
    // Centred copy (ScanParams::amean, section 3.2c of DESIGN.md carried over to int8): the copy holds the quantised
    // r_c = c/|c| - a_c m, amean[row] = a_c, the query fragments the quantised r_q, qmean[q] = a_q; a row's score is
    // a_q a_c + s_h s_q sum.  The a_q a_c term enters as the ACCUMULATOR'S INITIAL VALUE, I = trunc(a_c (a_q / s_q) (1 / s_h)) in
    // units of the half tile's s_h s_q -- an int32 the MFMAs add the sum to -- so the tile epilogue is the plain copy's: one
    // maximum tree over 16 integers, one conversion, two multiplications.  (Evaluating a_q a_c + s_h s_q sum per row in the
    // epilogue -- the first form of this round -- ran on EVERY half tile of an encoder-shaped corpus: the quick upper-bound test in
    // front of it never stopped one, profiles/r6_centred_int8_accumulator_init.txt.)  |I| < 2^30 + 1100 because the builder and
    // prep_queries_kernel keep both steps of a centred copy at or above kMinStep8 = 2^-15; what the truncation and the f32
    // roundings of I cost a score (<= s_h s_q + 5e-7) is inside kAccSlack.
    constexpr bool centred = CEN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31;  // query column of B and D
    // A wave none of whose query columns is wanted (padding of a small batch; queries parked during a retry pass) only
    // keeps the DMA stream and the barriers going: a single query costs the stream, not 256 queries' worth of MFMA energy
    // (not at 1536 dims, and not with two query groups per wave: the registers the branch costs are not there)
    const bool live = KC == 12 || QG == 2 || ((p.wave_mask >> wave) & 1u) != 0;

    // ---- register-resident query fragments (B operand): k-step ks = 32 dims, 16 int8 per lane
    i32x4 qf[QG * KC * 4];
    // MODE 1: a half tile with residual bound e passes when a score reaches theta - qb * e (theta_kernel);
    // MODE 0: the lane keeps the best LOWER bound of a cosine, score - (qa + qb * e)
    float theta[QG], qa[QG], qb[QG], sq[QG];
    float aq = 0.0f;
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int vw = wave * QG + g;  // wave index in the lane numbering of theta_kernel / finish_kernel
        const i32x4 *src = reinterpret_cast<const i32x4 *>(p.qfrag) + (size_t)vw * (KC * 4) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KC * 4; ++i) qf[g * KC * 4 + i] = src[(size_t)i * 64];
        theta[g] = MODE == 1 ? p.theta[vw * 32 + m] : 0.0f;
        if (CEN && g == 0) aq = p.qmean[vw * 32 + m];
        qa[g] = MODE == 0 ? p.qa[vw * 32 + m] : 0.0f;
        qb[g] = p.qb[vw * 32 + m];
        sq[g] = p.qscale[vw * 32 + m];  // 0 for an unusable (zero / padded) query
    }
    const float kq = CEN && sq[0] > 0.0f ? aq / sq[0] : 0.0f;  // a_q in units of the query's step

    // ---- 64-row tiles of this workgroup: tile_begin + (blockIdx + i*grid) * tile_stride
    const uint32_t grid = gridDim.x;
    const uint32_t stride = p.tile_stride;
    const uint32_t t0 = p.tile_begin + blockIdx.x * stride;
    const uint32_t tstep = grid * stride;
    const uint32_t nT = (t0 < p.tile_end) ? (p.tile_end - t0 + tstep - 1) / tstep : 0;
    const uint32_t tilebytes = p.ds * (uint32_t)kTile8Rows;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t lane4 = (uint32_t)lane * 4u;

    // ---- DMA stream: as scan16_kernel (always issued; past the last tile num_records = 0 -> no memory touched).
    // The quantisation steps and residual bounds of a tile's halves travel in the same stream: one 16-byte operation, issued right before
    // the tile's first slot, so they have landed when that slot has (a scalar load per tile would expose an
    // HBM latency per tile; a vector load would put its own wait on the ring's vmcnt).  Every wave issues it
    // (same 16 bytes, same LDS address: the waits below stay wave-uniform arithmetic).
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t is_ti = 0;
    // The per-tile operations of tile i ride one tile AHEAD of its slots (with the first slot of tile i-1; tile 0's before everything):
    // they have landed -- and every wave has passed a barrier behind the wait that says so -- from position 0 of tile i-1 on, so the
    // centred form can read tile i's a_c while tile i-1 is still in flight.  Two operations per tile start, as before: the wait
    // schedule below does not change.
    auto tile_ops = [&](uint32_t it) {
        const uint32_t tile = t0 + it * tstep;
        const bool live_tile = it < nT;
        __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.tscale + kTscaleFloats * (size_t)tile), 0, live_tile ? (uint32_t)(kTscaleFloats * 4) : 0u, 0x00020000);
        char *sdst = smem + __builtin_amdgcn_readfirstlane(kRing16 * kSlot16Bytes + (it & (kScaleRing8 - 1)) * kScale8Entry);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_void *)sdst, 4, lane4, 0, 0, 0);
        // a_c of the tile's 64 rows: 256 bytes behind the scales, fetched by ONE wave of the eight.  The others, and every wave of a
        // plain copy, issue the same operation on an empty descriptor (one wait schedule, no memory read) -- into a dump area of
        // their own: an out-of-range LDS-DMA lane still WRITES (zeros), and must not land on the values another wave fetched
        const float *abase = centred ? p.amean + (size_t)kTile8Rows * tile : p.tscale;
        const bool mine = live_tile && centred && (int)(it & 7u) == wave;
        __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void *)abase, 0, mine ? (uint32_t)(kTile8Rows * 4) : 0u, 0x00020000);
        char *adst = mine ? sdst + 256 : smem + kRing16 * kSlot16Bytes + kScaleRing8 * kScale8Entry;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (lds_void *)adst, 4, lane4, 0, 0, 0);
    };
    auto open_tile = [&]() {
        const uint32_t tile = t0 + is_ti * tstep;
        const bool live_tile = is_ti < nT;
        tile_ops(is_ti + 1);
        const char *base = reinterpret_cast<const char *>(p.xh) + (size_t)tile * tilebytes;
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, live_tile ? tilebytes : 0u, 0x00020000);
        ++is_ti;
    };
    auto issue = [&](int kci, uint32_t ring_pos) {  // kci is a compile-time constant at every call site
        if (kci == 0) open_tile();
        char *dst = smem + __builtin_amdgcn_readfirstlane(ring_pos * kSlot16Bytes + wave * 1024);
        MX_LDS_DMA16(rsrc, dst, lane16, kci * kSlot16Bytes + wave * 1024, 2 /* nt */);
    };

    auto mylane = [&](int g) { return (uint32_t)((wave * QG + g) * 64 + lane) * gridDim.x + blockIdx.x; };
    uint32_t cnt[QG];  // records written; bit 31: a record did not fit
    float best[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) cnt[g] = 0, best[g] = -INFINITY;

    tile_ops(0);
#pragma unroll
    for (int i = 0; i < kRing16 - 1; ++i) issue(i % KC, (uint32_t)i);

    i32x4 a[R];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ops_after<KC>(1, kRing16 - 2)) : "memory");  // slot 0 (and its tile's scales) landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int f = 0; f < R; ++f) a[f] = *reinterpret_cast<const i32x4 *>(smem + lane16 + f * 1024);

    // ---- centred copy: init[u][r] = trunc(a_c[row] * (a_q / s_q) * (1 / s_h)), what the first MFMA of each half tile takes as C.
    // All LDS reads of scales and a_c are inline asm (hipcc puts s_waitcnt vmcnt(0) in front of a plain LDS load that it thinks an
    // LDS-DMA may have written, which would drain the ring once per tile).  Tile 0's values are computed here; tile i+1's behind
    // tile i's epilogue: ten reads issued in front of the epilogue, one wait and 16 packed multiplications + 32 conversions behind
    // it.  (Spreading that work over the tile's MFMA slots -- eight groups of 4 rows, each read a ring depth ahead of its use -- was
    // built and measured: 4.5 % SLOWER on the same box, profiles/r6_centred_int8_accumulator_init.txt.)
    f32x4 shs_c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // (s_h0, s_h1, e_h0, e_h1) of the tile in flight
    f32x4 shs_n = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // ... of the next one
    f32x4 inv_n = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // (1/s_h0, 1/s_h1, -, -) of the tile whose initial values are being computed
    f32x4 acv[CEN ? 8 : 1];                        // a_c of this lane's rows, group j = half * 4 + (r >> 2): [j][r & 3]
    i32x16 init[CEN ? 2 : 1];                      // [half tile]
    i32x16 acc[QG * 2];                            // [query group][half tile]
    auto entry_of = [&](uint32_t it) { return (uint32_t)(kRing16 * kSlot16Bytes) + (it & (kScaleRing8 - 1)) * (uint32_t)kScale8Entry; };
    auto group_values = [&](int j) {
        const float ku = kq * ((j >> 2) ? inv_n[1] : inv_n[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) init[j >> 2][(j & 3) * 4 + i] = (int)(acv[j][i] * ku);
    };
    if constexpr (CEN) {
        if (live) {
            const uint32_t sa = entry_of(0), aa = sa + 256 + (lane >> 5) * 16;
            asm volatile("ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:16\n\t"
                         "ds_read_b128 %2, %11\n\tds_read_b128 %3, %11 offset:32\n\tds_read_b128 %4, %11 offset:64\n\tds_read_b128 %5, %11 offset:96\n\t"
                         "ds_read_b128 %6, %11 offset:128\n\tds_read_b128 %7, %11 offset:160\n\tds_read_b128 %8, %11 offset:192\n\tds_read_b128 %9, %11 offset:224\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(shs_c), "=&v"(inv_n), "=&v"(acv[0]), "=&v"(acv[1]), "=&v"(acv[2]), "=&v"(acv[3]), "=&v"(acv[4]), "=&v"(acv[5]),
                           "=&v"(acv[6]), "=&v"(acv[7])
                         : "v"(sa), "v"(aa)
                         : "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) group_values(j);
        }
    }
    uint32_t rp = 0;
#pragma unroll 1
    for (uint32_t ti = 0; ti < nT; ++ti) {
        const uint32_t tile = t0 + ti * tstep;
        if constexpr (!CEN) {
#pragma unroll
            for (int g = 0; g < QG * 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0;
        }

        static_for<0, KC>([&](auto kct) __attribute__((always_inline)) {
            constexpr int kc = decltype(kct)::value;
            const uint32_t rp1 = (rp + 1) & (kRing16 - 1);
            const uint32_t rpi = (rp + kRing16 - 1) & (kRing16 - 1);
            // slot j+1 landed (the ring reads ahead into it): everything issued after it may still be in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ops_after<KC>(kc + 2, kc + kRing16 - 2)) : "memory");
            __builtin_amdgcn_s_barrier();  // ... for every wave; slot j-1 is free for slot j+15
            const uint32_t fb0 = rp * kSlot16Bytes + lane16, fb1 = rp1 * kSlot16Bytes + lane16;
            if (live) {
                static_for<0, 8>([&](auto ft) __attribute__((always_inline)) {
                    constexpr int f = decltype(ft)::value;
#pragma unroll
                    for (int g = 0; g < QG; ++g)
                        acc[g * 2 + (f & 1)] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[f % R], qf[g * KC * 4 + kc * 4 + (f >> 1)],
                                                                                      CEN && kc == 0 && f < 2 ? init[CEN ? (f & 1) : 0] : acc[g * 2 + (f & 1)], 0, 0, 0);
#if MX_SCAN8_ABLATE == 2  /* scripts/scan8_ubench.hip: what would HALF the fragment reads buy (every read feeding two MFMAs; results meaningless) */
                    if ((f & 1) == 0)
#endif
                    a[f % R] = *reinterpret_cast<const i32x4 *>(smem + (f + R < 8 ? fb0 : fb1) + ((f + R) & 7) * 1024);
                    if (f == 1) issue((kc + kRing16 - 1) % KC, rpi);
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
                issue((kc + kRing16 - 1) % KC, rpi);
            }
            rp = rp1;
        });

        if (!live) continue;
        // ---- tile epilogue, per half: lane holds query (vw*32 + m), rows (r&3) + 8*(r>>2) + 4*(lane>>5) of the half
        // (read with inline asm: hipcc puts s_waitcnt vmcnt(0) in front of a plain LDS load that it thinks an LDS-DMA
        // may have written, which would drain the ring once per tile; the scales landed with the tile's first slot)
        f32x4 shs;  // steps of the two halves, residual bounds of the two halves
#if MX_SCAN8_ABLATE == 1  /* scripts/scan8_ubench.hip: no scale read */
        shs = f32x4{1.0f, 1.0f, 0.0f, 0.0f};
#else
        if constexpr (CEN) {
            shs = shs_c;  // read with this tile's initial values, a tile ago
            {  // the next tile's scales and a_c: reads issued here, consumed behind this tile's epilogue
                const uint32_t sa = entry_of(ti + 1), aa = sa + 256 + (lane >> 5) * 16;
                asm volatile("ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:16\n\t"
                             "ds_read_b128 %2, %11\n\tds_read_b128 %3, %11 offset:32\n\tds_read_b128 %4, %11 offset:64\n\tds_read_b128 %5, %11 offset:96\n\t"
                             "ds_read_b128 %6, %11 offset:128\n\tds_read_b128 %7, %11 offset:160\n\tds_read_b128 %8, %11 offset:192\n\tds_read_b128 %9, %11 offset:224"
                             : "=&v"(shs_n), "=&v"(inv_n), "=&v"(acv[0]), "=&v"(acv[1]), "=&v"(acv[2]), "=&v"(acv[3]), "=&v"(acv[4]), "=&v"(acv[5]),
                               "=&v"(acv[6]), "=&v"(acv[7])
                             : "v"(sa), "v"(aa)
                             : "memory");
            }
        } else {
            const uint32_t sa = kRing16 * kSlot16Bytes + (ti & (kScaleRing8 - 1)) * kScale8Entry;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(shs) : "v"(sa) : "memory");
        }
#endif
#pragma unroll
        for (int g = 0; g < QG; ++g) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const i32x16 &ac = acc[g * 2 + u];
                const float sh = u ? shs[1] : shs[0], er = u ? shs[3] : shs[2];
                int mxi = max(max(ac[0], ac[1]), ac[2]);
#pragma unroll
                for (int r = 3; r < 15; r += 2) mxi = max(max(mxi, ac[r]), ac[r + 1]);
                mxi = max(mxi, ac[15]);
                // score of a sum: ((float)sum * s_h) * s_q -- monotone in the sum, so the test on the maximum is the
                // test on "any of the 16 scores" as finish_kernel will see them
                const float mx = ((float)mxi * sh) * sq[g];
                const float thr = fmaf(-qb[g], er, theta[g]);
                if (MODE == 0) {
                    best[g] = fmaxf(best[g], mx - fmaf(qb[g], er, qa[g]));  // only full tiles are sampled (index.hip): every row is a real row
                } else if (__builtin_amdgcn_ballot_w64(mx >= thr) != 0) {
                    if (mx >= thr) {
                        if ((cnt[g] & 0x7fffffffu) < (uint32_t)kRecCap) {
                            const size_t at = (size_t)mylane(g) * kRecCap + (cnt[g] & 0x7fffffffu);
                            f32x4 *dst = reinterpret_cast<f32x4 *>(p.lane_rec + at * 16);
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                dst[i] = f32x4{((float)ac[4 * i] * sh) * sq[g], ((float)ac[4 * i + 1] * sh) * sq[g],
                                               ((float)ac[4 * i + 2] * sh) * sq[g], ((float)ac[4 * i + 3] * sh) * sq[g]};
                            p.lane_tile[at] = 2 * tile + u;  // 32-row tile index, as finish_kernel counts them
                            ++cnt[g];
                        } else {
                            cnt[g] |= 0x80000000u;
                        }
                    }
                }
            }
        }
        if constexpr (CEN) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(shs_n), "+v"(inv_n), "+v"(acv[0]), "+v"(acv[1]), "+v"(acv[2]), "+v"(acv[3]), "+v"(acv[4]), "+v"(acv[5]), "+v"(acv[6]), "+v"(acv[7])
                         :
                         : "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) group_values(j);
            shs_c = shs_n;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // dead DMA ops must not outlive the workgroup's LDS

#pragma unroll
    for (int g = 0; g < QG; ++g) {
        if (MODE == 0) {
            p.lane_max[mylane(g)] = best[g];
        } else {
            p.lane_cnt[mylane(g)] = cnt[g] & 0x7fffffffu;
            if (cnt[g] >> 31) p.overflow[(wave * QG + g) * 32 + m] = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// filter-copy construction: one workgroup per 32-row half tile
// ---------------------------------------------------------------------------------------------
// Every row is normalised, ROTATED (mx_rotate.h: the quantiser then sees Gaussian-looking elements whatever the
// embedding model's spectrum is) and quantised.  A wave handles a row at a time (8 of the half tile's 32); the step
// needs the largest rotated element of all 32 rows, so the rows are rotated twice: once for the maximum, once to
// quantise (the arithmetic is nothing next to the row reads).  The int8 rows are staged in LDS row-major and copied
// out in fragment order by the whole workgroup.
// Centred form (mean != nullptr): with a_c = (c/|c|) . mean (f64 sum, stored as f32 in amean[row]) the quantiser sees
// r_c = c/|c| - a_c mean, several times shorter than the unit row for a corpus that sits in a cone -- and so are its step and
// its residual; rc_max tracks max |r_c| (the factor of the QUERY's residual in the row bound, prep_queries_kernel).
__global__ __launch_bounds__(256) void shadow8_kernel(const float *__restrict__ x, const float *__restrict__ scale, int ds,
                                                      uint32_t half0, uint32_t half1, uint64_t row_hi,
                                                      i32x4 *__restrict__ x8, float *__restrict__ tscale,
                                                      uint32_t *__restrict__ ec_max, const float *__restrict__ mean,
                                                      float *__restrict__ amean, uint32_t *__restrict__ rc_max) {
    extern __shared__ __attribute__((aligned(16))) char s8[];
    // [4 waves][2][ds] f32 rotation buffers | [32][ds] int8 staged rows | mix [144] | per-row residuals, reductions
    float *rbuf = reinterpret_cast<float *>(s8);
    int8_t *stage = reinterpret_cast<int8_t *>(s8 + (size_t)8 * ds * sizeof(float));
    float *mix = reinterpret_cast<float *>(s8 + (size_t)8 * ds * sizeof(float) + (size_t)kTileRows * ds);
    float *s_r2 = mix + kRotMaxBlocks * kRotMaxBlocks;  // [32]
    float *s_red = s_r2 + kTileRows;                    // [4]
    float *s_ac = s_red + 4;                            // [32] a_c of the half tile's rows (centred form)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kc = ds >> 7;
    const uint32_t frags = (uint32_t)(ds >> 5) * 64u;  // 16-byte fragments per half tile
    float *in = rbuf + (size_t)wave * 2 * ds, *out = in + ds;
    rot_fill_mix(mix, kc, tid, 256);
    __syncthreads();
    float worst = 0.0f, longest = 0.0f;
    for (uint32_t h = half0 + blockIdx.x; h < half1; h += gridDim.x) {
        const float *xt = x + (size_t)h * kTileRows * ds;
        // a row of the half tile, normalised (centred) and rotated, in `out` (zeros for a zero-norm row and for rows >= row_hi)
        auto rotated_row = [&](int r) -> bool {
            const uint64_t grow = (uint64_t)h * kTileRows + (uint64_t)r;
            const float sc = grow < row_hi ? scale[grow] : 0.0f;
            if (sc == 0.0f) {
                if (mean && lane == 0) amean[grow] = 0.0f, s_ac[r] = 0.0f;
                return false;
            }
            for (int i = lane; i < ds; i += 64) in[i] = xt[(size_t)r * ds + i] * sc;
            if (mean) {
                double dp = 0.0;
                for (int i = lane; i < ds; i += 64) dp += (double)in[i] * (double)mean[i];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) dp += __shfl_xor(dp, o);
                const float ac = (float)dp;  // the f32 value the scan will use: r_c is what is left after THIS a_c
                float l2 = 0.0f;
                for (int i = lane; i < ds; i += 64) {
                    const float v = __builtin_fmaf(-ac, mean[i], in[i]);
                    in[i] = v;
                    l2 += v * v;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) l2 += __shfl_xor(l2, o);
                longest = fmaxf(longest, sqrtf(l2));
                if (lane == 0) amean[grow] = ac, s_ac[r] = ac;
            }
            rot_wave(in, out, kc, lane, mix);
            return true;
        };
        // ---- pass 1: largest |element| of the rotated rows
        float mx = 0.0f;
        for (int r = wave; r < kTileRows; r += 4)
            if (rotated_row(r))
                for (int i = lane; i < ds; i += 64) mx = fmaxf(mx, fabsf(out[i]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) s_red[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        // (centred form: the step stays at or above kMinStep8, so that the scan's a_q a_c / (s_h s_q) fits an int32 -- a floor
        // only a half tile of residuals shorter than 0.004 ever meets, and what it costs them is inside their measured residual)
        const float sh = mean ? fmaxf(mx / 127.0f, kMinStep8) : mx / 127.0f;
        const float inv = mean ? 1.0f / sh : (mx > 0.0f ? 127.0f / mx : 0.0f);
        // ---- pass 2: quantise, residual per row, int8 rows staged in LDS
        for (int r = wave; r < kTileRows; r += 4) {
            float r2 = 0.0f;
            if (rotated_row(r)) {
                for (int i = lane; i < ds; i += 64) {
                    const float v = out[i];
                    const float qv = fminf(fmaxf(rintf(v * inv), -127.0f), 127.0f);
                    const float dlt = v - qv * sh;
                    r2 += dlt * dlt;
                    stage[(size_t)r * ds + i] = (int8_t)(int)qv;
                }
            } else {
                for (int i = lane; i < ds; i += 64) stage[(size_t)r * ds + i] = 0;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r2 += __shfl_xor(r2, o);
            if (lane == 0) s_r2[r] = r2;
        }
        __syncthreads();
        // ---- copy-out in fragment order
        const uint32_t T = h >> 1, u = h & 1;
        for (uint32_t f = tid; f < frags; f += 256) {
            const uint32_t ks = f >> 6, l = f & 63, mm = l & 31, hh = l >> 5;  // k-step of 32 dims
            const uint32_t sl = ks >> 2, j = ks & 3;
            x8[(((size_t)T * kc + sl) * 8 + (j * 2 + u)) * 64 + l] =
                *reinterpret_cast<const i32x4 *>(stage + (size_t)mm * ds + ks * 32 + hh * 16);
        }
        float hw = tid < kTileRows ? sqrtf(s_r2[tid]) : 0.0f;  // worst residual of THIS half tile (waves 1-3 hold zeros)
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) hw = fmaxf(hw, __shfl_xor(hw, o));
        if (tid == 0) {
            tscale[kTscaleFloats * (size_t)T + u] = sh;
            tscale[kTscaleFloats * (size_t)T + 2 + u] = hw * 1.01f + 1e-6f;
            tscale[kTscaleFloats * (size_t)T + 4 + u] = mean ? 1.0f / sh : 0.0f;  // what the scan multiplies a_c (a_q / s_q) by
            tscale[kTscaleFloats * (size_t)T + 6 + u] = 0.0f;
        }
        if (tid < kTileRows) worst = fmaxf(worst, hw);
        __syncthreads();
    }
    if (tid < kTileRows) {
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) worst = fmaxf(worst, __shfl_xor(worst, o));
        if (tid == 0 && worst > 0.0f) atomicMax(ec_max, __float_as_uint(worst));
    }
    if (mean && rc_max && lane == 0 && longest > 0.0f) atomicMax(rc_max, __float_as_uint(longest));  // (every lane of a wave holds its wave's maximum)
}

static size_t shadow8_lds(int ds) {
    return (size_t)8 * ds * sizeof(float) + (size_t)kTileRows * ds + (kRotMaxBlocks * kRotMaxBlocks + 2 * kTileRows + 4) * sizeof(float);
}

hipError_t launch_shadow8(hipStream_t s, const float *x, const float *scale, int ds, uint32_t half0, uint32_t half1,
                          uint64_t row_hi, void *x8, float *tscale, uint32_t *ec_max, const float *mean, float *amean,
                          uint32_t *rc_max) {
    if (half1 <= half0) return hipSuccess;
    if (mean && (!amean || !rc_max)) return hipErrorInvalidValue;
    const uint32_t blocks = half1 - half0 < 16384u ? half1 - half0 : 16384u;
    hipLaunchKernelGGL(shadow8_kernel, dim3(blocks), dim3(256), shadow8_lds(ds), s, x, scale, ds, half0, half1, row_hi,
                       reinterpret_cast<i32x4 *>(x8), tscale, ec_max, mean, amean, rc_max);
    return hipGetLastError();
}

template <int KC, int MODE, int QG, bool CEN = false>
static hipError_t setup8_one() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&scan8_kernel<KC, MODE, QG, CEN>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kScan8LdsBytes);
}

hipError_t scan8_setup() {
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void *>(&shadow8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)shadow8_lds(kMaxKC16 * kChunkFloats))) != hipSuccess)
        return e;
#define MX_SETUP(KC)                                             \
    if ((e = setup8_one<KC, 0, 1>()) != hipSuccess) return e;    \
    if ((e = setup8_one<KC, 1, 1>()) != hipSuccess) return e;
    MX_SETUP(1) MX_SETUP(2) MX_SETUP(3) MX_SETUP(4) MX_SETUP(5) MX_SETUP(6)
    MX_SETUP(7) MX_SETUP(8) MX_SETUP(9) MX_SETUP(10) MX_SETUP(11) MX_SETUP(12)
#undef MX_SETUP
#define MX_SETUP(KC)                                             \
    if ((e = setup8_one<KC, 0, 2>()) != hipSuccess) return e;    \
    if ((e = setup8_one<KC, 1, 2>()) != hipSuccess) return e;
    MX_SETUP(1) MX_SETUP(2) MX_SETUP(3) MX_SETUP(4)
#undef MX_SETUP
#define MX_SETUP(KC)                                                   \
    if ((e = setup8_one<KC, 0, 1, true>()) != hipSuccess) return e;    \
    if ((e = setup8_one<KC, 1, 1, true>()) != hipSuccess) return e;
    MX_SETUP(1) MX_SETUP(2) MX_SETUP(3) MX_SETUP(4) MX_SETUP(5) MX_SETUP(6)  // the centred forms: up to kMaxKC slots
#undef MX_SETUP
    static_assert(kMaxKC == 6, "one centred scan8_kernel per slot count up to kMaxKC");
    return hipSuccess;
}

template <int KC, int QG, bool CEN = false>
static hipError_t launch8_kc(hipStream_t s, bool collect, int nwg, const ScanParams &p) {
    if (collect)
        hipLaunchKernelGGL((scan8_kernel<KC, 1, QG, CEN>), dim3(nwg), dim3(kScanThreads), kScan8LdsBytes, s, p);
    else
        hipLaunchKernelGGL((scan8_kernel<KC, 0, QG, CEN>), dim3(nwg), dim3(kScanThreads), kScan8LdsBytes, s, p);
    return hipGetLastError();
}

hipError_t launch_scan8(hipStream_t s, int kc, bool collect, int nwg, const ScanParams &p, bool two_groups) {
    if (two_groups) {  // 512 queries per pass: up to 512 dims (kMaxKC8x2)
        if (p.amean) return hipErrorInvalidValue;
        switch (kc) {
            case 1: return launch8_kc<1, 2>(s, collect, nwg, p);
            case 2: return launch8_kc<2, 2>(s, collect, nwg, p);
            case 3: return launch8_kc<3, 2>(s, collect, nwg, p);
            case 4: return launch8_kc<4, 2>(s, collect, nwg, p);
            default: return hipErrorInvalidValue;
        }
    }
    if (p.amean) {  // centred copy: one query group, up to kMaxKC slots (index.hip builds no other)
        switch (kc) {
            case 1: return launch8_kc<1, 1, true>(s, collect, nwg, p);
            case 2: return launch8_kc<2, 1, true>(s, collect, nwg, p);
            case 3: return launch8_kc<3, 1, true>(s, collect, nwg, p);
            case 4: return launch8_kc<4, 1, true>(s, collect, nwg, p);
            case 5: return launch8_kc<5, 1, true>(s, collect, nwg, p);
            case 6: return launch8_kc<6, 1, true>(s, collect, nwg, p);
            default: return hipErrorInvalidValue;
        }
    }
    switch (kc) {
        case 1: return launch8_kc<1, 1>(s, collect, nwg, p);
        case 2: return launch8_kc<2, 1>(s, collect, nwg, p);
        case 3: return launch8_kc<3, 1>(s, collect, nwg, p);
        case 4: return launch8_kc<4, 1>(s, collect, nwg, p);
        case 5: return launch8_kc<5, 1>(s, collect, nwg, p);
        case 6: return launch8_kc<6, 1>(s, collect, nwg, p);
        case 7: return launch8_kc<7, 1>(s, collect, nwg, p);
        case 8: return launch8_kc<8, 1>(s, collect, nwg, p);
        case 9: return launch8_kc<9, 1>(s, collect, nwg, p);
        case 10: return launch8_kc<10, 1>(s, collect, nwg, p);
        case 11: return launch8_kc<11, 1>(s, collect, nwg, p);
        case 12: return launch8_kc<12, 1>(s, collect, nwg, p);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mx
