// index_kernels.h -- device-side contract of the flat cosine index (launch wrappers + shared
// constants).  Kernels live in scan.hip (the streaming scan) and index_kernels.hip (everything
// else); index.hip holds the host logic behind the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mx {

// ---- geometry of the streaming scan (scan.hip) --------------------------------------------
constexpr int kScanThreads = 512;  // 8 waves: 2 per SIMD
constexpr int kScanWaves = 8;
constexpr int kTileRows = 32;      // one 32x32x16 MFMA A-operand worth of corpus rows
constexpr int kChunkFloats = 128;  // k-chunk: 32 rows x 128 f32 = one 16 KiB LDS slot
constexpr int kSlotBytes = kTileRows * kChunkFloats * 4;
constexpr int kNumSlots = 8;       // f32 LDS ring: 8 x 16 KiB = 128 KiB, all of it in flight
constexpr int kPrefetch = 8;
constexpr int kScaleRing = 16;     // per-tile 1/|c| vectors (128 B each)
// ring + two padded bf16 tiles (32 rows x 272 B) + scale ring
constexpr int kScanLdsBytes = kNumSlots * kSlotBytes + 2 * kTileRows * (kChunkFloats * 2 + 16) + kScaleRing * kTileRows * 4;
constexpr int kMaxKC = 6;          // k-chunks per row the MFMA scan supports (dim_pad <= 768)

// ---- scan over the bf16 filter copy (scan16_kernel): a slot is 32 rows x 128 bf16 = 8 KiB, already
// in MFMA A-fragment order in HBM, so the LDS image is a linear copy and needs no conversion pass
constexpr int kSlot16Bytes = kTileRows * kChunkFloats * 2;
constexpr int kRing16 = 16;        // 16 x 8 KiB = 128 KiB ring, 15 slots in flight
constexpr int kMeanRing16 = 32;    // tiles whose a_c values (centred copy: 128 B per tile in a 256-B entry) can be in flight
constexpr int kScan16LdsBytes = kRing16 * kSlot16Bytes + kMeanRing16 * 256 + 1024;  // slot ring | a_c ring | a_q of the 256 queries
// wide rows (scan16w_kernel, 768 < dim_pad <= 1536): the k-steps of a row are dealt to two waves, 128 queries per
// launch; the ring plus 16 KiB in which the even-slot waves hand their partial sums to the odd-slot waves
constexpr int kMaxKC16 = 12;
constexpr int kWideBatch = 128;
constexpr int kScan16WideLdsBytes = kRing16 * kSlot16Bytes + 4 * 64 * 16 * 4;

constexpr int kPassBatch = 256;    // queries per scan pass (8 waves x 32 MFMA columns)
constexpr int kMaxBatch = 512;     // queries per pipeline batch: a pass of the int8 scan with two query groups per wave (up to
                                   // 512 dims) serves 512; everything else splits a batch into passes of 256 (128: scan16w)
constexpr int kMaxKC8x2 = 4;       // ... up to this many 128-dim slots per row
constexpr int kRecCap = 64;        // lane-private records per collect launch: one record = the lane's 16 scores of a tile
constexpr int kMaxScanWGs = 256;   // persistent workgroups (<= CUs)
constexpr int kCandCap = 16384;    // candidates finish_kernel holds per query (LDS); more = rescan with a tight threshold
constexpr int kZeroCap = 1024;     // zero-norm rows an index tracks in its list (more: EXACT path)
constexpr int kWildCap = 64;       // rows with a norm outside [1e-15, 1e15] an f32 index lists (more: EXACT path)

// cosine-unit bounds on |approx - exact| of the bf16 scan.
//   a priori:  two bf16 roundings (unit roundoff 2^-8 each) of unit vectors, Cauchy-Schwarz:
//              2^-7 + 2^-16, + kAccSlack for f32 accumulation/normalisation  ->  kApproxErr
//   per query: e1 = min(kApproxErr, Ec + Eq + Ec*Eq + kAccSlack) with the MEASURED residual norms
//              Ec = max_rows |bf16(c/|c|) - c/|c|| (tracked while the filter copy is built) and
//              Eq = |bf16(q/|q|) - q/|q|| (prep_queries_kernel): typically 0.0045.
constexpr float kApproxErr = 0.0081f;
constexpr float kAccSlack = 2.7e-4f;

struct Cand {
    float score;   // approximate cosine
    uint32_t row;  // local row
};

// local row -> id reported to the caller.  Default (block_rows = 0): id_offset + row + 1, the
// reference's dense 1-based insertion ids (local.rs:63).  A shard of a block-cyclic sharded index
// (shard g of G, blocks of block_rows rows) owns global rows ((row / R) * G + g) * R + row % R.
struct IdMap {
    uint64_t id_offset;
    uint32_t block_rows, n_shards, shard;
    __host__ __device__ uint64_t id_of(uint32_t row) const {
        const uint64_t r = block_rows ? ((uint64_t)(row / block_rows) * n_shards + shard) * block_rows + row % block_rows
                                      : (uint64_t)row;
        return id_offset + r + 1;
    }
};

struct ScanParams {
    const float *x;          // [cap_rows, ds] f32 corpus (ds = padded dim, multiple of 128)
    const void *xh;          // bf16 filter copy in fragment order (see launch_shadow); scan16 only
    const float *scale;      // [cap_rows] 1/|c| (f32; +inf for zero rows)
    const void *qfrag;       // bf16 query fragments [8 waves][ds/16][64 lanes][8]
    const float *theta;      // [256] pass threshold per query (cosine units)
    uint64_t n_rows;         // valid rows
    uint32_t tile_begin;     // workgroup b handles tiles tile_begin + (b + i*grid)*tile_stride < tile_end
    uint32_t tile_end;
    uint32_t tile_stride;    // 1 = every tile; > 1 = the evenly spread sample
    uint32_t ds;             // floats per stored row
    uint32_t wave_mask = 0xff;  // bit w: wave w (queries 32w .. 32w+31) has a query that is wanted; the others skip their MFMAs
                                // (two query groups per wave: one bit per group of 32, 16 bits)
    // collect launch: a lane whose 16 scores of a tile contain one >= theta stores ALL 16 (one record =
    // 64 bytes + the tile index); finish_kernel picks the passing rows.  No per-row code on the stream.
    float *lane_rec;         // [512][nwg][kRecCap][16]  (thread-in-workgroup major; [1024] with two query groups per wave)
    uint32_t *lane_tile;     // [512][nwg][kRecCap] tile index of each record
    uint32_t *lane_cnt;      // [512][nwg] records written
    float *lane_max;         // [512][nwg] sample mode: running maximum of each lane
    uint32_t *overflow;      // [256]
    // 8-bit filter copy (scan8_kernel): xh holds int8 fragments, tiles are 64 rows
    const float *tscale = nullptr;  // [cap_rows / 64][kTscaleFloats]: quantisation steps of a tile's two halves, their residual bounds, 1 / step of a centred copy
    const float *qscale = nullptr;  // [256] quantisation step of each query
    const float *qa = nullptr, *qb = nullptr;  // [256] a row's bound is qa + qb * residual (launch_prep_queries)
    // centred bf16 copy (scan16_kernel, launch_shadow): the copy holds r_c = c/|c| - a_c m for a fixed unit direction m (the
    // corpus mean direction), amean[row] = a_c; the query fragments hold r_q = q/|q| - a_q m, qmean[q] = a_q; a row's score
    // is a_q a_c + (MFMA sum over r_q r_c).  null: the copy holds c/|c| itself.
    const float *amean = nullptr;   // [cap_rows]
    const float *qmean = nullptr;   // [256]
};

// launches ---------------------------------------------------------------------------------
hipError_t scan_setup();  // one-time function attributes (dynamic LDS size)
hipError_t scan16_setup();
// collect = false: sample launch (lane maxima only); collect = true: survivors of theta are appended
hipError_t launch_scan(hipStream_t s, int kc, bool collect, int nwg, const ScanParams &p);
hipError_t launch_scan16(hipStream_t s, int kc, bool collect, int nwg, const ScanParams &p);
// scan over the 8-bit filter copy: kc = ds / 128 in 1 .. 12, tile_begin / tile_end / tile_stride count 64-row tiles
constexpr int kTile8Rows = 64;
constexpr int kTscaleFloats = 8;  // per 64-row tile: steps of its two halves | residual bounds | 1 / step (centred copy, else 0) | 0
constexpr float kMinStep8 = 1.0f / 32768.0f;  // smallest quantisation step of a CENTRED int8 copy and of its queries (scan8.hip: |a_q a_c / (s_h s_q)| < 2^30)
constexpr int kScaleRing8 = 32;  // tiles whose scales can be in flight (15 slots ahead at one slot per tile, + the tile being multiplied)
constexpr int kScale8Entry = 512;  // per tile: [0, 16) the steps and residual bounds of its halves | [256, 512) a_c of its 64 rows (centred copy)
constexpr int kScan8LdsBytes = kRing16 * kSlot16Bytes + kScaleRing8 * kScale8Entry + 256;  // ... | 256 B that absorb the empty a_c operations
hipError_t scan8_setup();
hipError_t launch_scan8(hipStream_t s, int kc, bool collect, int nwg, const ScanParams &p, bool two_groups = false);
// (re)build half tiles [half0, half1) (32 rows each) of the 8-bit filter copy from the padded f32 store: per half
// tile one quantisation step = max |c_i/|c|| / 127 over its rows below row_hi (rows at or above row_hi, and zero-norm
// rows, are stored as zeros) and one residual bound = 1.01 * max_rows |c/|c| - step * c8| + 1e-6; both go to
// tscale[kTscaleFloats * (h / 2) + (h & 1)] and [.. + 2] (then 1 / step of the halves of a centred copy): the 32 bytes of a 64-row scan tile;
// ec_max as launch_shadow.
// Centred form (mean != nullptr, f32 corpora up to kMaxKC slots): the quantiser sees r_c = c/|c| - a_c mean, amean[row] = a_c
// (f32; zero for the zero rows), steps and residual bounds are those of the shorter vectors, rc_max (device word, atomicMax'ed
// float bits) = max |r_c|
hipError_t launch_shadow8(hipStream_t s, const float *x, const float *scale, int ds, uint32_t half0, uint32_t half1,
                          uint64_t row_hi, void *x8, float *tscale, uint32_t *ec_max, const float *mean = nullptr,
                          float *amean = nullptr, uint32_t *rc_max = nullptr);
hipError_t scan16w_setup();
hipError_t launch_scan16w(hipStream_t s, int kc, bool collect, int nwg, const ScanParams &p);  // kc in {8, 10, 12}, <= 128 queries

// (re)build tiles [tile0, tile1) of the bf16 filter copy from the padded f32 store.  Layout: tile t
// (32 rows), k-step s (16 dims), MFMA lane l -> 8 bf16 at ((t*(ds/16) + s)*64 + l)*8, holding row
// 32t + (l&31), dims 16s + 8(l>>5) .. +7: exactly the A operand of v_mfma_f32_32x32x16_bf16.
// Values are bf16(c_i * 1/|c|); a zero-norm row is stored as zeros (it scores 0 in the scan and reaches
// finish_kernel through the index's zero-row list instead).
// ec_max: device word, atomicMax'ed with the float bits of the largest |bf16(c/|c|) - c/|c|| built.
// src_tile0: x / scale hold the rows of tiles src_tile0 .. (a staging window; 0 = the whole store).
// Only rows in [row_lo, row_hi) are (re)written: a 16-byte fragment belongs to ONE row, so rows of a
// tile that are already in the copy stay untouched.
// Centred form (mean != nullptr; f32 corpora only): with a_c = (c/|c|) . mean the copy holds bf16(c/|c| - a_c mean) and
// amean[row] = a_c; ec_max then tracks |stored - (c/|c| - a_c mean)|.  Embedding corpora sit in a cone around a common
// direction: what is left after removing it is several times shorter than the unit vector, and so is its bf16 rounding
// error -- the scan's certificate.
hipError_t launch_shadow(hipStream_t s, const float *x, const float *scale, int ds, uint32_t tile0, uint32_t tile1,
                         void *xh, uint32_t *ec_max, uint32_t src_tile0 = 0, uint64_t row_lo = 0, uint64_t row_hi = ~0ull,
                         const float *mean = nullptr, float *amean = nullptr);
// mean[ds] = normalised sum of c/|c| over rows [0, n) (zeros when the sum vanishes); msum: [ds + 1] f32 scratch, on return
// msum[ds] = |sum of c/|c||
hipError_t launch_mean_dir(hipStream_t s, const float *x, const float *scale, uint64_t n, int ds, float *msum, float *mean);
// compressed corpus -> f32 rows [n, d]
hipError_t launch_unshadow(hipStream_t s, const void *xh, int ds, int d, uint64_t row0, uint64_t n, float *out);

// rows [n, d] (device) -> x[first.., ds] zero-padded + scale (0 for a zero-norm row); flags[0] += non-finite
// rows, flags[1] += rows whose norm is outside the range the bf16 scan is certified for, flags[3] +=
// zero-norm rows, whose local row numbers (row_base + r) go to zero_rows[] (first kZeroCap); raw & 2: flags[4] += rows
// with such a norm, stored as zeros and listed in wild_rows[] (first kWildCap)
hipError_t launch_ingest(hipStream_t s, const float *src, uint64_t n, int d, float *x, float *scale,
                         uint64_t first, int ds, uint32_t *flags, int raw, uint32_t *zero_rows, uint32_t *wild_rows, uint64_t row_base);

// queries [B, d] (device) -> qfrag (normalised bf16 fragments), qpad [256, ds] f32 original
// values zero padded, qnorm2 [256] f64 (sequential DistCosine accumulation), theta init, e1 [256]
// per-query error bound of the scan (ec_max: device word holding the filter copy's largest row
// residual as float bits, or null = a-priori bound); flags[0] |= 1 when a query is not finite
hipError_t launch_prep_queries(hipStream_t s, const float *q, int B, int d, int ds, void *qfrag,
                               float *qpad, double *qnorm2, float *theta, float *e1, const uint32_t *ec_max,
                               uint32_t *overflow, uint32_t *flags, float *qa, float *qb, bool filt8 = false,
                               float *qscale = nullptr, const float *mean = nullptr, float *qmean = nullptr,
                               const uint32_t *rc_max = nullptr);
// mean / qmean: the centred copy (ScanParams::amean; bf16 or int8): fragments of q/|q| - a_q mean, qmean[q] = a_q;
// rc_max (int8 copy): device word with max |r_c| over the rows as float bits (launch_shadow8)
// qa / qb [256]: the bound of one row's filter score is qa + qb * (residual of the row's half tile); scans that know
// one residual for all rows get qa = e1, qb = 0
// filt8: fragments for the 8-bit filter copy (scan8.hip: int8 [8 waves][ds/32][64 lanes][16]) and qscale[256] = the
// query's quantisation step (0 for an unusable query); ec_max then is that copy's residual word (required)

// theta[q] = (k-th largest of query q's lane maxima) - 2*e1[q]
// raw: the lane maxima are plain scores (theta = k-th - 2 * qa); otherwise they are lower bounds of cosines already
// (scan8_kernel: theta = k-th - qa)
hipError_t launch_theta(hipStream_t s, int B, int k, int nwg, const float *lane_max, const float *qa, bool raw, float *theta);

// per query: gather the collect launch's lane buffers -> keep [kth approx - 2*e1, inf) -> f32
// rescoring (error e2) -> keep [kth - 2*e2, inf) -> exact DistCosine -> order by (dist, id) -> emit.
// overflow[q] (in: 1 = a lane buffer overflowed) out: 0 = answered, 1 = rescan with theta_retry[q],
// >= 2 = answer on the EXACT path.
struct FinishParams {
    int k, ds, nwg;
    const float *x;             // [cap_rows, ds] f32 rows; null = compressed corpus (rows are read from xh)
    const void *xh;             // bf16 filter copy (fragment order)
    const float *scale;         // [cap_rows] 1/|c| (f32 rows only)
    uint64_t n_rows;
    IdMap idmap;
    const float *qpad;          // [256, ds]
    const double *qnorm2;       // [256]
    const float *e1;            // [256]
    const float *qa, *qb;       // [256] bound of a row's filter score: qa + qb * residual(row)
    const float *terr;          // the 8-bit copy's tscale array (residual of row r: [kTscaleFloats * (r / 64) + 2 + (r / 32 & 1)]); null: 0
    float e2;                   // bound on |f32 rescoring - cosine|
    const float *lane_rec;      // records of the collect launch (ScanParams)
    const uint32_t *lane_tile;
    const uint32_t *lane_cnt;
    const float *theta;         // [256] the collect launch's pass threshold
    const uint32_t *zero_rows;  // [kZeroCap] zero-norm rows of the index (their stored scores are 0: they enter here)
    uint32_t n_zero;
    const uint32_t *wild_rows;  // [kWildCap] rows with a norm outside the f32 stages' range, ascending: stage 3 takes them all
    uint32_t n_wild;
    uint32_t *overflow;         // [256]
    const uint32_t *todo;       // null = every query; else only queries with todo[q] != 0
    float *theta_retry;         // [256]
    uint32_t *cand_cnt;         // [256] candidates rescored in f32 (statistics)
    uint64_t *ids;
    float *scores, *dists;
    int32_t *n_found;
    float *max_err;             // null unless profiling
    // completion signal: the LAST workgroup of the launch writes a 4-word summary of the batch's flag block
    // (dev_flags: [overflow 256 | cand_cnt 256 | e1 256 | qbad 256]) into host-mapped memory and then stores
    // seq behind it (system scope) -- the host polls that word instead of queueing a D2H copy (and a flag memset) and waiting in
    // hipStreamSynchronize.  host_flags == nullptr: no signal.
    uint32_t *done_ctr;         // device word, 0 between launches
    const uint32_t *dev_flags;
    uint32_t *host_flags;       // [5]: max overflow code, sum of cand_cnt, any bad query, e1[0], seq
    int n_queries;              // B
    int host_out;               // ids / scores / dists / n_found are pinned host memory (mx_index_search): system-scope fence before the tick
    uint32_t seq;
};
hipError_t finish_setup();
hipError_t launch_finish(hipStream_t s, int B, const FinishParams &p);
hipError_t launch_retry_setup(hipStream_t s, float *theta, const float *theta_retry, uint32_t *overflow, uint32_t *todo);

// EXACT path, batched: a group of up to kExactGroup queries against every row in one pass (f32 products, sequential f64
// sums per pair: DistCosine), then a 3-pass radix select per query with ties ordered by row.  Queries are named by
// their slot in qpad / qnorm2 / the output arrays; scratch holds exact_group_scratch_bytes(n_rows, k).
constexpr int kExactGroup = 32;
constexpr int kExactSlices = 256;  // row slices per query in the selection passes
struct ExactGroup {
    int n;
    int q[kExactGroup];
};
size_t exact_group_scratch_bytes(uint64_t n_rows, int k, int gcap);  // gcap: queries per pass the dist array holds (<= kExactGroup)
hipError_t launch_exact_group(hipStream_t s, int k, int ds, const float *x, const void *xh, uint64_t n_rows, const IdMap &idmap,
                              const float *qpad, const double *qnorm2, const ExactGroup &grp, void *scratch, uint64_t *ids,
                              float *scores, float *dists, int32_t *n_found);

hipError_t launch_fill_nfound(hipStream_t s, int32_t *nf, int B, int32_t v);
hipError_t launch_merge(hipStream_t s, const void *ids, size_t ids_stride, const void *dists, size_t dists_stride,
                        int G, int B, int k, uint64_t *out_ids, float *out_dists, float *out_scores);

}  // namespace mx
