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
constexpr int kScan16LdsBytes = kRing16 * kSlot16Bytes;

constexpr int kMaxBatch = 256;     // queries per scan pass (8 waves x 32 MFMA columns)
constexpr int kLaneCap = 32;       // lane-private candidate slots per scan launch
constexpr int kMaxScanWGs = 256;   // persistent workgroups (<= CUs); pool sizing depends on it
constexpr int kPoolCap = kLaneCap * kMaxScanWGs;  // per-query candidate pool (entries)
constexpr int kFinalCap = 1024;    // candidates exactly rescored per query before falling back

// cosine-unit bound on |approx - exact| of the bf16 scan: two bf16 roundings (2^-8 each) on
// |q||c|-normalised products (<= 2^-7 + 2^-16 by Cauchy-Schwarz) + f32 accumulation/normalisation.
constexpr float kApproxErr = 0.0081f;
constexpr float kMargin = 2.0f * kApproxErr + 1e-4f;

struct Cand {
    float score;   // approximate cosine (NaN / +2 = "zero-norm row": exact dist is 0)
    uint32_t row;  // local row
};

struct ScanParams {
    const float *x;          // [cap_rows, ds] f32 corpus (ds = padded dim, multiple of 128)
    const void *xh;          // bf16 filter copy in fragment order (see launch_shadow); scan16 only
    const float *scale;      // [cap_rows] 1/|c| (f32; +inf for zero rows)
    const void *qfrag;       // bf16 query fragments [8 waves][ds/16][64 lanes][8]
    const float *theta;      // [256] pass threshold per query (cosine units)
    uint64_t n_rows;         // valid rows
    uint32_t tile_begin;     // tile range of this stage
    uint32_t tile_end;
    uint32_t ds;             // floats per stored row
    Cand *lane_buf;          // [512][nwg][kLaneCap]  (thread-in-workgroup major)
    uint32_t *lane_cnt;      // [512][nwg]
    uint32_t *overflow;      // [256]
};

// launches ---------------------------------------------------------------------------------
hipError_t scan_setup();  // one-time function attributes (dynamic LDS size)
hipError_t scan16_setup();
hipError_t launch_scan(hipStream_t s, int kc, bool main_stage, int nwg, const ScanParams &p);
hipError_t launch_scan16(hipStream_t s, int kc, bool main_stage, int nwg, const ScanParams &p);

// (re)build tiles [tile0, tile1) of the bf16 filter copy from the padded f32 store.  Layout: tile t
// (32 rows), k-step s (16 dims), MFMA lane l -> 8 bf16 at ((t*(ds/16) + s)*64 + l)*8, holding row
// 32t + (l&31), dims 16s + 8(l>>5) .. +7: exactly the A operand of v_mfma_f32_32x32x16_bf16.
// Values are bf16(c_i * 1/|c|) (NaN for a zero-norm row: it must pass every filter).
hipError_t launch_shadow(hipStream_t s, const float *x, const float *scale, int ds, uint32_t tile0, uint32_t tile1,
                         void *xh);

// rows [n, d] (device) -> x[first.., ds] zero-padded + scale; flags[0] += non-finite rows,
// flags[1] += rows whose norm is outside the range the bf16 scan is certified for
hipError_t launch_ingest(hipStream_t s, const float *src, uint64_t n, int d, float *x, float *scale,
                         uint64_t first, int ds, uint32_t *flags);

// queries [B, d] (device) -> qfrag (normalised bf16 fragments), qpad [256, ds] f32 original
// values zero padded, qnorm2 [256] f64 (sequential DistCosine accumulation), theta init
hipError_t launch_prep_queries(hipStream_t s, const float *q, int B, int d, int ds, void *qfrag,
                               float *qpad, double *qnorm2, float *theta, uint32_t *overflow,
                               uint32_t *pool_cnt);

// gather lane buffers of one scan stage into the per-query pool, select the k-th best approximate
// score, prune the pool to [kth - margin, +inf) and publish theta = kth - margin
hipError_t launch_update(hipStream_t s, int B, int k, int nwg, const Cand *lane_buf,
                         const uint32_t *lane_cnt, Cand *pool_in, Cand *pool_out, uint32_t *pool_cnt,
                         float *theta, uint32_t *overflow);

// exact DistCosine rescoring of the pool + ordering by (dist, id) + outputs
hipError_t launch_final(hipStream_t s, int B, int k, int d, int ds, const float *x, uint64_t n_rows,
                        uint64_t id_offset, const float *qpad, const double *qnorm2, const Cand *pool,
                        const uint32_t *pool_cnt, uint32_t *overflow, uint64_t *ids, float *scores,
                        float *dists, int32_t *n_found, float *max_err);

// EXACT path: one query against every row in f64, then a 64-step radix select on (dist,row) keys
hipError_t launch_exact_query(hipStream_t s, int k, int d, int ds, const float *x, uint64_t n_rows,
                              uint64_t id_offset, const float *qpad_row, uint64_t *keys,
                              uint64_t *sel_state, uint64_t *ids, float *scores, float *dists,
                              int32_t *n_found);

hipError_t launch_merge(hipStream_t s, const void *ids, size_t ids_stride, const void *dists, size_t dists_stride,
                        int G, int B, int k, uint64_t *out_ids, float *out_dists, float *out_scores);

}  // namespace mx
