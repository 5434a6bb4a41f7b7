// index.hip -- host logic of the GPU-resident flat cosine index behind the C ABI
// (include/memex_hip.h).  Drop-in for memex's HnswStore (reference
// lib/libmemex/src/storage/local.rs:21-166) reached through the VectorStore trait
// (lib/libmemex/src/storage/mod.rs:55-66).
//
// Search pipeline per batch of <= 512 queries (one scan pass serves 256 of them -- 512 on the int8 copy up to 512
// dims, 128 on scan16w; kernels: scan8.hip / scan16.hip / scan16w.hip / scan.hip,
// index_kernels.hip):
//   prep      normalise queries -> MFMA fragments (int8 with the query's own step, or bf16); f64 query norms
//             (DistCosine order); the bound of a row's filter score, qa + qb * (residual of the row's half
//             tile), from measured residuals (one residual for all rows with the bf16 / f32 scans: qb = 0)
//   sample    scan an evenly spread 1/4 .. 1/64 of the tiles keeping only each lane's best lower bound
//   theta     k-th largest of those - qa = pass threshold (certified: keeps the exact top-k)
//   collect   scan every tile; a lane with a passing row stores its 16 scores as one record
//   finish    gather -> k-th best lower bound, keep the rows whose upper bound reaches it -> f32 rescoring
//             (error e2 ~ 5e-5) -> keep [kth - 2*e2, inf) -> exact f64 DistCosine -> order by (dist, id)
// Five launches; corpora of <= 2 tiles per workgroup skip sample/theta (theta = -inf).
// What the scan streams: an int8 or a bf16 filter copy kept next to the f32 rows (the library chooses by row
// width and demotes int8 to bf16 on a corpus too dense for its certificate), or the f32 rows themselves.
// A query whose lane buffers overflowed (dense neighbourhoods, weak sample threshold) is rescanned
// ONCE with the tight threshold finish derived from what it did collect (all such queries of the
// batch share that one extra pass); only if that overflows too -- more than ~16k rows within the bound of
// the k-th neighbour -- is it answered on the EXACT path (f64 on every row).
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <chrono>
#include <ctime>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "index_kernels.h"
#include "mx_common.h"
#include "mx_debug.h"
#include "shard_pool.h"

namespace mx {

std::string &last_error_slot() {
    static thread_local std::string slot;
    return slot;
}

namespace {

// per-query words of a batch in HBM: [overflow 256 | cand_cnt 256 | e1 256 | qbad 256].  finish_kernel sends
// the host a 4-word summary + its sequence number (host_sum); the block itself is only copied when the
// summary reports an overflow, and on the EXACT path
constexpr int kFlagWords = 4 * kMaxBatch;
// Filter copy an f32 corpus keeps unless told otherwise: int8 up to 1024 dims, bf16 above.  The int8 certificate is
// ~0.016-0.026 wide whatever the width (quantisation noise of a unit vector does not depend on its length) while the
// spread of cosines shrinks like 1/sqrt(dim), so the int8 pass hands more rows to finish_kernel the wider the rows are.
// Measured on Gaussian rows, B = 256, k = 10, one box (scripts/r3_dims.sh; int8 at its best sample size / bf16, kQPS):
// 463 / 312 at 128 dims, 290 / 188 at 256, 188 / 117 at 384, 160 / 100 at 512, 102 / 70 at 768, 146 / 90 at 1024
// (4M rows), 51 / 61 at 1536 (4M rows).
constexpr int kAutoI8MaxDim = 1024;  // filter copy of an f32 corpus: int8 (scan8.hip) or bf16 (scan16.hip)
constexpr int kSumWords = 5;  // max overflow code, candidates, any bad query, e1 of query 0, seq

struct Scratch {
    void *qfrag = nullptr;
    float *qpad = nullptr;
    double *qnorm2 = nullptr;
    float *theta = nullptr, *theta_retry = nullptr;
    uint32_t *todo = nullptr;
    uint32_t *dev_flags = nullptr;   // [kFlagWords]
    uint32_t *host_flags = nullptr;  // pinned mirror of dev_flags [kFlagWords] (D2H copy, rare)
    uint32_t *host_sum = nullptr;    // pinned, device-visible [kSumWords]: written by finish_kernel's last workgroup
    bool out_on_host = false;        // this batch's output pointers are mapped host memory (run_combined)
    uint32_t *done_ctr = nullptr;    // finish_kernel's workgroup counter
    uint32_t flag_seq = 0;           // sequence number of the last finish launch
    uint32_t *overflow = nullptr, *cand_cnt = nullptr, *qflags = nullptr;  // views into dev_flags
    float *e1 = nullptr;
    float *lane_rec = nullptr;       // records of the collect launch (ScanParams)
    uint32_t *lane_tile = nullptr;
    uint32_t *lane_cnt = nullptr;
    float *lane_max = nullptr;
    float *qstage = nullptr;       // [256, dim] host->device query staging
    float *qscale = nullptr;       // [256] quantisation step of each query (8-bit filter copy)
    float *qa = nullptr, *qb = nullptr;  // [256] a row's filter-score bound is qa + qb * residual (launch_prep_queries)
    float *qmean = nullptr;        // [256] a_q of the centred bf16 copy
    uint64_t *out_ids = nullptr;   // [256, kcap] device outputs for the host API
    float *out_scores = nullptr;
    float *out_dists = nullptr;
    int32_t *out_nfound = nullptr;
    int kcap = 0;
    void *exact_scratch = nullptr;  // EXACT path: distances of a query group, selection state (exact_group_scratch_bytes)
    size_t exact_bytes = 0;
    float *max_err = nullptr;
    // pinned staging of the host API: queries in, results out (one H2D / D2H per combined batch)
    float *h_q = nullptr;
    uint64_t *h_ids = nullptr;
    float *h_scores = nullptr, *h_dists = nullptr;
    int32_t *h_nf = nullptr;
    bool ready = false;
};

// The lane buffers of the scan (records of the collect launch: 64 per lane x 68 B, 0.57 GB per set at 256 workgroups x 512
// lanes; twice that for a 512-query pass of the int8 scan, which numbers 1024 lanes per workgroup) are
// only live inside one search_batch call, which returns host-synchronised.  They are therefore leased from a pool per
// device instead of owned by every index: a process with many resident collections holds as many sets as it has
// searches in flight on a device at the same time, not one per collection.  The pool is freed when the last index
// on the device closes.
struct LaneBufs {
    float *rec = nullptr;
    uint32_t *tile = nullptr, *cnt = nullptr;
    float *max = nullptr;
    int nwg = 0;
    int groups = 1;  // query groups per wave the set is sized for (2: a 512-query pass)
};
constexpr int kMaxDevices = 64;
struct LanePool {
    std::mutex mu;
    std::vector<LaneBufs> idle;
    int open_indexes = 0;
};
LanePool g_lanes[kMaxDevices];

void lane_free(LaneBufs &b) {
    if (b.rec) (void)hipFree(b.rec);
    if (b.tile) (void)hipFree(b.tile);
    if (b.cnt) (void)hipFree(b.cnt);
    if (b.max) (void)hipFree(b.max);
    b = LaneBufs{};
}

// RCCL entry points, resolved with dlopen the first time a sharded index spans more than one device
// (libmemex_hip.so itself links only the HIP runtime; a single-GPU host never loads RCCL)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

}  // namespace
}  // namespace mx

using namespace mx;

// one host-API search call waiting to be served (see mx_index_search)
struct SearchReq {
    const float *q;
    int B, k;
    uint64_t *ids;
    float *scores, *dists;
    int32_t *n_found;
    int rc = MX_OK;
    std::string err;
    bool done = false;
};

struct mx_index {
    std::string key;
    int dim = 0, ds = 0, kc = 0, device = 0;
    int refs = 1;
    std::mutex mu;
    // request combining (mx_index_search): callers queue here; one of them, the leader, serves batches
    std::mutex cmu;
    std::condition_variable ccv;
    std::deque<SearchReq *> pending;
    bool leader = false;
    hipStream_t stream = nullptr;
    float *x = nullptr;
    float *scale = nullptr;
    void *xh = nullptr;          // bf16 filter copy (fragment order), cap/32 tiles; null = not kept
    bool want_filter = true;     // keep a filter copy when HBM allows (mx_index_set_filter_copy)
    // 8-bit filter copy (scan8.hip): xh holds int8 fragments in 64-row tiles, tsc one quantisation step per 32-row
    // half tile.  An f32 corpus only; the compressed corpus keeps its bf16 rows.
    bool filter_i8 = false;
    bool filter_auto = true;     // the library picks the kind (by row width) and may demote int8 to bf16 when a batch overflows
    bool pooled = false;         // counted in its device's lane-buffer pool (open_plain)
    uint32_t i8_batches = 0, i8_retry_batches = 0;  // since the int8 copy was built: batches served, batches that needed the retry pass
    uint64_t plain_bf16_rows = 0;  // rows the index held when its bf16 copy was last built WITHOUT centring (0: no such copy): a collection
                                   // that started small, or off a cone, is looked at again once it has doubled (add_device_locked)
    uint64_t demoted_at_rows = 0;  // rows the index held when an automatic int8 copy was demoted to bf16 (0: never); the
                                   // int8 copy gets another try once the collection has doubled (add_device_locked)
    float *tsc = nullptr;
    // centred bf16 copy (f32 corpus, up to kMaxKC slots; launch_shadow): `amean` lives with every such copy, `centred` says
    // whether it was built around `mean` (a full rebuild of a populated index: a demotion, mx_index_set_filter_copy)
    float *amean = nullptr;      // [cap] a_c = (c/|c|) . mean
    float *mean = nullptr;       // [ds] unit direction; msum: [ds] scratch of its computation
    float *msum = nullptr;
    bool centred = false;
    // compressed corpus (mx_index_set_corpus_mode): xh is the ONLY copy of the rows; x / scale are not
    // allocated, appends pass through the small f32 staging window xs / ss
    bool compressed = false;
    bool raw_ingest = false;     // rows being appended are stored values coming back from disk: no renormalisation
    float *xs = nullptr, *ss = nullptr;
    uint64_t xs_rows = 0;
    uint64_t n = 0, cap = 0;
    IdMap idmap{0, 0, 1, 0};
    uint32_t *flags = nullptr;  // device: [0] non-finite rows, [1] out-of-range-norm rows (last add), [2] ec_max (float bits), [3] zero-norm rows, [4] listed out-of-range-norm rows,
                                // [5] rc_max (float bits): max |r_c| of a centred int8 copy
    uint64_t wild_rows = 0;         // rows with a norm outside [1e-15, 1e15] (a compressed corpus answers on the EXACT path then)
    uint32_t *zero_rows = nullptr;  // device: [kZeroCap] local rows with zero norm, ascending (finish_kernel adds them to every query)
    uint64_t n_zero = 0;            // zero-norm rows in the index; more than kZeroCap -> EXACT path
    uint32_t *wild_list = nullptr;  // device: [kWildCap] local rows with a norm outside [1e-15, 1e15] (f32 corpus), ascending
    uint64_t n_wild = 0;            // ... how many; more than kWildCap -> EXACT path
    int mode = MX_SEARCH_AUTO;
    bool profiling = false;
    int n_cu = 0, nwg = 0;
    Scratch s;
    mx_index_stats stats{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_wait = nullptr;
    double wait_ema_us[4] = {0.0, 0.0, 0.0, 0.0};  // how long recent batches of <= 32 / 128 / 256 / more queries took from the
                                                   // finish launch to completion (sleeping wait)
    // persistence bookkeeping: what vectors.mxflat in `disk_dir` holds, as far as this handle knows
    std::string disk_dir;
    uint64_t disk_rows = 0;
    off_t disk_size = 0;
    struct timespec disk_mtime {};
    // ---- composite (mx_index_open_sharded): rows are dealt to `shards` in blocks of block_rows
    std::vector<mx_index *> shards;
    uint64_t block_rows = 0;
    uint64_t total = 0;
    bool use_rccl = false;
    std::vector<void *> comms;       // ncclComm_t per shard (RCCL exchange)
    std::vector<void *> sh_block;    // per shard, on its device: packed result block [ids | dists]
    std::vector<void *> sh_gather;   // per shard, on its device: [G] packed blocks (RCCL recv); [0] is the merge input
    std::vector<float *> sh_q, sh_scores;
    std::vector<int32_t *> sh_nf;
    int sh_kcap = 0;
    std::unique_ptr<ShardPool> pool;  // helper threads for shards 1 .. G-1 (shards on distinct devices only)
    bool composite() const { return !shards.empty(); }
};

namespace {

std::mutex g_reg_mu;
std::map<std::string, mx_index *> g_registry;
std::once_flag g_scan_once;
hipError_t g_scan_setup_err = hipSuccess;
std::mutex g_rccl_mu;
Rccl g_rccl;
std::atomic<bool> g_rccl_broken{false};  // an initialisation hung: no further attempts in this process (indexes open concurrently)

bool load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return g_rccl.ok;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return false;
    auto sym = [&](const char *n) { return dlsym(g_rccl.lib, n); };
    g_rccl.CommInitAll = reinterpret_cast<int (*)(void **, int, const int *)>(sym("ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<int (*)(void *)>(sym("ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, hipStream_t)>(sym("ncclAllGather"));
    g_rccl.GroupStart = reinterpret_cast<int (*)()>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<int (*)()>(sym("ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<const char *(*)(int)>(sym("ncclGetErrorString"));
    g_rccl.ok = g_rccl.CommInitAll && g_rccl.CommDestroy && g_rccl.AllGather && g_rccl.GroupStart && g_rccl.GroupEnd;
    return g_rccl.ok;
}

void free_composite_buffers(mx_index *idx) {
    for (size_t g = 0; g < idx->shards.size(); ++g) {
        DeviceGuard dg(idx->shards[g]->device);
        auto F = [](void *p) {
            if (p) (void)hipFree(p);
        };
        if (g < idx->sh_block.size()) F(idx->sh_block[g]);
        if (g < idx->sh_gather.size()) F(idx->sh_gather[g]);
        if (g < idx->sh_q.size()) F(idx->sh_q[g]);
        if (g < idx->sh_scores.size()) F(idx->sh_scores[g]);
        if (g < idx->sh_nf.size()) F(idx->sh_nf[g]);
    }
    idx->sh_block.clear(); idx->sh_gather.clear(); idx->sh_q.clear(); idx->sh_scores.clear(); idx->sh_nf.clear();
    idx->sh_kcap = 0;
}

int free_index(mx_index *idx) {
    {   // nobody may still be inside a call on this handle (a combined search holds idx->mu)
        std::lock_guard<std::mutex> lk(idx->mu);
    }
    if (idx->composite()) {
        idx->pool.reset();  // joins the helper threads
        for (mx_index *sh : idx->shards) {
            DeviceGuard dg(sh->device);
            if (sh->stream) (void)hipStreamSynchronize(sh->stream);
        }
        free_composite_buffers(idx);
        for (void *c : idx->comms)
            if (c && g_rccl.ok) (void)g_rccl.CommDestroy(c);
        for (mx_index *sh : idx->shards) free_index(sh);
        idx->shards.clear();
    }
    DeviceGuard g(idx->device);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    auto F = [](void *p) {
        if (p) (void)hipFree(p);
    };
    F(idx->x); F(idx->scale); F(idx->xh); F(idx->tsc); F(idx->flags); F(idx->xs); F(idx->ss); F(idx->zero_rows); F(idx->wild_list);
    F(idx->amean); F(idx->mean); F(idx->msum);
    Scratch &s = idx->s;
    F(s.qfrag); F(s.qpad); F(s.qnorm2); F(s.theta); F(s.theta_retry); F(s.todo); F(s.dev_flags); F(s.done_ctr);
    if (s.host_flags) (void)hipHostFree(s.host_flags);
    if (s.host_sum) (void)hipHostFree(s.host_sum);
    for (void *hp : {(void *)s.h_q, (void *)s.h_ids, (void *)s.h_scores, (void *)s.h_dists, (void *)s.h_nf})
        if (hp) (void)hipHostFree(hp);
    F(s.qstage); F(s.qscale); F(s.qa); F(s.qb); F(s.qmean); F(s.out_ids); F(s.out_scores); F(s.out_dists); F(s.out_nfound);
    F(s.exact_scratch); F(s.max_err);
    if (idx->ev0) (void)hipEventDestroy(idx->ev0);
    if (idx->ev1) (void)hipEventDestroy(idx->ev1);
    if (idx->ev_wait) (void)hipEventDestroy(idx->ev_wait);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    if (idx->pooled && idx->device >= 0 && idx->device < kMaxDevices) {  // the last index on the device takes the lane-buffer pool with it
        LanePool &lp = g_lanes[idx->device];
        std::lock_guard<std::mutex> lk(lp.mu);
        if (--lp.open_indexes <= 0) {
            lp.open_indexes = 0;
            for (LaneBufs &b : lp.idle) lane_free(b);
            lp.idle.clear();
        }
    }
    delete idx;
    return MX_OK;
}

// One set of lane buffers for the duration of a search_batch call (see LanePool).
struct LaneLease {
    mx_index *idx = nullptr;
    LaneBufs b;
    int take(mx_index *i, int groups) {
        if (i->device < 0 || i->device >= kMaxDevices) return fail(MX_EDEVICE, "device %d out of range", i->device);
        idx = i;
        LanePool &lp = g_lanes[i->device];
        {
            std::lock_guard<std::mutex> lk(lp.mu);
            for (size_t j = 0; j < lp.idle.size(); ++j)
                if (lp.idle[j].nwg == i->nwg && lp.idle[j].groups == groups) {
                    b = lp.idle[j];
                    lp.idle.erase(lp.idle.begin() + (long)j);
                    break;
                }
        }
        if (!b.rec) {
            b.nwg = i->nwg;
            b.groups = groups;
            const size_t lanes = (size_t)i->nwg * kScanThreads * (size_t)groups;  // a 512-query pass numbers 16 waves per workgroup
            hipError_t e = hipMalloc(reinterpret_cast<void **>(&b.rec), lanes * kRecCap * 16 * sizeof(float));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b.tile), lanes * kRecCap * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b.cnt), lanes * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&b.max), lanes * sizeof(float));
            if (e != hipSuccess) {
                lane_free(b);
                idx = nullptr;
                return fail(MX_ENOMEM, "hipMalloc(lane buffers): %s", hipGetErrorString(e));
            }
        }
        Scratch &s = i->s;
        s.lane_rec = b.rec;
        s.lane_tile = b.tile;
        s.lane_cnt = b.cnt;
        s.lane_max = b.max;
        return MX_OK;
    }
    void drop() {  // the caller is host-synchronised with everything that used the buffers
        if (!idx) return;
        Scratch &s = idx->s;
        s.lane_rec = nullptr;
        s.lane_tile = nullptr;
        s.lane_cnt = nullptr;
        s.lane_max = nullptr;
        LanePool &lp = g_lanes[idx->device];
        std::lock_guard<std::mutex> lk(lp.mu);
        if (lp.open_indexes > 0) lp.idle.push_back(b);
        else lane_free(b);
        idx = nullptr;
        b = LaneBufs{};
    }
    ~LaneLease() { drop(); }
};

int ensure_scratch(mx_index *idx) {
    Scratch &s = idx->s;
    if (s.ready) return MX_OK;
    const size_t ds = (size_t)idx->ds;
    MX_HIP(hipMalloc(&s.qfrag, (size_t)kMaxBatch * ds * 2));
    MX_HIP(hipMalloc(&s.qpad, (size_t)kMaxBatch * ds * 4));
    MX_HIP(hipMalloc(&s.qnorm2, kMaxBatch * sizeof(double)));
    MX_HIP(hipMalloc(&s.theta, kMaxBatch * sizeof(float)));
    MX_HIP(hipMalloc(&s.theta_retry, kMaxBatch * sizeof(float)));
    MX_HIP(hipMalloc(&s.todo, kMaxBatch * sizeof(uint32_t)));
    MX_HIP(hipMalloc(&s.dev_flags, kFlagWords * sizeof(uint32_t)));
    MX_HIP(hipMemsetAsync(s.dev_flags, 0, kFlagWords * sizeof(uint32_t), idx->stream));
    s.overflow = s.dev_flags;
    s.cand_cnt = s.dev_flags + kMaxBatch;
    s.e1 = reinterpret_cast<float *>(s.dev_flags + 2 * kMaxBatch);
    s.qflags = s.dev_flags + 3 * kMaxBatch;
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.host_flags), kFlagWords * sizeof(uint32_t), hipHostMallocDefault));
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.host_sum), kSumWords * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    memset(s.host_sum, 0, kSumWords * sizeof(uint32_t));
    MX_HIP(hipMalloc(&s.done_ctr, sizeof(uint32_t)));
    MX_HIP(hipMemsetAsync(s.done_ctr, 0, sizeof(uint32_t), idx->stream));
    MX_HIP(hipMalloc(&s.qstage, (size_t)kMaxBatch * idx->dim * sizeof(float)));
    MX_HIP(hipMalloc(&s.qscale, kMaxBatch * sizeof(float)));
    MX_HIP(hipMalloc(&s.qa, kMaxBatch * sizeof(float)));
    MX_HIP(hipMalloc(&s.qb, kMaxBatch * sizeof(float)));
    MX_HIP(hipMalloc(&s.qmean, kMaxBatch * sizeof(float)));
    MX_HIP(hipMemsetAsync(s.qmean, 0, kMaxBatch * sizeof(float), idx->stream));
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_q), (size_t)kMaxBatch * idx->dim * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_nf), kMaxBatch * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent));
    MX_HIP(hipMalloc(&s.max_err, sizeof(float)));
    MX_HIP(hipMemsetAsync(s.max_err, 0, sizeof(float), idx->stream));
    s.ready = true;
    return MX_OK;
}

int ensure_out(mx_index *idx, int k) {
    Scratch &s = idx->s;
    if (k <= s.kcap) return MX_OK;
    auto F = [](void *p) {
        if (p) (void)hipFree(p);
    };
    F(s.out_ids); F(s.out_scores); F(s.out_dists); F(s.out_nfound);
    s.out_ids = nullptr; s.out_scores = nullptr; s.out_dists = nullptr; s.out_nfound = nullptr;
    for (void *hp : {(void *)s.h_ids, (void *)s.h_scores, (void *)s.h_dists})
        if (hp) (void)hipHostFree(hp);
    s.h_ids = nullptr; s.h_scores = nullptr; s.h_dists = nullptr;
    s.kcap = 0;
    const int kc = std::max(k, 16);
    MX_HIP(hipMalloc(&s.out_ids, (size_t)kMaxBatch * kc * sizeof(uint64_t)));
    MX_HIP(hipMalloc(&s.out_scores, (size_t)kMaxBatch * kc * sizeof(float)));
    MX_HIP(hipMalloc(&s.out_dists, (size_t)kMaxBatch * kc * sizeof(float)));
    MX_HIP(hipMalloc(&s.out_nfound, kMaxBatch * sizeof(int32_t)));
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_ids), (size_t)kMaxBatch * kc * sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent));
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_scores), (size_t)kMaxBatch * kc * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_dists), (size_t)kMaxBatch * kc * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
    s.kcap = kc;
    return MX_OK;
}

// EXACT path scratch for passes of `gcap` queries (4 / 8 / 16 / 32: the kernel's query-per-thread buckets).  The EXACT path
// is the correctness backstop -- one doubly-overflowed query, one k > 256 search -- so it asks for what THIS batch needs
// (4 B per row and query of the pass), not for a full group, and on a full device it settles for smaller passes.
int ensure_exact(mx_index *idx, int k, int *gcap) {
    Scratch &s = idx->s;
    for (int g = *gcap;; g /= 2) {
        const size_t need = exact_group_scratch_bytes(idx->n, k, g);
        if (s.exact_bytes >= need) {
            *gcap = g;
            return MX_OK;
        }
        if (s.exact_scratch) {
            MX_HIP(hipStreamSynchronize(idx->stream));
            (void)hipFree(s.exact_scratch);
        }
        s.exact_scratch = nullptr;
        s.exact_bytes = 0;
        const size_t want = need + need / 4;  // room for appends before the next reallocation
        if (hipMalloc(&s.exact_scratch, want) == hipSuccess) {
            s.exact_bytes = want;
        } else {
            (void)hipGetLastError();
            if (hipMalloc(&s.exact_scratch, need) == hipSuccess) s.exact_bytes = need;
            else (void)hipGetLastError();
        }
        if (s.exact_bytes >= need) {
            *gcap = g;
            return MX_OK;
        }
        if (g <= 4) return fail(MX_ENOMEM, "hipMalloc(EXACT-path scratch, %zu bytes) failed", need);
    }
}

// device buffers being built by ensure_capacity: freed unless handed over
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    void *release() {
        void *r = p;
        p = nullptr;
        return r;
    }
};

int ensure_capacity(mx_index *idx, uint64_t rows) {
    if (rows <= idx->cap) return MX_OK;
    uint64_t want = std::max<uint64_t>(rows, idx->cap + idx->cap / 2);
    want = round_up(std::max<uint64_t>(want, 1024), kTile8Rows);
    if (idx->compressed) {  // the bf16 copy is the corpus: it must grow, there is nothing to fall back to
        DevBuf nh2;
        const size_t hb = (size_t)want * idx->ds * 2;
        MX_HIP(hipMalloc(&nh2.p, hb));
        const size_t used = idx->xh ? (size_t)round_up(idx->n, kTileRows) * idx->ds * 2 : 0;
        if (used) MX_HIP(hipMemcpyAsync(nh2.p, idx->xh, used, hipMemcpyDeviceToDevice, idx->stream));
        MX_HIP(hipMemsetAsync(static_cast<char *>(nh2.p) + used, 0, hb - used, idx->stream));
        MX_HIP(hipStreamSynchronize(idx->stream));
        if (idx->xh) (void)hipFree(idx->xh);
        idx->xh = nh2.release();
        idx->cap = want;
        return MX_OK;
    }
    DevBuf nx, nsc, nh, nts, nam;
    const size_t rowb = (size_t)idx->ds * sizeof(float);
    MX_HIP(hipMalloc(&nx.p, want * rowb));
    MX_HIP(hipMalloc(&nsc.p, want * sizeof(float)));
    float *fx = static_cast<float *>(nx.p), *fsc = static_cast<float *>(nsc.p);
    if (idx->n) {
        MX_HIP(hipMemcpyAsync(fx, idx->x, idx->n * rowb, hipMemcpyDeviceToDevice, idx->stream));
        MX_HIP(hipMemcpyAsync(fsc, idx->scale, idx->n * sizeof(float), hipMemcpyDeviceToDevice, idx->stream));
    }
    MX_HIP(hipMemsetAsync(fx + idx->n * (size_t)idx->ds, 0, (want - idx->n) * rowb, idx->stream));
    MX_HIP(hipMemsetAsync(fsc + idx->n, 0, (want - idx->n) * sizeof(float), idx->stream));
    if (idx->want_filter && idx->kc <= kMaxKC16) {
        // the filter copy is an accelerator, not a requirement: without HBM for it the index
        // keeps working on the f32 scan
        const size_t eb = idx->filter_i8 ? 1 : 2;  // bytes per stored element
        const size_t hb = (size_t)want * idx->ds * eb, tb = (size_t)(want / kTile8Rows) * kTscaleFloats * sizeof(float);
        if (hipMalloc(&nh.p, hb) != hipSuccess || (idx->filter_i8 && hipMalloc(&nts.p, tb) != hipSuccess)) {
            (void)hipGetLastError();
            if (nh.p) (void)hipFree(nh.release());
            nh.p = nullptr;
        } else {
            const uint64_t used_rows = idx->xh ? round_up(idx->n, kTile8Rows) : 0;
            const size_t used = (size_t)used_rows * idx->ds * eb;
            if (used) MX_HIP(hipMemcpyAsync(nh.p, idx->xh, used, hipMemcpyDeviceToDevice, idx->stream));
            MX_HIP(hipMemsetAsync(static_cast<char *>(nh.p) + used, 0, hb - used, idx->stream));
            if (idx->filter_i8) {
                const size_t tused = (size_t)(used_rows / kTile8Rows) * kTscaleFloats * sizeof(float);
                if (tused) MX_HIP(hipMemcpyAsync(nts.p, idx->tsc, tused, hipMemcpyDeviceToDevice, idx->stream));
                MX_HIP(hipMemsetAsync(static_cast<char *>(nts.p) + tused, 0, tb - tused, idx->stream));
                if (idx->centred && idx->xh && idx->amean) {
                    // the a_c array of a CENTRED int8 copy grows with it; without room for it the copy is rewritten plain from the rows
                    if (hipMalloc(&nam.p, want * sizeof(float)) == hipSuccess) {
                        const size_t aused = (size_t)used_rows * sizeof(float);
                        MX_HIP(hipMemcpyAsync(nam.p, idx->amean, std::min(aused, (size_t)idx->cap * sizeof(float)), hipMemcpyDeviceToDevice, idx->stream));
                        if (want * sizeof(float) > aused) MX_HIP(hipMemsetAsync(static_cast<char *>(nam.p) + aused, 0, want * sizeof(float) - aused, idx->stream));
                    } else {
                        (void)hipGetLastError();
                        if (idx->n) {
                            const uint32_t t1c = (uint32_t)((idx->n + kTileRows - 1) / kTileRows);
                            MX_HIP(launch_shadow8(idx->stream, fx, fsc, idx->ds, 0, (uint32_t)round_up(t1c, 2), idx->n, nh.p, static_cast<float *>(nts.p), idx->flags + 2));
                        }
                    }
                }
            } else if (idx->kc <= kMaxKC && hipMalloc(&nam.p, want * sizeof(float)) == hipSuccess) {
                // the a_c array of a (possibly centred) bf16 copy grows with it
                const size_t aused = idx->amean && idx->xh ? (size_t)idx->n * sizeof(float) : 0;
                if (aused) MX_HIP(hipMemcpyAsync(nam.p, idx->amean, aused, hipMemcpyDeviceToDevice, idx->stream));
                MX_HIP(hipMemsetAsync(static_cast<char *>(nam.p) + aused, 0, want * sizeof(float) - aused, idx->stream));
            } else {
                (void)hipGetLastError();
                if (idx->centred && idx->xh && idx->n) {
                    // no room for the a_c array of a CENTRED copy: the fragments just copied hold c/|c| - a_c m and would be
                    // read as c/|c| -- rewrite the copy as a plain one from the rows (rare: 4 bytes per row did not fit)
                    const uint32_t t1c = (uint32_t)((idx->n + kTileRows - 1) / kTileRows);
                    MX_HIP(launch_shadow(idx->stream, fx, fsc, idx->ds, 0, t1c, nh.p, idx->flags + 2));
                }
            }
            if (!idx->xh && idx->n) {  // (re)enabled on a populated index
                const uint32_t t1 = (uint32_t)((idx->n + kTileRows - 1) / kTileRows);
                if (idx->filter_i8)
                    MX_HIP(launch_shadow8(idx->stream, fx, fsc, idx->ds, 0, t1, idx->n, nh.p, static_cast<float *>(nts.p), idx->flags + 2));
                else
                    MX_HIP(launch_shadow(idx->stream, fx, fsc, idx->ds, 0, t1, nh.p, idx->flags + 2));
            }
        }
    }
    MX_HIP(hipStreamSynchronize(idx->stream));
    if (idx->x) (void)hipFree(idx->x);
    if (idx->scale) (void)hipFree(idx->scale);
    // (a centred copy stays centred only if its a_c array came along: a copy built afresh above is a plain one)
    const bool keep_centre = idx->centred && idx->xh && idx->amean && nh.p && nam.p;
    if (idx->xh) (void)hipFree(idx->xh);
    if (idx->tsc) (void)hipFree(idx->tsc);
    if (idx->amean) (void)hipFree(idx->amean);
    idx->x = static_cast<float *>(nx.release());
    idx->scale = static_cast<float *>(nsc.release());
    idx->xh = nh.release();
    idx->tsc = static_cast<float *>(nts.release());
    idx->amean = static_cast<float *>(nam.release());
    idx->centred = keep_centre;
    idx->cap = want;
    return MX_OK;
}

int build_filter_copy(mx_index *idx, bool i8);

// reads the ingest flags back; a batch with non-finite rows is rejected (zero-row list rolled back),
// otherwise the zero-norm rows it brought are committed (list kept ascending for finish_kernel)
int commit_ingest(mx_index *idx, uint32_t (&fl)[5]) {
    MX_HIP(hipMemcpyAsync(fl, idx->flags, sizeof(fl), hipMemcpyDeviceToHost, idx->stream));
    MX_HIP(hipStreamSynchronize(idx->stream));
    if (fl[0] != 0) {
        const uint32_t keep[2] = {(uint32_t)idx->n_zero, (uint32_t)idx->n_wild};
        MX_HIP(hipMemcpyAsync(idx->flags + 3, keep, sizeof(keep), hipMemcpyHostToDevice, idx->stream));
        MX_HIP(hipStreamSynchronize(idx->stream));
        return fail(MX_EINVAL, "%u row(s) contain non-finite values; nothing inserted", fl[0]);
    }
    if (fl[4] != idx->n_wild) {  // new listed rows arrive in atomic order: keep the list ascending
        const size_t have = std::min<size_t>(fl[4], kWildCap);
        if (have > std::min<uint64_t>(idx->n_wild, kWildCap)) {
            std::vector<uint32_t> z(have);
            MX_HIP(hipMemcpy(z.data(), idx->wild_list, have * sizeof(uint32_t), hipMemcpyDeviceToHost));
            std::sort(z.begin(), z.end());
            MX_HIP(hipMemcpy(idx->wild_list, z.data(), have * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        idx->n_wild = fl[4];
    }
    if (fl[3] != idx->n_zero) {
        const size_t have = std::min<size_t>(fl[3], kZeroCap);
        if (have > std::min<uint64_t>(idx->n_zero, kZeroCap)) {  // new entries arrive in atomic order: sort
            std::vector<uint32_t> z(have);
            MX_HIP(hipMemcpy(z.data(), idx->zero_rows, have * sizeof(uint32_t), hipMemcpyDeviceToHost));
            std::sort(z.begin(), z.end());
            MX_HIP(hipMemcpy(idx->zero_rows, z.data(), have * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        idx->n_zero = fl[3];
    }
    return MX_OK;
}

// rows already on the device ([n, dim]); appends and validates
int add_device_locked(mx_index *idx, const float *d_rows, uint64_t n, uint64_t *first_id) {
    if (n == 0) {
        if (first_id) *first_id = idx->idmap.id_of((uint32_t)idx->n);
        return MX_OK;
    }
    if (idx->n + n > 0xfffffff0ull) return fail(MX_EINSERT, "index shard limited to 2^32 rows");
    int rc = ensure_capacity(idx, idx->n + n);
    if (rc != MX_OK) return rc;
    MX_HIP(hipMemsetAsync(idx->flags, 0, 2 * sizeof(uint32_t), idx->stream));
    if (idx->compressed) {
        // rows -> f32 staging window (validated, padded, 1/|c|) -> bf16 fragments.  The window is aligned
        // to the tile of the first new row; rows of that tile that are already stored are not touched.
        constexpr uint64_t kWin = 65536;
        if (!idx->xs) {
            MX_HIP(hipMalloc(reinterpret_cast<void **>(&idx->xs), (size_t)(kWin + kTileRows) * idx->ds * sizeof(float)));
            MX_HIP(hipMalloc(reinterpret_cast<void **>(&idx->ss), (size_t)(kWin + kTileRows) * sizeof(float)));
            idx->xs_rows = kWin + kTileRows;
        }
        uint32_t fl[5] = {0, 0, 0, 0, 0};
        for (uint64_t done = 0; done < n; done += kWin) {
            const uint64_t m = std::min(kWin, n - done), g0 = idx->n + done;     // global rows [g0, g0 + m)
            const uint64_t t0 = g0 / kTileRows, off = g0 % kTileRows;
            MX_HIP(launch_ingest(idx->stream, d_rows + (size_t)done * idx->dim, m, idx->dim, idx->xs, idx->ss, off, idx->ds, idx->flags,
                                 idx->raw_ingest ? 1 : 0, idx->zero_rows, idx->wild_list, g0));
            MX_HIP(launch_shadow(idx->stream, idx->xs, idx->ss, idx->ds, (uint32_t)t0, (uint32_t)((g0 + m + kTileRows - 1) / kTileRows),
                                 idx->xh, idx->flags + 2, (uint32_t)t0, g0, g0 + m));
        }
        int rc2 = commit_ingest(idx, fl);
        if (rc2 != MX_OK) return rc2;  // rows past idx->n are never read by a search; the next append overwrites them
        idx->wild_rows += fl[1];
        if (first_id) *first_id = idx->idmap.id_of((uint32_t)idx->n);
        idx->n += n;
        return MX_OK;
    }
    MX_HIP(launch_ingest(idx->stream, d_rows, n, idx->dim, idx->x, idx->scale, idx->n, idx->ds, idx->flags, 2, idx->zero_rows, idx->wild_list, idx->n));
    // tiles touched by this append (the first one may already be partly filled; an 8-bit half tile is requantised
    // as a whole: its step depends on all of its rows)
    auto refilter = [&](uint64_t row_lo, uint64_t row_hi) -> hipError_t {
        const uint32_t h0 = (uint32_t)(row_lo / kTileRows), h1 = (uint32_t)((row_hi + kTileRows - 1) / kTileRows);
        const bool ctr = idx->centred && idx->amean && idx->mean;
        if (idx->filter_i8)  // both halves of the last 64-row scan tile: the one past row_hi becomes zeros with step 0
            return launch_shadow8(idx->stream, idx->x, idx->scale, idx->ds, h0, (uint32_t)round_up(std::max(h1, h0 + 1), 2), row_hi, idx->xh, idx->tsc, idx->flags + 2,
                                  ctr ? idx->mean : nullptr, ctr ? idx->amean : nullptr, ctr ? idx->flags + 5 : nullptr);
        return launch_shadow(idx->stream, idx->x, idx->scale, idx->ds, h0, std::max(h1, h0 + 1), idx->xh, idx->flags + 2, 0, 0, ~0ull,
                             ctr ? idx->mean : nullptr, ctr ? idx->amean : nullptr);
    };
    if (idx->xh) MX_HIP(refilter(idx->n, idx->n + n));
    uint32_t fl[5] = {0, 0, 0, 0, 0};
    rc = commit_ingest(idx, fl);
    if (rc != MX_OK) {
        // rows past idx->n are never read, but the filter copy's tile of row n may now hold garbage: rebuild it
        if (idx->xh) {
            const std::string keep = last_error_slot();
            MX_HIP(refilter(idx->n, idx->n));
            last_error_slot() = keep;
        }
        return rc;
    }
    idx->wild_rows += fl[1];
    if (first_id) *first_id = idx->idmap.id_of((uint32_t)idx->n);  // local.rs:63: next_id = len + 1
    idx->n += n;
    // A demotion is not for life: it was decided on the rows and the queries of its day.  Once the collection has doubled
    // since, the automatic choice gets its int8 copy back (one pass over the rows, as when it was first built) and the
    // demotion rule judges it afresh; mx_index_set_filter_copy(idx, 1) does the same at once.
    if (idx->filter_auto && idx->xh && !idx->filter_i8 && idx->demoted_at_rows && idx->n >= 2 * idx->demoted_at_rows &&
        idx->ds <= kAutoI8MaxDim) {
        const std::string keep = last_error_slot();
        if (build_filter_copy(idx, true) == MX_OK) {
            idx->demoted_at_rows = 0;
            idx->stats.filter_promotions += 1;
        } else {
            idx->demoted_at_rows = idx->n;  // no room now: ask again after the next doubling
            last_error_slot() = keep;
        }
    }
    // ... and a bf16 copy that was built plain (fewer than 256 rows at the time, or rows that did not sit in a cone) is rebuilt once the
    // collection has doubled: should the rows sit in a cone by now, it comes back centred (section 3.2c) -- one pass over the rows per
    // doubling.  (An automatic copy gets there through the promotion above; this is for a pinned bf16 copy.)
    // (plain_bf16_rows == 0: the copy grew with the index from empty and was never built in one piece: first look at 256 rows)
    if (idx->xh && !idx->filter_i8 && !idx->compressed && !idx->centred && idx->kc <= kMaxKC && idx->n >= 256 &&
        idx->n >= 2 * std::max<uint64_t>(idx->plain_bf16_rows, 128) && !(idx->filter_auto && idx->demoted_at_rows)) {
        const std::string keep = last_error_slot();
        if (build_filter_copy(idx, false) != MX_OK) {
            idx->plain_bf16_rows = idx->n;  // no room now: ask again after the next doubling
            last_error_slot() = keep;
        }
    }
    return MX_OK;
}

// how many tiles the sample launch visits: enough that the k-th largest of 2*nwg lane maxima is a
// useful threshold (expected survivors of the collect launch ~ k * N / sample, times the margin's
// share) and small enough to stay a few percent of the pass
uint32_t sample_stride(uint64_t full_tiles, int nwg, int k, bool filt8, int ds, bool centred8 = false) {
    // 1/64 of the tiles for k <= 10.  Measured at 10M x 384 (B = 256): the sample launch costs 65 / 38 / 24 /
    // 17 us at 1/32, 1/64, 1/128, 1/256; a weaker threshold means more records for finish_kernel to sift
    // (62 / 63 / 72 us, and at 1/256 lanes start to overflow their 32 records), while the collect launch
    // hardly notices since appends became whole-record stores (1.76 / 1.77 / 1.79 ms).  1/64 is the
    // fastest end to end.  The int8 copy's certificate is four to five times wider, so a weak threshold costs it far
    // more rows: 1/16 of the tiles up to 512 dims, 1/8 at 768, 1/4 at 1024 (a smaller sample makes lanes overflow
    // their 64 records and the retry pass costs a whole scan: scripts/r3_i8_sample.sh, r3_dims.sh).
    // (A tuning constant: results do not depend on it.)
    // (MEMEX_HIP_DEBUG=sample_div=N tries others: a tuning knob, results do not depend on it.)
    // (a centred int8 copy, whose certificate is four to five times tighter again, gains nothing from a smaller sample: 1/32 against
    // 1/16 on the enc_like leg 169.3k against 169.0k queries/s; at 1/64 lanes overflow and the copy is demoted: gpurun_out/r6m_*)
    (void)centred8;
    double div = !filt8 ? 64.0 : ds <= 512 ? 16.0 : ds <= 768 ? 8.0 : 4.0;
    if (const int dv = debug_flag("sample_div", 0); dv >= 2 && dv <= 4096) div = (double)dv;
    // larger k: the sample grows like k / 640 for every copy (int8 at k = 30 / 100: 1.33 / 1.56 ms per step with
    // 1/16 / 0.16 of the tiles against 1.93 / 1.77 with three and ten times the k = 10 sample)
    const double f = std::min(0.5, std::max(1.0 / div, (double)k / 640.0));
    const uint64_t target = std::max<uint64_t>((uint64_t)nwg, (uint64_t)((double)full_tiles * f));
    return (uint32_t)std::max<uint64_t>(1, full_tiles / std::max<uint64_t>(target, 1));
}

constexpr size_t kExactKeepBytes = 256ull << 20;  // EXACT-path scratch kept between batches up to this size

int run_exact(mx_index *idx, const std::vector<int> &qs, int k, uint64_t *d_ids, float *d_scores, float *d_dists,
              int32_t *d_nfound) {
    if (qs.empty()) return MX_OK;
    int gcap = qs.size() <= 4 ? 4 : qs.size() <= 8 ? 8 : qs.size() <= 16 ? 16 : kExactGroup;
    int rc = ensure_exact(idx, k, &gcap);
    if (rc != MX_OK) return rc;
    Scratch &s = idx->s;
    // groups of up to gcap (<= 32) queries share one pass over the rows (launch_exact_group); the groups of a batch reuse the
    // scratch in stream order
    for (size_t g0 = 0; g0 < qs.size(); g0 += (size_t)gcap) {
        ExactGroup grp{};
        grp.n = (int)std::min<size_t>((size_t)gcap, qs.size() - g0);
        for (int j = 0; j < grp.n; ++j) grp.q[j] = qs[g0 + j];
        MX_HIP(launch_exact_group(idx->stream, k, idx->ds, idx->compressed ? nullptr : idx->x, idx->xh, idx->n, idx->idmap, s.qpad,
                                  s.qnorm2, grp, s.exact_scratch, d_ids, d_scores, d_dists, d_nfound));
    }
    return MX_OK;
}

// builds the filter copy of kind (i8 ? int8 : bf16) from the f32 rows, complete before it replaces whatever copy is resident
int build_filter_copy(mx_index *idx, bool i8) {
    DevBuf nh, nts, nec, nam;
    const size_t hb = (size_t)idx->cap * idx->ds * (i8 ? 1 : 2), tb = (size_t)(idx->cap / kTile8Rows) * kTscaleFloats * sizeof(float);
    hipError_t e = hipMalloc(&nh.p, hb);
    if (e == hipSuccess && i8) e = hipMalloc(&nts.p, tb);
    if (e == hipSuccess) e = hipMalloc(&nec.p, 2 * sizeof(uint32_t));  // [0] largest residual, [1] longest centred row (int8)
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(MX_ENOMEM, "hipMalloc(filter copy, %zu bytes): %s", hb, hipGetErrorString(e));
    }
    uint32_t *ec = static_cast<uint32_t *>(nec.p);
    MX_HIP(hipMemsetAsync(nh.p, 0, hb, idx->stream));
    if (i8) MX_HIP(hipMemsetAsync(nts.p, 0, tb, idx->stream));
    MX_HIP(hipMemsetAsync(ec, 0, 2 * sizeof(uint32_t), idx->stream));
    const uint32_t t1 = (uint32_t)((idx->n + kTileRows - 1) / kTileRows);
    // A copy rebuilt from a populated index is CENTRED on the rows' mean direction (launch_shadow / launch_shadow8): a corpus
    // that a plain int8 certificate could not resolve is a dense one, and embedding corpora are dense because they sit in a
    // cone -- what is left of a row after its component along the cone's axis is removed is several times shorter, and so is
    // the rounding (bf16) or quantisation (int8: round 6) error the scan's certificate has to cover.
    const bool centre_ok = true;
    bool centre = false;
    if (idx->kc <= kMaxKC && !idx->compressed && hipMalloc(&nam.p, (size_t)idx->cap * sizeof(float)) == hipSuccess) {
        MX_HIP(hipMemsetAsync(nam.p, 0, (size_t)idx->cap * sizeof(float), idx->stream));
        if (centre_ok && idx->n >= 256) {
            if (!idx->mean) MX_HIP(hipMalloc(reinterpret_cast<void **>(&idx->mean), (size_t)idx->ds * sizeof(float)));
            if (!idx->msum) MX_HIP(hipMalloc(reinterpret_cast<void **>(&idx->msum), ((size_t)idx->ds + 1) * sizeof(float)));
            MX_HIP(launch_mean_dir(idx->stream, idx->x, idx->scale, idx->n, idx->ds, idx->msum, idx->mean));
            // worth it only for a corpus that does sit in a cone: |mean of the unit rows| = the typical a_c; below 0.3 the
            // centred rows are < 5 % shorter and the scan's extra epilogue work buys nothing
            float msq = 0.f;
            MX_HIP(hipMemcpyAsync(&msq, idx->msum + idx->ds, sizeof(float), hipMemcpyDeviceToHost, idx->stream));
            MX_HIP(hipStreamSynchronize(idx->stream));
            centre = std::isfinite(msq) && msq / (float)idx->n >= 0.3f;
        }
    } else {
        (void)hipGetLastError();
    }
    if (i8 && !centre && nam.p) (void)hipFree(nam.release());  // (a plain int8 copy has no use for the a_c array)
    if (i8)
        MX_HIP(launch_shadow8(idx->stream, idx->x, idx->scale, idx->ds, 0, (uint32_t)round_up(t1, 2), idx->n, nh.p, static_cast<float *>(nts.p), ec,
                              centre ? idx->mean : nullptr, centre ? static_cast<float *>(nam.p) : nullptr, centre ? ec + 1 : nullptr));
    else
        MX_HIP(launch_shadow(idx->stream, idx->x, idx->scale, idx->ds, 0, t1, nh.p, ec, 0, 0, ~0ull, centre ? idx->mean : nullptr,
                             centre ? static_cast<float *>(nam.p) : nullptr));
    MX_HIP(hipMemcpyAsync(idx->flags + 2, ec, sizeof(uint32_t), hipMemcpyDeviceToDevice, idx->stream));
    MX_HIP(hipMemcpyAsync(idx->flags + 5, ec + 1, sizeof(uint32_t), hipMemcpyDeviceToDevice, idx->stream));
    MX_HIP(hipStreamSynchronize(idx->stream));
    if (idx->xh) (void)hipFree(idx->xh);
    if (idx->tsc) (void)hipFree(idx->tsc);
    if (idx->amean) (void)hipFree(idx->amean);
    idx->xh = nh.release();
    idx->tsc = static_cast<float *>(nts.release());
    idx->amean = static_cast<float *>(nam.release());
    idx->centred = centre;
    idx->filter_i8 = i8;
    idx->plain_bf16_rows = (!i8 && !centre) ? std::max<uint64_t>(idx->n, 1) : 0;
    idx->i8_batches = idx->i8_retry_batches = 0;
    for (double &w : idx->wait_ema_us) w = 0.0;
    return MX_OK;
}

// int8 filter copy -> bf16 filter copy, rebuilt from the f32 rows (an automatic choice that did not suit the data).
// The bf16 copy is complete (and its residual word known) before the int8 copy is released: any failure leaves the
// index exactly as it was.  MX_ENOMEM: no room for the wider copy, the caller stays on int8.
int demote_filter(mx_index *idx) {
    const std::string keep = last_error_slot();
    const int rc = build_filter_copy(idx, false);
    if (rc == MX_ENOMEM) last_error_slot() = keep;
    if (rc == MX_OK) idx->demoted_at_rows = std::max<uint64_t>(idx->n, 1);
    return rc;
}

// one batch (B <= 256) with queries and outputs on the device
int search_batch(mx_index *idx, const float *d_q, int B, int k, uint64_t *d_ids, float *d_scores, float *d_dists,
                 int32_t *d_nfound) {
    int rc = ensure_scratch(idx);
    if (rc != MX_OK) return rc;
    Scratch &s = idx->s;
    hipStream_t st = idx->stream;
    const bool trivial = idx->n == 0 || k == 0;
    if (idx->compressed && idx->kc > kMaxKC16 && !trivial)
        return fail(MX_EUNSUPPORTED, "a compressed corpus supports dim <= %d", kMaxKC16 * kChunkFloats);
    // wide rows (768 < dim_pad <= 1536) have their own scan kernel over the filter copy: 128 queries per pass
    const bool filt8 = idx->xh != nullptr && idx->filter_i8 && !idx->compressed;  // 8-bit copy: 256 queries per pass at every width
    const bool wide = idx->kc > kMaxKC && !filt8;
    const bool fast = !trivial && idx->mode == MX_SEARCH_AUTO && (idx->kc > kMaxKC ? idx->xh != nullptr && idx->kc <= kMaxKC16 : true) &&
                      (!idx->compressed || idx->wild_rows == 0) && k <= 256 && idx->n_zero <= (uint64_t)kZeroCap &&
                      idx->n_wild <= (uint64_t)kWildCap;
    // a batch of more than 256 queries is one pass of the int8 scan with two query groups per wave (up to 512 dims),
    // otherwise two passes
    // (a centred int8 copy runs passes of 256: the two-group variant has no register left for the per-row epilogue)
    const bool centred8 = idx->centred && filt8 && idx->amean && idx->mean && idx->kc <= kMaxKC;
    const bool x2 = fast && filt8 && !centred8 && idx->kc <= kMaxKC8x2 && B > kPassBatch;
    if (B > kPassBatch && !x2 && !(fast && wide)) {
        rc = search_batch(idx, d_q, kPassBatch, k, d_ids, d_scores, d_dists, d_nfound);
        if (rc != MX_OK) return rc;
        const size_t o = (size_t)kPassBatch * k;
        return search_batch(idx, d_q + (size_t)kPassBatch * idx->dim, B - kPassBatch, k, d_ids + o, d_scores + o,
                            d_dists ? d_dists + o : nullptr, d_nfound + kPassBatch);
    }
    if (fast && wide && B > kWideBatch) {
        rc = search_batch(idx, d_q, kWideBatch, k, d_ids, d_scores, d_dists, d_nfound);
        if (rc != MX_OK) return rc;
        const size_t o = (size_t)kWideBatch * k;
        return search_batch(idx, d_q + (size_t)kWideBatch * idx->dim, B - kWideBatch, k, d_ids + o, d_scores + o,
                            d_dists ? d_dists + o : nullptr, d_nfound + kWideBatch);
    }
    // the bf16 copy of this index is centred on its rows' mean direction (build_filter_copy): queries are split the same way
    const bool centred = centred8 || (idx->centred && idx->xh && !filt8 && !wide && !idx->compressed && idx->amean && idx->mean && idx->kc <= kMaxKC);
    LaneLease lease;  // every return below is host-synchronised with the kernels that used the lane buffers
    if ((rc = lease.take(idx, x2 ? 2 : 1)) != MX_OK) return rc;
    MX_HIP(launch_prep_queries(st, d_q, B, idx->dim, idx->ds, s.qfrag, s.qpad, s.qnorm2, s.theta, s.e1,
                               idx->xh ? idx->flags + 2 : nullptr, s.overflow, s.qflags, s.qa, s.qb, filt8, s.qscale,
                               centred ? idx->mean : nullptr, s.qmean, centred8 ? idx->flags + 5 : nullptr));
    const uint32_t *h_ovf = s.host_flags, *h_qfl = s.host_flags + 3 * kMaxBatch;
    auto any_bad_query = [&] {
        uint32_t bad = 0;
        for (int b = 0; b < B; ++b) bad |= h_qfl[b];
        return bad != 0;
    };
    bool timed = false;

    FinishParams fp;
    fp.k = k;
    fp.ds = idx->ds;
    fp.nwg = idx->nwg;
    fp.x = idx->compressed ? nullptr : idx->x;
    fp.xh = idx->xh;
    fp.scale = idx->scale;
    fp.n_rows = trivial ? 0 : idx->n;
    fp.idmap = idx->idmap;
    fp.qpad = s.qpad;
    fp.qnorm2 = s.qnorm2;
    fp.e1 = s.e1;
    fp.qa = s.qa;
    fp.qb = s.qb;
    fp.terr = filt8 ? idx->tsc : nullptr;
    fp.e2 = (float)(idx->ds + 8) * 5.9604645e-8f + 1e-6f;  // f32 fma dot of <= ds terms of unit vectors, any order
    fp.lane_rec = s.lane_rec;
    fp.lane_tile = s.lane_tile;
    fp.lane_cnt = s.lane_cnt;
    fp.theta = s.theta;
    fp.zero_rows = idx->zero_rows;
    fp.n_zero = (uint32_t)std::min<uint64_t>(idx->n_zero, kZeroCap);
    fp.wild_rows = idx->wild_list;
    fp.n_wild = (uint32_t)std::min<uint64_t>(idx->n_wild, kWildCap);
    fp.overflow = s.overflow;
    fp.todo = nullptr;
    fp.theta_retry = s.theta_retry;
    fp.cand_cnt = s.cand_cnt;
    fp.ids = d_ids;
    fp.scores = d_scores;
    fp.dists = d_dists;
    fp.n_found = d_nfound;
    fp.max_err = idx->profiling ? s.max_err : nullptr;
    fp.done_ctr = s.done_ctr;
    fp.dev_flags = s.dev_flags;
    fp.host_flags = s.host_sum;
    fp.n_queries = B;
    fp.host_out = s.out_on_host ? 1 : 0;
    // finish + completion: the kernel's last workgroup writes the batch summary into pinned host memory and
    // stores the launch's sequence number behind it; the host spins on that word (no D2H copy command and no
    // memset between batches: the host gap between two batches drops from 45 to 22 us, the kernel grows by
    // 8: scripts/r2_step_gaps.sh; waiting in hipStreamSynchronize is as fast, see MEMEX_HIP_SPIN below).  A kernel that never signals (fault) is
    // caught by the synchronize after the spin budget.  -> the summary's overflow code
    auto finish_and_wait = [&]() -> int {
        fp.seq = ++s.flag_seq;
        MX_HIP(launch_finish(st, B, fp));
        // Default: a SLEEPING wait -- a server thread must not burn a core for the ~1 ms of a batch.  The HIP runtime's
        // own waits do not help: hipStreamSynchronize and hipEventSynchronize (also on a hipEventBlockingSync event)
        // poll with the default scheduling policy -- 100 % of a core in all three forms (scripts/gpu_wait_modes.py) --
        // and hipSetDeviceFlags(BlockingSync) is not this library's to set in a host process.  So: sleep for most of
        // what the last batches took (clock_nanosleep), then poll the completion word; the estimate follows the
        // workload (an EMA of the wait just observed).  MEMEX_HIP_SPIN=1 (bench.py sets it) polls from the start.
        static const bool no_spin = [] {
            const char *sp = getenv("MEMEX_HIP_SPIN");
            return !(sp && sp[0] == '1');
        }();
        const auto t0 = std::chrono::steady_clock::now();
        double &ema = idx->wait_ema_us[B <= 32 ? 0 : B <= 128 ? 1 : B <= 256 ? 2 : 3];
        bool overslept = false;  // the batch was already complete when the nap ended: the estimate is too long
        if (no_spin && ema > 150.0) {
            const double nap_us = 0.8 * ema - 60.0;  // timer slack and wake-up latency stay inside the estimate
            if (nap_us > 50.0) {
                struct timespec ts;
                ts.tv_sec = (time_t)(nap_us / 1e6);
                ts.tv_nsec = (long)((nap_us - (double)ts.tv_sec * 1e6) * 1e3);
                (void)clock_nanosleep(CLOCK_MONOTONIC, 0, &ts, nullptr);
                overslept = __atomic_load_n(&s.host_sum[4], __ATOMIC_ACQUIRE) == fp.seq;
            }
        }
        bool done = false;
        for (unsigned spins = 1;; ++spins) {
            if (__atomic_load_n(&s.host_sum[4], __ATOMIC_ACQUIRE) == fp.seq) {
                done = true;
                break;
            }
            if ((spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
            __builtin_ia32_pause();
        }
        if (done) {
            // what was observed includes the nap: when the batch had finished before the nap did (batches got faster: a
            // cleared or smaller index, another k) only an upper bound is known, so the estimate is halved instead of
            // creeping down 5 % per batch
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            ema = overslept ? 0.5 * ema : ema > 0.0 ? 0.75 * ema + 0.25 * us : us;
            return MX_OK;
        }
        MX_HIP(hipStreamSynchronize(st));  // a kernel that never signals (fault): the synchronize reports it
        if (__atomic_load_n(&s.host_sum[4], __ATOMIC_ACQUIRE) == fp.seq) return MX_OK;
        return fail(MX_EDEVICE, "finish_kernel did not signal completion");
    };
    auto fetch_flags = [&]() -> int {  // the per-query words (overflowed batches and the EXACT path only)
        MX_HIP(hipMemcpyAsync(s.host_flags, s.dev_flags, kFlagWords * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        MX_HIP(hipStreamSynchronize(st));
        return MX_OK;
    };

    std::vector<int> exact;
    if (trivial) {
        fp.seq = ++s.flag_seq;
        MX_HIP(launch_finish(st, B, fp));  // n_found = 0, empty slots
    } else if (fast) {
        const uint64_t trows = filt8 ? kTile8Rows : kTileRows;  // rows per scan tile
        const uint64_t tiles = (idx->n + trows - 1) / trows, full = idx->n / trows;
        ScanParams p;
        p.x = idx->x;
        p.xh = idx->xh;
        p.scale = idx->scale;
        p.qfrag = s.qfrag;
        p.theta = s.theta;
        p.n_rows = idx->n;
        p.ds = (uint32_t)idx->ds;
        p.wave_mask = (1u << ((B + 31) / 32)) - 1u;  // waves whose 32 columns are all padding skip their MFMAs
        p.lane_rec = s.lane_rec;
        p.lane_tile = s.lane_tile;
        p.lane_cnt = s.lane_cnt;
        p.lane_max = s.lane_max;
        p.overflow = s.overflow;
        p.tscale = idx->tsc;
        p.qscale = s.qscale;
        p.qa = s.qa;
        p.qb = s.qb;
        p.amean = centred ? idx->amean : nullptr;
        p.qmean = s.qmean;
        auto scan = [&](bool collect) {
            if (filt8) return launch_scan8(st, idx->kc, collect, idx->nwg, p, x2);
            if (wide) return launch_scan16w(st, idx->kc, collect, idx->nwg, p);
            return idx->xh ? launch_scan16(st, idx->kc, collect, idx->nwg, p) : launch_scan(st, idx->kc, collect, idx->nwg, p);
        };
        auto collect = [&](bool first) -> int {
            p.tile_begin = 0;
            p.tile_end = (uint32_t)tiles;
            p.tile_stride = 1;
            // the first collect launch of a batch is the one the roofline is quoted on
            if (first && idx->profiling) MX_HIP(hipEventRecord(idx->ev0, st));
            MX_HIP(scan(true));
            if (first && idx->profiling) {
                MX_HIP(hipEventRecord(idx->ev1, st));
                timed = true;
            }
            if (first) {
                idx->stats.scan_launches += 1;
                idx->stats.scan_bytes += tiles * trows * idx->ds * (filt8 ? 1ull : idx->xh ? 2ull : 4ull);
            }
            return MX_OK;
        };
        // a lane holds 16 scores per tile of its workgroup: up to 2 tiles per workgroup everything fits
        // in the lane buffers and no threshold is needed
        if (tiles > 2ull * idx->nwg) {
            p.tile_begin = 0;
            p.tile_end = (uint32_t)full;
            p.tile_stride = sample_stride(full, idx->nwg, k, filt8, idx->ds, centred8);
            MX_HIP(scan(false));
            MX_HIP(launch_theta(st, B, k, idx->nwg, s.lane_max, s.qa, !filt8, s.theta));
        }
        if ((rc = collect(true)) != MX_OK) return rc;
        if ((rc = finish_and_wait()) != MX_OK) return rc;
        if (timed) {
            float ms = 0.f;
            MX_HIP(hipEventElapsedTime(&ms, idx->ev0, idx->ev1));
            idx->stats.scan_ms += ms;
            timed = false;
        }
        if (s.host_sum[2]) return fail(MX_EINVAL, "a query contains non-finite values");
        idx->stats.candidates += s.host_sum[1];
        if (s.host_sum[0]) {
            if ((rc = fetch_flags()) != MX_OK) return rc;
            int retry = 0;
            for (int b = 0; b < B; ++b) {
                if (h_ovf[b] == 1) ++retry;
                else if (h_ovf[b] >= 2) exact.push_back(b);
            }
            if (filt8) idx->i8_retry_batches += 1;
            // ... or the retry pass (a second whole scan) has become the rule: more than a quarter of the batches
            const bool habitual = filt8 && idx->i8_batches >= 8 && idx->i8_retry_batches * 4 > idx->i8_batches;
            if (filt8 && idx->filter_auto && ((size_t)(retry + (int)exact.size()) * 16 > (size_t)std::max(B, 16) || habitual)) {
                // More than 1/16 of the batch (and more than one query) did not fit the int8 pass: this corpus is too dense for the int8
                // certificate (neighbourhoods narrower than ~0.05 in cosine).  Rebuild the copy as bf16 (one pass
                // over the f32 rows) and answer the batch on it; the index stays on bf16.
                // Round 6: first the same copy CENTRED on the rows' mean direction (section 3.2c carried over to int8: what an encoder
                // produces sits in a cone, and the quantisation error of the short residual vectors is several times smaller); only
                // a corpus without a cone, or one that overflows the centred copy as well, goes to bf16.
                if (!idx->centred && idx->kc <= kMaxKC && idx->n >= 256) {
                    const std::string keep = last_error_slot();
                    if (build_filter_copy(idx, true) == MX_OK && idx->centred) {
                        lease.drop();
                        return search_batch(idx, d_q, B, k, d_ids, d_scores, d_dists, d_nfound);
                    }
                    last_error_slot() = keep;
                }
                const int drc = demote_filter(idx);
                if (drc == MX_OK) {
                    idx->stats.filter_demotions += 1;
                    lease.drop();
                    return search_batch(idx, d_q, B, k, d_ids, d_scores, d_dists, d_nfound);
                }
                if (drc != MX_ENOMEM) return drc;  // (no room for the wider copy: the batch finishes on int8 below)
            }
            if (retry) {
                // ONE more pass for all overflowed queries of the batch, with the threshold finish derived
                // from what they did collect (everyone else is parked at theta = +inf)
                idx->stats.retry_queries += (uint64_t)retry;
                p.wave_mask = 0;  // only the waves that hold a rescanned query multiply: the pass runs at the stream's rate
                for (int b = 0; b < B; ++b)
                    if (h_ovf[b] == 1) p.wave_mask |= 1u << (b >> 5);
                MX_HIP(launch_retry_setup(st, s.theta, s.theta_retry, s.overflow, s.todo));
                if ((rc = collect(false)) != MX_OK) return rc;
                fp.todo = s.todo;
                if ((rc = finish_and_wait()) != MX_OK) return rc;
                exact.clear();
                if (s.host_sum[0]) {
                    if ((rc = fetch_flags()) != MX_OK) return rc;
                    for (int b = 0; b < B; ++b)
                        if (h_ovf[b] != 0) exact.push_back(b);
                }
            }
        }
        idx->stats.fallback_queries += exact.size();
    } else {
        if ((rc = fetch_flags()) != MX_OK) return rc;
        if (any_bad_query()) return fail(MX_EINVAL, "a query contains non-finite values");
        for (int b = 0; b < B; ++b) exact.push_back(b);
    }
    if (!exact.empty() || trivial || !fast) {
        rc = run_exact(idx, exact, k, d_ids, d_scores, d_dists, d_nfound);
        if (rc != MX_OK) return rc;
        MX_HIP(hipStreamSynchronize(st));
        if (idx->mode == MX_SEARCH_AUTO && idx->s.exact_bytes > kExactKeepBytes) {  // a large scratch does not stay behind a fallback batch
            (void)hipFree(idx->s.exact_scratch);
            idx->s.exact_scratch = nullptr;
            idx->s.exact_bytes = 0;
        }
    }
    // (fast path with nothing left to do: finish_kernel's completion word was stored after every result of
    // the batch had been fenced to device scope -- the results are in HBM, nothing else is queued)
    idx->stats.searches += 1;
    idx->stats.queries += (uint64_t)B;
    if (filt8 && fast) idx->i8_batches += 1;
    return MX_OK;
}

// ---------------------------------------------------------------------------------------------
// composite: rows dealt block-cyclically to shard indexes (one per device), local top-k per shard,
// exchange of the packed [ids | dists] blocks (RCCL all-gather over xGMI, or peer copies), merge
// ---------------------------------------------------------------------------------------------
void sync_shard_streams(mx_index *idx) {
    for (mx_index *sh : idx->shards) {
        DeviceGuard dg(sh->device);
        if (sh->stream) (void)hipStreamSynchronize(sh->stream);
    }
}

// the same with a deadline (after a failed collective part of it may sit on some shard's stream and never complete):
// polls hipStreamQuery; false = a stream did not drain in time
bool sync_shard_streams_within(mx_index *idx, double seconds) {
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
    for (mx_index *sh : idx->shards) {
        DeviceGuard dg(sh->device);
        if (!sh->stream) continue;
        for (;;) {
            const hipError_t e = hipStreamQuery(sh->stream);
            if (e == hipSuccess) break;
            if (e != hipErrorNotReady) {
                (void)hipGetLastError();
                return false;
            }
            if (std::chrono::steady_clock::now() > t_end) return false;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    return true;
}

int ensure_composite_buffers(mx_index *idx, int k) {
    if (k <= idx->sh_kcap) return MX_OK;
    sync_shard_streams(idx);  // the last batch's all-gather may still be running on shards >= 1
    free_composite_buffers(idx);
    const int kc = std::max(k, 16);
    const size_t G = idx->shards.size();
    const size_t blk = (size_t)kMaxBatch * kc * 12;
    idx->sh_block.assign(G, nullptr); idx->sh_gather.assign(G, nullptr); idx->sh_q.assign(G, nullptr);
    idx->sh_scores.assign(G, nullptr); idx->sh_nf.assign(G, nullptr);
    for (size_t g = 0; g < G; ++g) {
        DeviceGuard dg(idx->shards[g]->device);
        MX_HIP(hipMalloc(&idx->sh_block[g], blk));
        if (g == 0 || idx->use_rccl) MX_HIP(hipMalloc(&idx->sh_gather[g], blk * G));
        MX_HIP(hipMalloc(reinterpret_cast<void **>(&idx->sh_q[g]), (size_t)kMaxBatch * idx->dim * sizeof(float)));
        MX_HIP(hipMalloc(reinterpret_cast<void **>(&idx->sh_scores[g]), (size_t)kMaxBatch * kc * sizeof(float)));
        MX_HIP(hipMalloc(reinterpret_cast<void **>(&idx->sh_nf[g]), kMaxBatch * sizeof(int32_t)));
    }
    idx->sh_kcap = kc;
    return MX_OK;
}

// local row count of shard g when the composite holds `total` rows
uint64_t shard_rows(uint64_t total, uint64_t R, uint64_t G, uint64_t g) {
    const uint64_t full_blocks = total / R, tail = total % R;
    uint64_t rows = (full_blocks / G) * R + (g < full_blocks % G ? R : 0);
    if (full_blocks % G == g) rows += tail;
    return rows;
}

// peer copies between the shards' devices and shards[0]'s
void enable_peer_access(mx_index *idx) {
    for (size_t g = 1; g < idx->shards.size(); ++g) {
        if (idx->shards[g]->device == idx->shards[0]->device) continue;
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, idx->shards[0]->device, idx->shards[g]->device);
        if (can) {
            DeviceGuard dg(idx->shards[0]->device);
            (void)hipDeviceEnablePeerAccess(idx->shards[g]->device, 0);
            (void)hipGetLastError();
            DeviceGuard d2(idx->shards[g]->device);
            (void)hipDeviceEnablePeerAccess(idx->shards[0]->device, 0);
            (void)hipGetLastError();
        }
    }
}

// One tiny all-gather on throw-away streams, bounded by a deadline: a communicator that initialises but whose first
// collective fails or never completes (a first run on a new node) must cost a sharded index its RCCL exchange, not its
// searches.  false: do not use the communicators (a hung collective's streams and buffers are abandoned, not destroyed).
bool rccl_selftest(mx_index *idx, double seconds) {
    const int G = (int)idx->shards.size();
    std::vector<hipStream_t> st(G, nullptr);
    std::vector<unsigned char *> snd(G, nullptr), rcv(G, nullptr);
    bool ok = true;
    for (int g = 0; g < G && ok; ++g) {
        DeviceGuard dg(idx->shards[g]->device);
        ok = hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking) == hipSuccess && hipMalloc(&snd[g], 64) == hipSuccess &&
             hipMalloc(&rcv[g], 64 * (size_t)G) == hipSuccess && hipMemsetAsync(snd[g], g + 1, 64, st[g]) == hipSuccess &&
             hipMemsetAsync(rcv[g], 0, 64 * (size_t)G, st[g]) == hipSuccess;
    }
    bool hung = false;
    if (ok) {
        int e = g_rccl.GroupStart();
        for (int g = 0; g < G && e == 0; ++g) e = g_rccl.AllGather(snd[g], rcv[g], 64, 1 /*ncclUint8*/, idx->comms[g], st[g]);
        const int e2 = g_rccl.GroupEnd();
        ok = e == 0 && e2 == 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int g = 0; g < G && ok && !hung; ++g) {
            DeviceGuard dg(idx->shards[g]->device);
            for (;;) {
                const hipError_t q = hipStreamQuery(st[g]);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) {
                    ok = false;
                    break;
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) {
                    hung = true;
                    break;
                }
                struct timespec ts = {0, 200 * 1000};
                (void)clock_nanosleep(CLOCK_MONOTONIC, 0, &ts, nullptr);
            }
        }
        if (ok && !hung) {
            std::vector<unsigned char> h(64 * (size_t)G);
            DeviceGuard dg(idx->shards[0]->device);
            ok = hipMemcpy(h.data(), rcv[0], h.size(), hipMemcpyDeviceToHost) == hipSuccess;
            for (int g = 0; g < G && ok; ++g) ok = h[64 * (size_t)g] == (unsigned char)(g + 1) && h[64 * (size_t)g + 63] == (unsigned char)(g + 1);
        }
    }
    if (hung) {
        fprintf(stderr, "memex-hip: the RCCL self-test did not complete within %.0f s; exchanging by peer copies\n", seconds);
        return false;  // the collective may still own the streams and buffers: leave them
    }
    for (int g = 0; g < G; ++g) {
        DeviceGuard dg(idx->shards[g]->device);
        if (snd[g]) (void)hipFree(snd[g]);
        if (rcv[g]) (void)hipFree(rcv[g]);
        if (st[g]) (void)hipStreamDestroy(st[g]);
    }
    (void)hipGetLastError();
    if (!ok) fprintf(stderr, "memex-hip: the RCCL self-test failed; exchanging by peer copies\n");
    return ok;
}

// one batch on a composite: d_q and the outputs live on shards[0]'s device
int composite_batch(mx_index *idx, const float *d_q, int B, int k, uint64_t *d_ids, float *d_scores, float *d_dists,
                    int32_t *d_nfound) {
    const int G = (int)idx->shards.size();
    int rc = ensure_composite_buffers(idx, std::max(k, 1));
    if (rc != MX_OK) return rc;
    const size_t ids_bytes = (size_t)B * k * sizeof(uint64_t), blk = ids_bytes + (size_t)B * k * sizeof(float);
    std::vector<int> rcs(G, MX_OK);
    std::vector<std::string> errs(G);
    auto local = [&](int g) {
        mx_index *sh = idx->shards[g];
        std::lock_guard<std::mutex> lk(sh->mu);
        DeviceGuard dg(sh->device);
        auto run = [&]() -> int {
            MX_HIP(hipMemcpyAsync(idx->sh_q[g], d_q, (size_t)B * idx->dim * sizeof(float), hipMemcpyDefault, sh->stream));
            char *blkp = static_cast<char *>(idx->sh_block[g]);
            int r = search_batch(sh, idx->sh_q[g], B, k, reinterpret_cast<uint64_t *>(blkp), idx->sh_scores[g],
                                 reinterpret_cast<float *>(blkp + ids_bytes), idx->sh_nf[g]);
            if (r != MX_OK) return r;
            if (!idx->use_rccl && k > 0) {  // peer copy into slot g of the gather area on shards[0]'s device
                MX_HIP(hipMemcpyAsync(static_cast<char *>(idx->sh_gather[0]) + (size_t)g * blk, blkp, blk, hipMemcpyDefault, sh->stream));
                MX_HIP(hipStreamSynchronize(sh->stream));
            }
            return MX_OK;
        };
        try {  // (a helper thread of the pool: an exception must not leave it)
            rcs[g] = run();
        } catch (...) {
            rcs[g] = guard_exception();
        }
        if (rcs[g] != MX_OK) errs[g] = last_error_slot();
    };
    // the query batch must be complete on shards[0]'s stream before other devices read it
    {
        DeviceGuard dg(idx->shards[0]->device);
        MX_HIP(hipStreamSynchronize(idx->shards[0]->stream));
    }
    if (idx->pool) {  // one persistent helper thread per shard >= 1 (shard_pool.h); shard 0 on this thread
        idx->pool->run(local);
    } else {          // logical shards on one device: their streams would only take turns on the GPU anyway
        for (int g = 0; g < G; ++g) local(g);
    }
    for (int g = 0; g < G; ++g)
        if (rcs[g] != MX_OK) {
            last_error_slot() = errs[g];
            return rcs[g];
        }
    mx_index *s0 = idx->shards[0];
    DeviceGuard dg(s0->device);
    const auto t_tail = std::chrono::steady_clock::now();  // every shard has answered: what follows is the step's serial tail
    const bool copies_pending = idx->use_rccl;  // the shards left their blocks in place for the all-gather
    if (k > 0) {
        if (idx->use_rccl) {
            // ONE all-gather of B*k*12 bytes per shard over xGMI; every device receives all
            // blocks, device 0 merges
            int e = 0, e2 = 0;
#ifdef MEMEX_TESTING  // libmemex_hip_testing.so only: the first all-gather of the process reports an error
            static std::atomic<bool> injected{false};
            if (!injected.exchange(true)) {
                e = 1;
            } else
#endif
            {
                e = g_rccl.GroupStart();
                for (int g = 0; g < G && e == 0; ++g)
                    e = g_rccl.AllGather(idx->sh_block[g], idx->sh_gather[g], blk, 1 /*ncclUint8*/, idx->comms[g], idx->shards[g]->stream);
                e2 = g_rccl.GroupEnd();
            }
            if (e != 0 || e2 != 0) {
                // The collective could not be queued: this batch and every later one exchange by copies into the slots
                // of the gather area on shards[0]'s device (what an index without RCCL does from the start).  Every shard
                // is host-synchronised at this point (search_batch returned), so the blocks are complete.
                fprintf(stderr, "memex-hip: RCCL all-gather failed (%s); the sharded index continues on peer copies\n",
                        g_rccl.GetErrorString && (e > 1 || e2) ? g_rccl.GetErrorString(e ? e : e2) : "error");
                // (part of the group may have been queued on some shard streams before the failure: wait with a deadline, and
                // give the batch up rather than hang the caller if a stream never drains)
                if (!sync_shard_streams_within(idx, 10.0))
                    return fail(MX_EDEVICE, "RCCL all-gather failed and a shard stream did not drain within 10 s");
                idx->use_rccl = false;
                idx->stats.exchange_fallbacks += 1;
                enable_peer_access(idx);
            }
            // no host wait on shards >= 1: their part of the collective is ordered on their own streams (the
            // next batch's kernels queue behind it), and the merge below follows shard 0's part in stream order
        }
        if (!idx->use_rccl && copies_pending) {  // (only right after a fallback: the shards did not copy their blocks themselves)
            for (int g = 0; g < G; ++g)
                MX_HIP(hipMemcpyAsync(static_cast<char *>(idx->sh_gather[0]) + (size_t)g * blk, idx->sh_block[g], blk, hipMemcpyDefault, s0->stream));
        }
        MX_HIP(launch_merge(s0->stream, idx->sh_gather[0], blk, static_cast<const char *>(idx->sh_gather[0]) + ids_bytes, blk, G, B,
                            k, d_ids, d_dists ? d_dists : reinterpret_cast<float *>(static_cast<char *>(idx->sh_block[0]) + ids_bytes),
                            d_scores));
    }
    MX_HIP(launch_fill_nfound(s0->stream, d_nfound, B, (int32_t)std::min<uint64_t>((uint64_t)k, idx->total)));
    MX_HIP(hipStreamSynchronize(s0->stream));
    idx->stats.exchange_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tail).count();
    idx->stats.searches += 1;
    idx->stats.queries += (uint64_t)B;
    return MX_OK;
}

int any_batch(mx_index *idx, const float *d_q, int B, int k, uint64_t *d_ids, float *d_scores, float *d_dists,
              int32_t *d_nfound) {
    return idx->composite() ? composite_batch(idx, d_q, B, k, d_ids, d_scores, d_dists, d_nfound)
                            : search_batch(idx, d_q, B, k, d_ids, d_scores, d_dists, d_nfound);
}

// what an append can change in a plain index, and how to undo it: an insert is all-or-nothing, also when it
// spans several shards or several staging chunks and a later part fails (non-finite device rows, HBM)
struct RowMark {
    uint64_t n, n_zero, wild_rows, n_wild;
};
RowMark mark_rows(const mx_index *idx) { return RowMark{idx->n, idx->n_zero, idx->wild_rows, idx->n_wild}; }
void rollback_rows(mx_index *idx, const RowMark &m) {
    if (idx->n == m.n) return;
    const std::string keep = last_error_slot();
    DeviceGuard dg(idx->device);
    idx->n = m.n;  // rows past n are never read by a search (finish_kernel drops them); the next append overwrites them
    idx->n_zero = m.n_zero;
    idx->wild_rows = m.wild_rows;
    idx->n_wild = m.n_wild;
    const uint32_t zc[2] = {(uint32_t)m.n_zero, (uint32_t)m.n_wild};  // the device-side list counts: entries past them are dead
    (void)hipMemcpyAsync(idx->flags + 3, zc, sizeof(zc), hipMemcpyHostToDevice, idx->stream);
    (void)hipStreamSynchronize(idx->stream);
    idx->disk_dir.clear();
    last_error_slot() = keep;
}

// append host rows to a composite: global row r -> block b = r / R, shard b % G
int composite_add(mx_index *idx, const float *rows, uint64_t n, uint64_t *first_id, bool on_device) {
    const uint64_t R = idx->block_rows, G = idx->shards.size();
    if (first_id) *first_id = idx->total + 1;
    if (n == 0) return MX_OK;
    const size_t rowf = (size_t)idx->dim;
    // per shard: the (contiguous in the source) segments it receives, in order
    std::vector<std::vector<std::pair<uint64_t, uint64_t>>> seg(G);  // (source row, count)
    for (uint64_t r = idx->total, done = 0; done < n;) {
        const uint64_t b = r / R, take = std::min(n - done, R - r % R);
        seg[b % G].push_back({done, take});
        r += take;
        done += take;
    }
    // all-or-nothing: validate host rows up front (device rows are validated per shard by the ingest kernel)
    if (!on_device) {
        const size_t totalf = (size_t)n * rowf;
        for (size_t i = 0; i < totalf; ++i)
            if (!std::isfinite(rows[i])) return fail(MX_EINVAL, "row %zu contains a non-finite value; nothing inserted", i / rowf);
    }
    std::vector<RowMark> marks;
    for (mx_index *sh : idx->shards) marks.push_back(mark_rows(sh));
    auto append_to = [&](uint64_t g) -> int {
        mx_index *sh = idx->shards[g];
        std::lock_guard<std::mutex> lk(sh->mu);
        DeviceGuard dg(sh->device);
        uint64_t cnt = 0;
        for (auto &sgm : seg[g]) cnt += sgm.second;
        float *stage = nullptr;
        MX_HIP(hipMalloc(reinterpret_cast<void **>(&stage), (size_t)cnt * rowf * sizeof(float)));
        DevBuf hold;
        hold.p = stage;
        uint64_t at = 0;
        for (auto &sgm : seg[g]) {
            MX_HIP(hipMemcpyAsync(stage + (size_t)at * rowf, rows + (size_t)sgm.first * rowf, (size_t)sgm.second * rowf * sizeof(float),
                                  on_device ? hipMemcpyDefault : hipMemcpyHostToDevice, sh->stream));
            at += sgm.second;
        }
        return add_device_locked(sh, stage, cnt, nullptr);
    };
    for (uint64_t g = 0; g < G; ++g) {
        if (seg[g].empty()) continue;
        const int rc = append_to(g);
        if (rc != MX_OK) {  // a non-finite device row or an allocation on shard g: take back what shards < g got
            for (uint64_t h = 0; h < g; ++h) {
                std::lock_guard<std::mutex> lk(idx->shards[h]->mu);
                rollback_rows(idx->shards[h], marks[h]);
            }
            return rc;
        }
    }
    idx->total += n;
    return MX_OK;
}

const char kMagic[8] = {'M', 'X', 'F', 'L', 'A', 'T', '0', '1'};
constexpr long kHeaderBytes = 24;
std::string store_file(const char *dir) { return std::string(dir) + "/vectors.mxflat"; }

int mkdir_p(const std::string &dir) {  // create_dir_all (local.rs:144)
    struct stat sb;
    if (stat(dir.c_str(), &sb) == 0) return S_ISDIR(sb.st_mode) ? 0 : -1;
    const size_t slash = dir.find_last_of('/');
    if (slash != std::string::npos && slash > 0 && mkdir_p(dir.substr(0, slash)) != 0) return -1;
    return (mkdir(dir.c_str(), 0755) == 0 || errno == EEXIST) ? 0 : -1;
}

void remember_disk(mx_index *idx, const char *dir, uint64_t rows) {
    struct stat sb;
    idx->disk_dir.clear();
    if (stat(store_file(dir).c_str(), &sb) != 0) return;
    idx->disk_dir = dir;
    idx->disk_rows = rows;
    idx->disk_size = sb.st_size;
    idx->disk_mtime = sb.st_mtim;
}

bool disk_in_sync(mx_index *idx, const char *dir) {  // vectors.mxflat in dir is what this handle last wrote / read
    if (idx->disk_dir.empty() || idx->disk_dir != dir) return false;
    struct stat sb;
    if (stat(store_file(dir).c_str(), &sb) != 0) return false;
    return sb.st_size == idx->disk_size && sb.st_mtim.tv_sec == idx->disk_mtime.tv_sec &&
           sb.st_mtim.tv_nsec == idx->disk_mtime.tv_nsec;
}

// local rows [r0, r0 + m) of a plain index -> host buffer [m, dim].  A compressed corpus hands out its stored
// (bf16, near-unit) rows widened to f32.
int fetch_local(mx_index *idx, uint64_t r0, uint64_t m, float *out) {
    DeviceGuard dg(idx->device);
    if (!idx->compressed) {
        MX_HIP(hipMemcpy2D(out, (size_t)idx->dim * 4, idx->x + (size_t)r0 * idx->ds, (size_t)idx->ds * 4, (size_t)idx->dim * 4, m,
                           hipMemcpyDeviceToHost));
        return MX_OK;
    }
    const uint64_t chunk = std::max<uint64_t>(1, (idx->xs_rows ? idx->xs_rows : 1) * (uint64_t)idx->ds / (uint64_t)idx->dim);
    if (!idx->xs) return fail(MX_EINVAL, "empty compressed index");
    for (uint64_t done = 0; done < m; done += chunk) {
        const uint64_t c = std::min(chunk, m - done);
        MX_HIP(launch_unshadow(idx->stream, idx->xh, idx->ds, idx->dim, r0 + done, c, idx->xs));
        MX_HIP(hipMemcpyAsync(out + (size_t)done * idx->dim, idx->xs, (size_t)c * idx->dim * 4, hipMemcpyDeviceToHost, idx->stream));
        MX_HIP(hipStreamSynchronize(idx->stream));
    }
    return MX_OK;
}

// rows [r0, r0 + m) of the index in global order -> host buffer [m, dim]
int fetch_rows(mx_index *idx, uint64_t r0, uint64_t m, float *out, std::vector<float> &tmp) {
    (void)tmp;
    if (!idx->composite()) return fetch_local(idx, r0, m, out);
    const uint64_t R = idx->block_rows, G = idx->shards.size();
    for (uint64_t r = r0, done = 0; done < m;) {
        const uint64_t b = r / R, take = std::min(m - done, R - r % R);
        mx_index *sh = idx->shards[b % G];
        const uint64_t lrow = (b / G) * R + r % R;
        int rc = fetch_local(sh, lrow, take, out + (size_t)done * idx->dim);
        if (rc != MX_OK) return rc;
        r += take;
        done += take;
    }
    return MX_OK;
}

uint64_t rows_of(mx_index *idx) { return idx->composite() ? idx->total : idx->n; }
bool is_compressed(mx_index *idx) { return idx->composite() ? idx->shards[0]->compressed : idx->compressed; }
void set_raw_ingest(mx_index *idx, bool on) {
    idx->raw_ingest = on;
    for (mx_index *sh : idx->shards) sh->raw_ingest = on;
}

int clear_locked(mx_index *idx) {
    if (idx->composite()) {
        for (mx_index *sh : idx->shards) {
            std::lock_guard<std::mutex> lk(sh->mu);
            clear_locked(sh);
        }
        idx->total = 0;
    } else {
        idx->n = 0;  // ids restart at 1 (local.rs:50,63); HBM is kept for reuse
        idx->centred = false;  // (the centre belonged to the rows that are gone: appends refill the copy uncentred)
        idx->plain_bf16_rows = 0;
        idx->wild_rows = 0;
        idx->n_zero = 0;
        idx->n_wild = 0;
        for (double &w : idx->wait_ema_us) w = 0.0;
        if (idx->flags) {
            DeviceGuard dg(idx->device);
            (void)hipMemsetAsync(idx->flags + 2, 0, 4 * sizeof(uint32_t), idx->stream);  // ec_max, zero-row count, listed-row count, rc_max
        }
    }
    idx->disk_dir.clear();
    return MX_OK;
}

int add_host_locked(mx_index *idx, const float *rows, uint64_t n, uint64_t *first_id) {
    if (idx->composite()) return composite_add(idx, rows, n, first_id, false);
    DeviceGuard g(idx->device);
    if (n == 0) return add_device_locked(idx, nullptr, 0, first_id);
    // validate on the host first so that a rejected call leaves the index untouched
    const size_t total = (size_t)n * idx->dim;
    for (size_t i = 0; i < total; ++i)
        if (!std::isfinite(rows[i])) return fail(MX_EINVAL, "row %zu contains a non-finite value; nothing inserted", i / idx->dim);
    const uint64_t chunk_rows = std::max<uint64_t>(1, (64ull << 20) / ((uint64_t)idx->dim * 4));
    float *stage = nullptr;
    MX_HIP(hipMalloc(&stage, (size_t)std::min(chunk_rows, n) * idx->dim * sizeof(float)));
    int rc = MX_OK;
    uint64_t first = 0;
    const RowMark mark = mark_rows(idx);
    for (uint64_t done = 0; done < n && rc == MX_OK; done += chunk_rows) {
        const uint64_t m = std::min(chunk_rows, n - done);
        hipError_t e = hipMemcpyAsync(stage, rows + (size_t)done * idx->dim, (size_t)m * idx->dim * sizeof(float),
                                      hipMemcpyHostToDevice, idx->stream);
        if (e != hipSuccess) {
            rc = fail(MX_EDEVICE, "hipMemcpy H2D: %s", hipGetErrorString(e));
            break;
        }
        uint64_t f = 0;
        rc = add_device_locked(idx, stage, m, &f);
        if (done == 0) first = f;
    }
    (void)hipFree(stage);
    if (rc != MX_OK) rollback_rows(idx, mark);  // a later chunk failed: the earlier ones go too
    if (rc == MX_OK && first_id) *first_id = first;
    return rc;
}

int open_plain(const std::string &k, int dim, int device, mx_index **out) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(MX_EDEVICE, "no HIP device available (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(MX_EDEVICE, "device %d out of range (have %d)", device, ndev);
    if (device >= kMaxDevices) return fail(MX_EUNSUPPORTED, "device %d: at most %d devices per process", device, kMaxDevices);
    DeviceGuard g(device);
    if (!g.ok) return fail(MX_EDEVICE, "hipSetDevice(%d) failed", device);
    std::call_once(g_scan_once, [] {
        g_scan_setup_err = scan_setup();
        if (g_scan_setup_err == hipSuccess) g_scan_setup_err = scan16_setup();
        if (g_scan_setup_err == hipSuccess) g_scan_setup_err = scan16w_setup();
        if (g_scan_setup_err == hipSuccess) g_scan_setup_err = scan8_setup();
        if (g_scan_setup_err == hipSuccess) g_scan_setup_err = finish_setup();
    });
    if (g_scan_setup_err != hipSuccess)
        return fail(MX_EDEVICE, "scan kernel setup failed: %s (is this a gfx950 device?)", hipGetErrorString(g_scan_setup_err));
    std::unique_ptr<mx_index> idx(new mx_index());
    idx->key = k;
    idx->dim = dim;
    // stored row width: a multiple of 128 dims (one scan slot); wide rows (> 768) of 256, so that the two waves
    // that share a row's k-steps in scan16w_kernel get whole slots each
    idx->ds = (int)round_up((uint64_t)dim, dim > kMaxKC * kChunkFloats ? 2 * kChunkFloats : kChunkFloats);
    idx->kc = idx->ds / kChunkFloats;
    {   // which filter copy an f32 corpus keeps (mx_index_set_filter_copy overrides): MEMEX_HIP_FILTER=bf16|i8
        const char *fk = getenv("MEMEX_HIP_FILTER");
        idx->filter_auto = !(fk && fk[0]);
        idx->filter_i8 = idx->filter_auto ? idx->ds <= kAutoI8MaxDim : (fk[0] == 'i' || fk[0] == '8');
    }
    idx->device = device;
    hipDeviceProp_t prop;
    MX_HIP(hipGetDeviceProperties(&prop, device));
    idx->n_cu = prop.multiProcessorCount;
    idx->nwg = std::max(1, std::min(idx->n_cu, kMaxScanWGs));
    MX_HIP(hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking));
    MX_HIP(hipEventCreate(&idx->ev0));
    MX_HIP(hipEventCreate(&idx->ev1));
    MX_HIP(hipEventCreateWithFlags(&idx->ev_wait, hipEventDisableTiming));
    MX_HIP(hipMalloc(&idx->flags, 6 * sizeof(uint32_t)));
    MX_HIP(hipMemset(idx->flags, 0, 6 * sizeof(uint32_t)));
    MX_HIP(hipMalloc(&idx->zero_rows, kZeroCap * sizeof(uint32_t)));
    MX_HIP(hipMalloc(&idx->wild_list, kWildCap * sizeof(uint32_t)));
    if (device < kMaxDevices) {
        std::lock_guard<std::mutex> lk(g_lanes[device].mu);
        g_lanes[device].open_indexes += 1;
        idx->pooled = true;
    }
    *out = idx.release();
    return MX_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *mx_last_error(void) { return last_error_slot().c_str(); }
const char *mx_version(void) { return "memex-hip 0.5.0 (gfx950)"; }
size_t mx_index_stats_size(void) { return sizeof(mx_index_stats); }

int mx_device_count(int *n) try {
    if (!n) return fail(MX_EINVAL, "null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return fail(MX_EDEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n = c;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_open(const char *key, int dim, int device, mx_index **out) try {
    if (!out) return fail(MX_EINVAL, "out is null");
    *out = nullptr;
    if (dim < 1 || dim > (1 << 16)) return fail(MX_EINVAL, "dim %d out of range", dim);
    const std::string k = key ? key : "";
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (!k.empty()) {
        auto it = g_registry.find(k);
        if (it != g_registry.end()) {
            mx_index *idx = it->second;
            if (idx->dim != dim) return fail(MX_EINVAL, "index '%s' is open with dim %d, not %d", k.c_str(), idx->dim, dim);
            if (!idx->composite() && idx->device != device)
                return fail(MX_EINVAL, "index '%s' lives on device %d, not %d", k.c_str(), idx->device, device);
            idx->refs += 1;
            *out = idx;
            return MX_OK;
        }
    }
    mx_index *raw = nullptr;
    int rc = open_plain(k, dim, device, &raw);
    if (rc != MX_OK) return rc;
    if (!k.empty()) g_registry[k] = raw;
    *out = raw;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_open_sharded(const char *key, int dim, int n_dev, const int *devices, uint64_t block_rows, mx_index **out) try {
    if (!out) return fail(MX_EINVAL, "out is null");
    *out = nullptr;
    if (dim < 1 || dim > (1 << 16)) return fail(MX_EINVAL, "dim %d out of range", dim);
    if (n_dev < 1 || n_dev > 64) return fail(MX_EINVAL, "n_dev %d out of range", n_dev);
    const std::string k = key ? key : "";
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (!k.empty()) {
        auto it = g_registry.find(k);
        if (it != g_registry.end()) {
            mx_index *idx = it->second;
            if (idx->dim != dim) return fail(MX_EINVAL, "index '%s' is open with dim %d, not %d", k.c_str(), idx->dim, dim);
            if ((int)idx->shards.size() != n_dev) return fail(MX_EINVAL, "index '%s' is open with %zu shards, not %d", k.c_str(), idx->shards.size(), n_dev);
            idx->refs += 1;
            *out = idx;
            return MX_OK;
        }
    }
    std::unique_ptr<mx_index> idx(new mx_index());
    idx->key = k;
    idx->dim = dim;
    idx->block_rows = round_up(block_rows ? block_rows : 65536, kTileRows);
    bool distinct = n_dev > 1;
    for (int g = 0; g < n_dev; ++g) {
        const int dev = devices ? devices[g] : g;
        mx_index *sh = nullptr;
        int rc = open_plain("", dim, dev, &sh);
        if (rc != MX_OK) {
            for (mx_index *s2 : idx->shards) free_index(s2);
            return rc;
        }
        sh->idmap = IdMap{0, (uint32_t)idx->block_rows, (uint32_t)n_dev, (uint32_t)g};
        idx->shards.push_back(sh);
        for (int h = 0; h < g; ++h) distinct = distinct && idx->shards[h]->device != dev;
    }
    idx->device = idx->shards[0]->device;
    // exchange: RCCL all-gather when every shard has its own device (MEMEX_HIP_EXCHANGE=p2p forces
    // peer copies, =rccl insists on RCCL and fails without it); logical shards on one device use copies
    const char *ex = getenv("MEMEX_HIP_EXCHANGE");
    const bool want_rccl = ex ? strcmp(ex, "rccl") == 0 : distinct;
    const bool forbid_rccl = ex && strcmp(ex, "p2p") == 0;
    if (want_rccl && !forbid_rccl && (distinct || n_dev == 1)) {
        std::string why;
        if (g_rccl_broken) {
            why = "an earlier RCCL initialisation of this process hung";
        } else if (!load_rccl()) {
            why = "librccl.so.1 cannot be loaded";
        } else {
            // ncclCommInitAll under a watchdog (MEMEX_HIP_RCCL_TIMEOUT seconds, default 30): on a node where it never
            // returns, the thread is abandoned and this process exchanges by peer copies from here on
            struct Init {
                std::mutex mu;
                std::condition_variable cv;
                bool done = false;
                int rc = -1;
                std::vector<void *> comms;
                std::vector<int> devs;
            };
            auto init = std::make_shared<Init>();
            for (mx_index *sh : idx->shards) init->devs.push_back(sh->device);
            init->comms.assign(n_dev, nullptr);
            std::thread([init, n_dev] {
                const int rc = g_rccl.CommInitAll(init->comms.data(), n_dev, init->devs.data());
                std::lock_guard<std::mutex> l2(init->mu);
                init->rc = rc;
                init->done = true;
                init->cv.notify_all();
            }).detach();
            const char *tv = getenv("MEMEX_HIP_RCCL_TIMEOUT");
            const double secs = tv && atof(tv) > 0.0 ? atof(tv) : 30.0;
            std::unique_lock<std::mutex> l2(init->mu);
            if (!init->cv.wait_for(l2, std::chrono::duration<double>(secs), [&] { return init->done; })) {
                g_rccl_broken = true;
                why = "ncclCommInitAll did not return in time";
            } else if (init->rc != 0) {
                why = std::string("ncclCommInitAll failed: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(init->rc) : "?");
            } else {
                idx->comms = init->comms;
                if (rccl_selftest(idx.get(), secs)) {
                    idx->use_rccl = true;
                } else {
                    why = "the first all-gather failed or did not complete";
                    idx->comms.clear();  // (not destroyed: a communicator with a collective in an unknown state)
                }
            }
        }
        if (!idx->use_rccl) {
            if (ex) {  // MEMEX_HIP_EXCHANGE=rccl insists
                for (mx_index *s2 : idx->shards) free_index(s2);
                return fail(MX_EDEVICE, "MEMEX_HIP_EXCHANGE=rccl: %s", why.c_str());
            }
            fprintf(stderr, "memex-hip: sharded index '%s' exchanges by peer copies (%s)\n", k.c_str(), why.c_str());
        }
    }
    if (!idx->use_rccl && distinct) enable_peer_access(idx.get());
    {   // helper threads: one per shard >= 1 when every shard has its own device.  MEMEX_HIP_SHARD_THREADS=1
        // forces them for logical shards on one device too (how the hand-off is tested on a 1-GPU box), =0 never
        const char *tv = getenv("MEMEX_HIP_SHARD_THREADS");
        const bool threads = tv ? tv[0] == '1' : distinct;
        if (threads && n_dev > 1) idx->pool.reset(new ShardPool(n_dev - 1));
    }
    mx_index *raw = idx.release();
    if (!k.empty()) g_registry[k] = raw;
    *out = raw;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

// how the shards of `idx` exchange their top-k blocks: 0 = not sharded, 1 = copies into a slot per shard on
// devices[0] (peer-to-peer between devices), 2 = RCCL all-gather
int mx_index_exchange(mx_index *idx, int *kind) try {
    if (!idx || !kind) return fail(MX_EINVAL, "null argument");
    *kind = !idx->composite() ? 0 : (idx->use_rccl ? 2 : 1);
    return MX_OK;
} catch (...) {
    return guard_exception();
}

void mx_index_close(mx_index *idx) {
    if (!idx) return;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        if (--idx->refs > 0) return;
        if (!idx->key.empty()) g_registry.erase(idx->key);
    }
    free_index(idx);
}

int mx_index_dim(mx_index *idx, int *dim) try {
    if (!idx || !dim) return fail(MX_EINVAL, "null argument");
    *dim = idx->dim;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_n_shards(mx_index *idx, int *n) try {
    if (!idx || !n) return fail(MX_EINVAL, "null argument");
    *n = idx->composite() ? (int)idx->shards.size() : 1;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_size(mx_index *idx, uint64_t *n) try {
    if (!idx || !n) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    *n = rows_of(idx);
    return MX_OK;
} catch (...) {
    return guard_exception();
}

static int reserve_locked(mx_index *idx, uint64_t rows) {
    if (idx->composite()) {
        const uint64_t G = idx->shards.size();
        for (uint64_t g = 0; g < G; ++g) {
            mx_index *sh = idx->shards[g];
            std::lock_guard<std::mutex> l2(sh->mu);
            DeviceGuard dg(sh->device);
            int rc = ensure_capacity(sh, shard_rows(rows, idx->block_rows, G, g));
            if (rc != MX_OK) return rc;
        }
        return MX_OK;
    }
    DeviceGuard g(idx->device);
    return ensure_capacity(idx, rows);
}

int mx_index_reserve(mx_index *idx, uint64_t rows) try {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    return reserve_locked(idx, rows);
} catch (...) {
    return guard_exception();
}

int mx_index_set_id_offset(mx_index *idx, uint64_t off) try {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->idmap.id_offset = off;
    for (mx_index *sh : idx->shards) sh->idmap.id_offset = off;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_wait_stream(mx_index *idx, void *stream) try {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    mx_index *t = idx->composite() ? idx->shards[0] : idx;
    DeviceGuard g(t->device);
    // nothing pending on the caller's stream (the usual case between two searches): no event, no dependency to process --
    // one query call instead of a record + wait pair in front of every batch
    if (hipStreamQuery(static_cast<hipStream_t>(stream)) == hipSuccess) return MX_OK;
    (void)hipGetLastError();
    MX_HIP(hipEventRecord(t->ev_wait, static_cast<hipStream_t>(stream)));
    MX_HIP(hipStreamWaitEvent(t->stream, t->ev_wait, 0));
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_add_device(mx_index *idx, const float *d_rows, uint64_t n, uint64_t *first_id) try {
    if (!idx || (!d_rows && n)) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->composite()) {
        {   // rows must be complete on shards[0]'s stream before other devices copy them
            DeviceGuard dg(idx->shards[0]->device);
            MX_HIP(hipStreamSynchronize(idx->shards[0]->stream));
        }
        return composite_add(idx, d_rows, n, first_id, true);
    }
    DeviceGuard g(idx->device);
    return add_device_locked(idx, d_rows, n, first_id);
} catch (...) {
    return guard_exception();
}

int mx_index_add(mx_index *idx, const float *rows, uint64_t n, uint64_t *first_id) try {
    if (!idx || (!rows && n)) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    return add_host_locked(idx, rows, n, first_id);
} catch (...) {
    return guard_exception();
}

int mx_index_clear(mx_index *idx) try {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    return clear_locked(idx);
} catch (...) {
    return guard_exception();
}

int mx_index_set_search_mode(mx_index *idx, int mode) try {
    if (!idx) return fail(MX_EINVAL, "null index");
    if (mode != MX_SEARCH_AUTO && mode != MX_SEARCH_EXACT) return fail(MX_EINVAL, "unknown search mode %d", mode);
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->mode = mode;
    for (mx_index *sh : idx->shards) sh->mode = mode;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_search_device(mx_index *idx, const float *d_q, int B, int k, uint64_t *d_ids, float *d_scores,
                           float *d_dists, int32_t *d_nfound) try {
    if (!idx) return fail(MX_ESEARCH, "null index");
    if (B < 0 || k < 0) return fail(MX_EINVAL, "negative batch or k");
    if (B == 0) return MX_OK;
    if (!d_q || !d_nfound || (k > 0 && (!d_ids || !d_scores))) return fail(MX_EINVAL, "null argument");
    if (k > 4096) return fail(MX_EUNSUPPORTED, "k = %d > 4096", k);
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->device);
    for (int b0 = 0; b0 < B; b0 += kMaxBatch) {
        const int nb = std::min(kMaxBatch, B - b0);
        int rc = any_batch(idx, d_q + (size_t)b0 * idx->dim, nb, k, d_ids + (size_t)b0 * k,
                           d_scores + (size_t)b0 * k, d_dists ? d_dists + (size_t)b0 * k : nullptr, d_nfound + b0);
        if (rc != MX_OK) return rc;
    }
    return MX_OK;
} catch (...) {
    return guard_exception();
}

namespace {

// one GPU batch (sum of B <= 256, same k) for a group of host requests: queries are packed into
// pinned memory, one H2D, the search pipeline, one D2H per output array, results scattered to the callers
int run_combined(mx_index *idx, const std::vector<SearchReq *> &batch) {
    std::lock_guard<std::mutex> lk(idx->mu);
    mx_index *t = idx->composite() ? idx->shards[0] : idx;  // owner of the staging buffers and the stream
    DeviceGuard g(t->device);
    const int k = batch[0]->k;
    int rc = ensure_scratch(t);
    if (rc != MX_OK) return rc;
    rc = ensure_out(t, k);
    if (rc != MX_OK) return rc;
    Scratch &s = t->s;
    const size_t dim = (size_t)idx->dim;
    int nb = 0;
    for (const SearchReq *r : batch) {
        memcpy(s.h_q + (size_t)nb * dim, r->q, (size_t)r->B * dim * sizeof(float));
        nb += r->B;
    }
    // A plain index reads the queries from, and writes the answers into, the pinned staging buffers themselves (they are mapped
    // into the device's address space): prep_queries_kernel is the only reader of the raw queries, finish_kernel (or the EXACT
    // kernels) the only writers of the outputs, and every exit of search_batch is host-synchronised -- through the completion
    // word, behind a system-scope fence per workgroup, on the fast path.  That is one H2D copy, four D2H copies and a stream
    // synchronise less per call: 25-30 us of a single query's 95-115 us on a small collection (profiles/r5_small_corpus_latency.txt).
    // A sharded index merges on the device and copies as before.
    if (!idx->composite()) {
        s.out_on_host = true;
        rc = any_batch(idx, s.h_q, nb, k, s.h_ids, s.h_scores, s.h_dists, s.h_nf);
        s.out_on_host = false;
        if (rc != MX_OK) return rc;
        std::atomic_thread_fence(std::memory_order_acquire);
    } else {
    MX_HIP(hipMemcpyAsync(s.qstage, s.h_q, (size_t)nb * dim * sizeof(float), hipMemcpyHostToDevice, t->stream));
    rc = any_batch(idx, s.qstage, nb, k, s.out_ids, s.out_scores, s.out_dists, s.out_nfound);
    if (rc != MX_OK) return rc;
    if (k > 0) {
        MX_HIP(hipMemcpyAsync(s.h_ids, s.out_ids, (size_t)nb * k * sizeof(uint64_t), hipMemcpyDeviceToHost, t->stream));
        MX_HIP(hipMemcpyAsync(s.h_scores, s.out_scores, (size_t)nb * k * sizeof(float), hipMemcpyDeviceToHost, t->stream));
        MX_HIP(hipMemcpyAsync(s.h_dists, s.out_dists, (size_t)nb * k * sizeof(float), hipMemcpyDeviceToHost, t->stream));
    }
    MX_HIP(hipMemcpyAsync(s.h_nf, s.out_nfound, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, t->stream));
    MX_HIP(hipStreamSynchronize(t->stream));
    }
    int b0 = 0;
    for (SearchReq *r : batch) {
        if (k > 0) {
            memcpy(r->ids, s.h_ids + (size_t)b0 * k, (size_t)r->B * k * sizeof(uint64_t));
            memcpy(r->scores, s.h_scores + (size_t)b0 * k, (size_t)r->B * k * sizeof(float));
            if (r->dists) memcpy(r->dists, s.h_dists + (size_t)b0 * k, (size_t)r->B * k * sizeof(float));
        }
        memcpy(r->n_found, s.h_nf + b0, (size_t)r->B * sizeof(int32_t));
        b0 += r->B;
    }
    return MX_OK;
}

}  // namespace

// The reference serves one query per HTTP request on a multi-threaded runtime (handlers.rs:55-109):
// calls arrive concurrently, each with B = 1.  A GPU pass over the corpus costs the same for 1 and for
// 256 queries, so concurrent callers are COMBINED: every call queues its request; whoever finds no
// leader becomes the leader and serves batches (up to 256 queries with the same k, FIFO) until its
// own request is done, then hands over.  A lone caller runs immediately (no timer, no added
// latency); under load the batch is whatever queued up while the previous pass was on the GPU.
int mx_index_search(mx_index *idx, const float *q, int B, int k, uint64_t *ids, float *scores, float *dists,
                    int32_t *n_found) try {
    if (!idx) return fail(MX_ESEARCH, "null index");
    if (B < 0 || k < 0) return fail(MX_EINVAL, "negative batch or k");
    if (B == 0) return MX_OK;
    if (!q || !n_found || (k > 0 && (!ids || !scores))) return fail(MX_EINVAL, "null argument");
    if (k > 4096) return fail(MX_EUNSUPPORTED, "k = %d > 4096", k);
    {   // reject non-finite queries here, per caller: inside a combined batch they would fail everyone
        const size_t total = (size_t)B * idx->dim;
        for (size_t i = 0; i < total; ++i)
            if (!std::isfinite(q[i])) return fail(MX_EINVAL, "query %zu contains a non-finite value", i / idx->dim);
    }
    if (B > kMaxBatch) {  // large requests are their own batches: split and recurse
        for (int b0 = 0; b0 < B; b0 += kMaxBatch) {
            const int nb = std::min(kMaxBatch, B - b0);
            int rc = mx_index_search(idx, q + (size_t)b0 * idx->dim, nb, k, ids ? ids + (size_t)b0 * k : nullptr,
                                     scores ? scores + (size_t)b0 * k : nullptr,
                                     dists ? dists + (size_t)b0 * k : nullptr, n_found + b0);
            if (rc != MX_OK) return rc;
        }
        return MX_OK;
    }
    SearchReq req{q, B, k, ids, scores, dists, n_found};
    std::unique_lock<std::mutex> ql(idx->cmu);
    idx->pending.push_back(&req);
    idx->ccv.wait(ql, [&] { return req.done || !idx->leader; });
    if (!req.done) {
        idx->leader = true;
        while (!req.done) {
            // FIFO batch: the oldest request decides k; later requests with the same k join while they fit
            std::vector<SearchReq *> batch;
            int total = 0;
            const int bk = idx->pending.front()->k;
            for (auto it = idx->pending.begin(); it != idx->pending.end();) {
                SearchReq *r = *it;
                if (r->k == bk && total + r->B <= kMaxBatch) {
                    batch.push_back(r);
                    total += r->B;
                    it = idx->pending.erase(it);
                } else {
                    ++it;
                }
            }
            ql.unlock();
            int rc;
            try {  // the leader answers for the others: an exception must reach them as an error code, not leave them waiting
                rc = run_combined(idx, batch);
            } catch (...) {
                rc = guard_exception();
            }
            const std::string err = rc == MX_OK ? std::string() : last_error_slot();
            ql.lock();
            for (SearchReq *r : batch) {
                r->rc = rc;
                r->err = err;
                r->done = true;
            }
            idx->ccv.notify_all();
        }
        idx->leader = false;
        idx->ccv.notify_all();  // a waiter (if any) takes over
    }
    ql.unlock();
    if (req.rc != MX_OK) last_error_slot() = req.err;  // the leader's message, in the caller's thread
    return req.rc;
} catch (...) {
    return guard_exception();
}

int mx_index_set_filter_copy(mx_index *idx, int on) try {
    if (!idx) return fail(MX_EINVAL, "null index");
    if (on < 0 || on > 3) return fail(MX_EINVAL, "filter copy: 0 = none, 1 = kind chosen by the library, 2 = int8, 3 = bf16 (got %d)", on);
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->composite()) {
        for (mx_index *sh : idx->shards) {
            int rc = mx_index_set_filter_copy(sh, on);
            if (rc != MX_OK) return rc;
        }
        return MX_OK;
    }
    DeviceGuard g(idx->device);
    if (idx->compressed) return on ? MX_OK : fail(MX_EINVAL, "a compressed corpus has no f32 rows to fall back to");
    const bool i8 = on == 2 || (on == 1 && idx->ds <= kAutoI8MaxDim);
    auto drop = [&]() -> int {
        if (idx->xh) {
            MX_HIP(hipStreamSynchronize(idx->stream));
            (void)hipFree(idx->xh);
            if (idx->tsc) (void)hipFree(idx->tsc);
            if (idx->amean) (void)hipFree(idx->amean);
            idx->xh = nullptr;
            idx->tsc = nullptr;
            idx->amean = nullptr;
            idx->centred = false;
        }
        return MX_OK;
    };
    idx->want_filter = on != 0;
    if (!on) return drop();
    if (idx->xh && idx->filter_i8 != i8) {  // the other kind is resident: replace it
        int rc = drop();
        if (rc != MX_OK) return rc;
    }
    idx->filter_i8 = i8;
    idx->filter_auto = on == 1;
    idx->i8_batches = idx->i8_retry_batches = 0;
    idx->demoted_at_rows = 0;
    for (double &w : idx->wait_ema_us) w = 0.0;
    if (idx->xh || idx->cap == 0 || idx->kc > kMaxKC16) return MX_OK;  // present, or built with the first rows
    return build_filter_copy(idx, i8);
} catch (...) {
    return guard_exception();
}

int mx_index_set_corpus_mode(mx_index *idx, int mode) try {
    if (!idx) return fail(MX_EINVAL, "null index");
    if (mode != MX_CORPUS_F32 && mode != MX_CORPUS_BF16) return fail(MX_EINVAL, "unknown corpus mode %d", mode);
    std::lock_guard<std::mutex> lk(idx->mu);
    if (rows_of(idx) != 0) return fail(MX_EINVAL, "the corpus mode can only be chosen while the index is empty");
    if (idx->composite()) {
        for (mx_index *sh : idx->shards) {
            int rc = mx_index_set_corpus_mode(sh, mode);
            if (rc != MX_OK) return rc;
        }
        return MX_OK;
    }
    if (mode == MX_CORPUS_BF16 && idx->kc > kMaxKC16)
        return fail(MX_EUNSUPPORTED, "a compressed corpus supports dim <= %d", kMaxKC16 * kChunkFloats);
    DeviceGuard g(idx->device);
    MX_HIP(hipStreamSynchronize(idx->stream));
    auto F = [](void *p) {
        if (p) (void)hipFree(p);
    };
    F(idx->x); F(idx->scale); F(idx->xh); F(idx->tsc); F(idx->amean);
    idx->x = nullptr; idx->scale = nullptr; idx->xh = nullptr; idx->tsc = nullptr; idx->amean = nullptr;
    idx->centred = false;
    idx->cap = 0;
    idx->compressed = mode == MX_CORPUS_BF16;
    idx->want_filter = true;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_get_rows(mx_index *idx, uint64_t first_row, uint64_t n, float *out) try {
    if (!idx || (!out && n)) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (first_row + n > rows_of(idx)) return fail(MX_EINVAL, "rows [%llu, %llu) outside the index", (unsigned long long)first_row,
                                                 (unsigned long long)(first_row + n));
    if (n == 0) return MX_OK;
    std::vector<float> tmp;
    return fetch_rows(idx, first_row, n, out, tmp);
} catch (...) {
    return guard_exception();
}

int mx_index_set_profiling(mx_index *idx, int on) try {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->profiling = on != 0;
    for (mx_index *sh : idx->shards) sh->profiling = on != 0;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_get_stats(mx_index *idx, mx_index_stats *out) try {
    if (!idx || !out) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->composite()) {
        mx_index_stats acc = idx->stats;  // searches / queries are counted on the composite
        for (mx_index *sh : idx->shards) {
            mx_index_stats s1;
            int rc = mx_index_get_stats(sh, &s1);
            if (rc != MX_OK) return rc;
            acc.fallback_queries += s1.fallback_queries;
            acc.retry_queries += s1.retry_queries;
            acc.scan_launches += s1.scan_launches;
            acc.scan_bytes += s1.scan_bytes;
            acc.scan_ms += s1.scan_ms;
            acc.candidates += s1.candidates;
            acc.max_abs_err = std::max(acc.max_abs_err, s1.max_abs_err);
            acc.approx_err_bound = std::max(acc.approx_err_bound, s1.approx_err_bound);
            acc.filter_copy_bytes += s1.filter_copy_bytes;
            acc.filter_kind = std::max(acc.filter_kind, s1.filter_kind);  // shards choose alike; a demoted one shows as bf16
            acc.filter_demotions += s1.filter_demotions;
            acc.filter_promotions += s1.filter_promotions;
            acc.listed_rows += s1.listed_rows;
            acc.filter_centred = std::max(acc.filter_centred, s1.filter_centred);
            // (exchange_ms is counted on the composite)
        }
        *out = acc;
        return MX_OK;
    }
    if (idx->s.max_err) {  // device-side running maximum (profiling mode); fetched on demand
        DeviceGuard g(idx->device);
        float e = 0.f;
        MX_HIP(hipMemcpy(&e, idx->s.max_err, sizeof(float), hipMemcpyDeviceToHost));
        idx->stats.max_abs_err = std::max(idx->stats.max_abs_err, (double)e);
    }
    if (idx->s.host_sum) {  // largest e1 of the last batch (all queries share Ec; Eq is the query's own)
        float e1 = 0.f;
        memcpy(&e1, idx->s.host_sum + 3, sizeof(float));
        idx->stats.approx_err_bound = e1;
    }
    idx->stats.filter_kind = !idx->xh ? 0u : (idx->filter_i8 && !idx->compressed ? 2u : 3u);
    idx->stats.filter_copy_bytes = idx->xh ? (uint64_t)idx->cap * idx->ds * (idx->filter_i8 && !idx->compressed ? 1ull : 2ull) : 0;
    idx->stats.listed_rows = idx->n_zero + idx->n_wild;
    idx->stats.filter_centred = idx->centred && idx->xh ? 1u : 0u;
    *out = idx->stats;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_reset_stats(mx_index *idx) try {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->stats = mx_index_stats{};
    for (mx_index *sh : idx->shards) (void)mx_index_reset_stats(sh);
    if (idx->s.max_err) {
        DeviceGuard g(idx->device);
        (void)hipMemset(idx->s.max_err, 0, sizeof(float));
    }
    return MX_OK;
} catch (...) {
    return guard_exception();
}

// ---- persistence (replaces hnsw file_dump / load_hnsw, local.rs:115-165) ----------------------
// vectors.mxflat: magic[8] | u32 dim | u32 0 | u64 n_rows | n_rows * dim f32 (row-major, global id
// order: the file does not depend on how many devices hold the index).  The reference saves after
// EVERY insert (local.rs:67); to make that affordable a save into the directory this handle last
// saved to / loaded from APPENDS the new rows and patches the header instead of rewriting the file.
int mx_index_save(mx_index *idx, const char *dir) try {
    if (!idx || !dir) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (mkdir_p(dir) != 0) return fail(MX_EIO, "cannot create directory %s", dir);
    const uint64_t n = rows_of(idx);
    const std::string path = store_file(dir);
    const uint64_t chunk = std::max<uint64_t>(1, (32ull << 20) / ((uint64_t)idx->dim * 4));
    std::vector<float> host((size_t)std::min<uint64_t>(chunk, std::max<uint64_t>(n, 1)) * idx->dim), tmpv;
    auto write_rows = [&](FILE *f, uint64_t r0) -> int {
        for (uint64_t r = r0; r < n; r += chunk) {
            const uint64_t m = std::min(chunk, n - r);
            int rc = fetch_rows(idx, r, m, host.data(), tmpv);
            if (rc != MX_OK) return rc;
            if (fwrite(host.data(), sizeof(float), (size_t)m * idx->dim, f) != (size_t)m * idx->dim)
                return fail(MX_EIO, "write to %s failed", path.c_str());
        }
        return MX_OK;
    };
    if (disk_in_sync(idx, dir) && idx->disk_rows <= n) {
        if (idx->disk_rows == n) return MX_OK;  // nothing new
        FILE *f = fopen(path.c_str(), "r+b");
        if (!f) return fail(MX_EIO, "cannot open %s for appending", path.c_str());
        int rc = fseek(f, kHeaderBytes + (long)(idx->disk_rows * (uint64_t)idx->dim * 4), SEEK_SET) == 0 ? MX_OK : fail(MX_EIO, "seek in %s failed", path.c_str());
        if (rc == MX_OK) rc = write_rows(f, idx->disk_rows);
        // the header is patched last: a crash before this point leaves the old, consistent store
        if (rc == MX_OK && (fflush(f) != 0 || fseek(f, 16, SEEK_SET) != 0 || fwrite(&n, sizeof(n), 1, f) != 1))
            rc = fail(MX_EIO, "write to %s failed", path.c_str());
        if (fclose(f) != 0 && rc == MX_OK) rc = fail(MX_EIO, "write to %s failed", path.c_str());
        if (rc == MX_OK) remember_disk(idx, dir, n);
        else idx->disk_dir.clear();
        return rc;
    }
    const std::string tmp = path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(MX_EIO, "cannot open %s for writing", tmp.c_str());
    uint32_t hdr[2] = {(uint32_t)idx->dim, is_compressed(idx) ? 1u : 0u};  // [1] = rows are the stored values of a compressed corpus
    int rc = (fwrite(kMagic, 1, 8, f) == 8 && fwrite(hdr, sizeof(hdr), 1, f) == 1 && fwrite(&n, sizeof(n), 1, f) == 1)
                 ? MX_OK : fail(MX_EIO, "write to %s failed", tmp.c_str());
    if (rc == MX_OK) rc = write_rows(f, 0);
    if (fclose(f) != 0 && rc == MX_OK) rc = fail(MX_EIO, "write to %s failed", tmp.c_str());
    if (rc == MX_OK && rename(tmp.c_str(), path.c_str()) != 0) rc = fail(MX_EIO, "cannot rename %s", tmp.c_str());
    if (rc != MX_OK) {
        unlink(tmp.c_str());
        return rc;
    }
    remember_disk(idx, dir, n);
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_load(mx_index *idx, const char *dir) try {
    if (!idx || !dir) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    const std::string path = store_file(dir);
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return fail(MX_EIO, "cannot open %s", path.c_str());
    char magic[8];
    uint32_t hdr[2];
    uint64_t n = 0;
    struct stat sb;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kMagic, 8) != 0 || fread(hdr, sizeof(hdr), 1, f) != 1 ||
        fread(&n, sizeof(n), 1, f) != 1 || fstat(fileno(f), &sb) != 0) {
        fclose(f);
        return fail(MX_EIO, "%s: bad header", path.c_str());
    }
    if ((int)hdr[0] != idx->dim) {
        fclose(f);
        return fail(MX_EIO, "%s holds dim %u, index has dim %d", path.c_str(), hdr[0], idx->dim);
    }
    // validate BEFORE touching the live contents: a truncated file must not destroy them
    if (n > (UINT64_MAX - (uint64_t)kHeaderBytes) / ((uint64_t)idx->dim * 4) ||  // (a damaged row count must not wrap the product)
        (uint64_t)sb.st_size < (uint64_t)kHeaderBytes + n * (uint64_t)idx->dim * 4) {
        fclose(f);
        return fail(MX_EIO, "%s: truncated (%lld bytes for %llu rows)", path.c_str(), (long long)sb.st_size, (unsigned long long)n);
    }
    // get_vector_storage loads the store on every request (storage/mod.rs:115-116): when the resident
    // rows are exactly what this file holds, attaching is O(1)
    if (disk_in_sync(idx, dir) && idx->disk_rows == n && rows_of(idx) == n) {
        fclose(f);
        return MX_OK;
    }
    clear_locked(idx);
    set_raw_ingest(idx, hdr[1] == 1);  // stored values of a compressed corpus go back in unchanged
    // Cold start (a collection after a restart): the file is read in 32 MB pieces into two PINNED buffers; a piece
    // goes to the device on a copy stream while the next one is being read and the previous one is ingested
    // (validated on the device like any device-row append): the load runs at the speed of the read, not at the
    // sum of read + pageable copy + host validation + ingest (round 2).  A sharded index takes its pieces
    // through composite_add (each shard validates its part).
    const uint64_t chunk = std::max<uint64_t>(1, (32ull << 20) / ((uint64_t)idx->dim * 4));
    const size_t chunk_bytes = (size_t)chunk * idx->dim * sizeof(float);
    mx_index *t0 = idx->composite() ? idx->shards[0] : idx;
    DeviceGuard dg(t0->device);
    float *pin[2] = {nullptr, nullptr}, *dev[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t cs = nullptr;
    int rc = MX_OK;
    auto release = [&] {
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (dev[i]) (void)hipFree(dev[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
        if (cs) (void)hipStreamDestroy(cs);
    };
    for (int i = 0; i < 2 && rc == MX_OK; ++i) {
        if (hipHostMalloc(reinterpret_cast<void **>(&pin[i]), chunk_bytes, hipHostMallocDefault) != hipSuccess ||
            (!idx->composite() && hipMalloc(reinterpret_cast<void **>(&dev[i]), chunk_bytes) != hipSuccess) ||
            hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess)
            rc = fail(MX_ENOMEM, "staging buffers for %s", path.c_str());
    }
    if (rc == MX_OK && hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) rc = fail(MX_EDEVICE, "stream creation failed");
    if (rc == MX_OK && n > 0) rc = reserve_locked(idx, n);
    uint64_t pending_rows = 0;  // rows of the piece that is on its way to dev[pending_buf]
    int pending_buf = 0;
    auto ingest_pending = [&]() -> int {
        if (!pending_rows) return MX_OK;
        MX_HIP(hipStreamWaitEvent(idx->stream, ev[pending_buf], 0));
        const int r = add_device_locked(idx, dev[pending_buf], pending_rows, nullptr);
        pending_rows = 0;
        return r;
    };
    int buf = 0;
    for (uint64_t r = 0; r < n && rc == MX_OK; r += chunk, buf ^= 1) {
        const uint64_t m = std::min(chunk, n - r);
        if (fread(pin[buf], sizeof(float), (size_t)m * idx->dim, f) != (size_t)m * idx->dim) {
            rc = fail(MX_EIO, "%s: read failed", path.c_str());
            break;
        }
        if (idx->composite()) {
            rc = composite_add(idx, pin[buf], m, nullptr, false);
            continue;
        }
        // dev[buf] was last read by the ingest of the piece before the previous one: that call returned synchronised
        hipError_t e = hipMemcpyAsync(dev[buf], pin[buf], (size_t)m * idx->dim * sizeof(float), hipMemcpyHostToDevice, cs);
        if (e == hipSuccess) e = hipEventRecord(ev[buf], cs);
        if (e != hipSuccess) {
            rc = fail(MX_EDEVICE, "hipMemcpy H2D: %s", hipGetErrorString(e));
            break;
        }
        rc = ingest_pending();  // the previous piece, while this one is in flight
        pending_rows = m;
        pending_buf = buf;
    }
    if (rc == MX_OK) rc = ingest_pending();
    if (cs) (void)hipStreamSynchronize(cs);
    release();
    fclose(f);
    set_raw_ingest(idx, false);
    if (rc != MX_OK) {
        const std::string keep = last_error_slot();
        clear_locked(idx);
        last_error_slot() = keep;
        return rc;
    }
    remember_disk(idx, dir, n);
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_has_store(const char *dir, int *exists) try {
    if (!dir || !exists) return fail(MX_EINVAL, "null argument");
    struct stat sb;
    *exists = stat(store_file(dir).c_str(), &sb) == 0 ? 1 : 0;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_store_info(const char *dir, int *dim, uint64_t *n_rows) try {
    if (!dir || !dim || !n_rows) return fail(MX_EINVAL, "null argument");
    FILE *f = fopen(store_file(dir).c_str(), "rb");
    if (!f) return fail(MX_EIO, "cannot open %s", store_file(dir).c_str());
    char magic[8];
    uint32_t hdr[2];
    uint64_t n = 0;
    const bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, kMagic, 8) == 0 && fread(hdr, sizeof(hdr), 1, f) == 1 &&
                    fread(&n, sizeof(n), 1, f) == 1;
    fclose(f);
    if (!ok) return fail(MX_EIO, "%s: bad header", store_file(dir).c_str());
    *dim = (int)hdr[0];
    *n_rows = n;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_index_remove_files(const char *dir) try {
    if (!dir) return fail(MX_EINVAL, "null argument");
    const std::string p = store_file(dir);
    struct stat sb;
    if (stat(p.c_str(), &sb) == 0 && unlink(p.c_str()) != 0) return fail(MX_EIO, "cannot remove %s", p.c_str());
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_topk_merge_device(int device, const uint64_t *d_ids, const float *d_dists, int G, int B, int k,
                         uint64_t *d_out_ids, float *d_out_dists, float *d_out_scores) try {
    if (G < 1 || B < 0 || k < 0) return fail(MX_EINVAL, "bad merge shape");
    if (B == 0 || k == 0) return MX_OK;
    if (!d_ids || !d_dists || !d_out_ids || !d_out_dists) return fail(MX_EINVAL, "null argument");
    DeviceGuard g(device);
    if (!g.ok) return fail(MX_EDEVICE, "hipSetDevice(%d) failed", device);
    MX_HIP(launch_merge(hipStreamPerThread, d_ids, (size_t)B * k * sizeof(uint64_t), d_dists, (size_t)B * k * sizeof(float),
                        G, B, k, d_out_ids, d_out_dists, d_out_scores));
    MX_HIP(hipStreamSynchronize(hipStreamPerThread));
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_topk_merge_packed_device(int device, const void *d_packed, int G, int B, int k, uint64_t *d_out_ids,
                                float *d_out_dists, float *d_out_scores) try {
    if (G < 1 || B < 0 || k < 0) return fail(MX_EINVAL, "bad merge shape");
    if (B == 0 || k == 0) return MX_OK;
    if (!d_packed || !d_out_ids || !d_out_dists) return fail(MX_EINVAL, "null argument");
    DeviceGuard g(device);
    if (!g.ok) return fail(MX_EDEVICE, "hipSetDevice(%d) failed", device);
    const size_t ids_bytes = (size_t)B * k * sizeof(uint64_t), blk = ids_bytes + (size_t)B * k * sizeof(float);
    MX_HIP(launch_merge(hipStreamPerThread, d_packed, blk, static_cast<const char *>(d_packed) + ids_bytes, blk, G, B, k,
                        d_out_ids, d_out_dists, d_out_scores));
    MX_HIP(hipStreamSynchronize(hipStreamPerThread));
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_topk_merge_packed_async(int device, void *hip_stream, const void *d_packed, int G, int B, int k, uint64_t *d_out_ids,
                               float *d_out_dists, float *d_out_scores) try {
    if (G < 1 || B < 0 || k < 0) return fail(MX_EINVAL, "bad merge shape");
    if (B == 0 || k == 0) return MX_OK;
    if (!d_packed || !d_out_ids || !d_out_dists) return fail(MX_EINVAL, "null argument");
    DeviceGuard g(device);
    if (!g.ok) return fail(MX_EDEVICE, "hipSetDevice(%d) failed", device);
    const size_t ids_bytes = (size_t)B * k * sizeof(uint64_t), blk = ids_bytes + (size_t)B * k * sizeof(float);
    MX_HIP(launch_merge(static_cast<hipStream_t>(hip_stream), d_packed, blk, static_cast<const char *>(d_packed) + ids_bytes, blk, G,
                        B, k, d_out_ids, d_out_dists, d_out_scores));
    return MX_OK;
} catch (...) {
    return guard_exception();
}

}  // extern "C"
