// index.hip -- host logic of the GPU-resident flat cosine index behind the C ABI
// (include/memex_hip.h).  Drop-in for memex's HnswStore (reference
// lib/libmemex/src/storage/local.rs:21-166) reached through the VectorStore trait
// (lib/libmemex/src/storage/mod.rs:55-66).
//
// Search pipeline per batch of <= 256 queries (kernels: scan.hip, index_kernels.hip):
//   prep      normalise queries -> bf16 MFMA fragments; f64 query norms (DistCosine order)
//   stage 0   scan the first 32*nwg rows with theta = -inf: every score lands in a lane buffer
//   update    gather -> k-th best approximate cosine -> theta = kth - margin, prune pool
//   stage i   scan geometrically growing row ranges, appending only rows with score >= theta
//   final     exact f64 DistCosine on the surviving pool, order by (dist_f32, id), emit
// The pool provably contains the exact top-k: |approx - exact| <= kApproxErr for every row, and a
// row is only ever discarded when its approximate score is more than 2*kApproxErr below the k-th
// best approximate score seen so far.  Buffer overflows (pathological duplicates / orderings) are
// detected per query and re-answered on the EXACT path (f64 on every row).
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "index_kernels.h"
#include "mx_common.h"

namespace mx {

std::string &last_error_slot() {
    static thread_local std::string slot;
    return slot;
}

namespace {

struct Scratch {
    void *qfrag = nullptr;
    float *qpad = nullptr;
    double *qnorm2 = nullptr;
    float *theta = nullptr;
    uint32_t *overflow = nullptr;
    uint32_t *pool_cnt = nullptr;
    Cand *pool[2] = {nullptr, nullptr};
    Cand *lane_buf = nullptr;
    uint32_t *lane_cnt = nullptr;
    float *qstage = nullptr;       // [256, dim] host->device query staging
    uint64_t *out_ids = nullptr;   // [256, kcap] device outputs for the host API
    float *out_scores = nullptr;
    float *out_dists = nullptr;
    int32_t *out_nfound = nullptr;
    int kcap = 0;
    uint64_t *exact_keys = nullptr;
    uint64_t exact_cap = 0;
    uint64_t *sel_state = nullptr;  // [4 + ksel]
    int ksel = 0;
    float *max_err = nullptr;
    // pinned staging of the host API: queries in, results out (one H2D / D2H per combined batch)
    float *h_q = nullptr;
    uint64_t *h_ids = nullptr;
    float *h_scores = nullptr, *h_dists = nullptr;
    int32_t *h_nf = nullptr;
    uint32_t *host_flags = nullptr;  // pinned: [overflow 256 | pool_cnt 256], one async D2H per batch
    uint32_t *dev_flags = nullptr;   // device: same layout (overflow and pool_cnt live back to back)
    bool ready = false;
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
    uint64_t n = 0, cap = 0;
    uint64_t id_offset = 0;
    uint32_t *flags = nullptr;  // device: [0] non-finite rows, [1] out-of-range-norm rows (last add)
    uint64_t wild_rows = 0;
    int mode = MX_SEARCH_AUTO;
    bool profiling = false;
    int n_cu = 0, nwg = 0;
    Scratch s;
    mx_index_stats stats{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

std::mutex g_reg_mu;
std::map<std::string, mx_index *> g_registry;
std::once_flag g_scan_once;
hipError_t g_scan_setup_err = hipSuccess;

int free_index(mx_index *idx) {
    DeviceGuard g(idx->device);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    auto F = [](void *p) {
        if (p) (void)hipFree(p);
    };
    F(idx->x); F(idx->scale); F(idx->xh); F(idx->flags);
    Scratch &s = idx->s;
    F(s.qfrag); F(s.qpad); F(s.qnorm2); F(s.theta); F(s.dev_flags); F(s.pool[0]); F(s.pool[1]);
    if (s.host_flags) (void)hipHostFree(s.host_flags);
    for (void *hp : {(void *)s.h_q, (void *)s.h_ids, (void *)s.h_scores, (void *)s.h_dists, (void *)s.h_nf})
        if (hp) (void)hipHostFree(hp);
    F(s.lane_buf); F(s.lane_cnt); F(s.qstage); F(s.out_ids); F(s.out_scores); F(s.out_dists); F(s.out_nfound);
    F(s.exact_keys); F(s.sel_state); F(s.max_err);
    if (idx->ev0) (void)hipEventDestroy(idx->ev0);
    if (idx->ev1) (void)hipEventDestroy(idx->ev1);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    delete idx;
    return MX_OK;
}

int ensure_scratch(mx_index *idx) {
    Scratch &s = idx->s;
    if (s.ready) return MX_OK;
    const size_t ds = (size_t)idx->ds;
    MX_HIP(hipMalloc(&s.qfrag, (size_t)kMaxBatch * ds * 2));
    MX_HIP(hipMalloc(&s.qpad, (size_t)kMaxBatch * ds * 4));
    MX_HIP(hipMalloc(&s.qnorm2, kMaxBatch * sizeof(double)));
    MX_HIP(hipMalloc(&s.theta, kMaxBatch * sizeof(float)));
    MX_HIP(hipMalloc(&s.dev_flags, 2 * kMaxBatch * sizeof(uint32_t)));
    s.overflow = s.dev_flags;
    s.pool_cnt = s.dev_flags + kMaxBatch;
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.host_flags), 2 * kMaxBatch * sizeof(uint32_t), hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) MX_HIP(hipMalloc(&s.pool[i], (size_t)kMaxBatch * kPoolCap * sizeof(Cand)));
    MX_HIP(hipMalloc(&s.lane_buf, (size_t)idx->nwg * kScanThreads * kLaneCap * sizeof(Cand)));
    MX_HIP(hipMalloc(&s.lane_cnt, (size_t)idx->nwg * kScanThreads * sizeof(uint32_t)));
    MX_HIP(hipMalloc(&s.qstage, (size_t)kMaxBatch * idx->dim * sizeof(float)));
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_q), (size_t)kMaxBatch * idx->dim * sizeof(float), hipHostMallocDefault));
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_nf), kMaxBatch * sizeof(int32_t), hipHostMallocDefault));
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
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_ids), (size_t)kMaxBatch * kc * sizeof(uint64_t), hipHostMallocDefault));
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_scores), (size_t)kMaxBatch * kc * sizeof(float), hipHostMallocDefault));
    MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&s.h_dists), (size_t)kMaxBatch * kc * sizeof(float), hipHostMallocDefault));
    s.kcap = kc;
    return MX_OK;
}

int ensure_exact(mx_index *idx, int k) {
    Scratch &s = idx->s;
    if (s.exact_cap < idx->n) {
        if (s.exact_keys) (void)hipFree(s.exact_keys);
        s.exact_keys = nullptr;
        s.exact_cap = 0;
        const uint64_t cap = std::max<uint64_t>(idx->n, 1024);
        MX_HIP(hipMalloc(&s.exact_keys, cap * sizeof(uint64_t)));
        s.exact_cap = cap;
    }
    if (s.ksel < k) {
        if (s.sel_state) (void)hipFree(s.sel_state);
        s.sel_state = nullptr;
        s.ksel = 0;
        const int kk = std::max(k, 64);
        MX_HIP(hipMalloc(&s.sel_state, (size_t)(4 + kk) * sizeof(uint64_t)));
        s.ksel = kk;
    }
    return MX_OK;
}

int ensure_capacity(mx_index *idx, uint64_t rows) {
    if (rows <= idx->cap) return MX_OK;
    uint64_t want = std::max<uint64_t>(rows, idx->cap + idx->cap / 2);
    want = round_up(std::max<uint64_t>(want, 1024), kTileRows);
    float *nx = nullptr, *nsc = nullptr;
    const size_t rowb = (size_t)idx->ds * sizeof(float);
    MX_HIP(hipMalloc(&nx, want * rowb));
    hipError_t e = hipMalloc(&nsc, want * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(nx);
        return fail(MX_ENOMEM, "hipMalloc(scale): %s", hipGetErrorString(e));
    }
    if (idx->n) {
        MX_HIP(hipMemcpyAsync(nx, idx->x, idx->n * rowb, hipMemcpyDeviceToDevice, idx->stream));
        MX_HIP(hipMemcpyAsync(nsc, idx->scale, idx->n * sizeof(float), hipMemcpyDeviceToDevice, idx->stream));
    }
    MX_HIP(hipMemsetAsync(nx + idx->n * (size_t)idx->ds, 0, (want - idx->n) * rowb, idx->stream));
    MX_HIP(hipMemsetAsync(nsc + idx->n, 0, (want - idx->n) * sizeof(float), idx->stream));
    void *nh = nullptr;
    if (idx->want_filter && idx->kc <= kMaxKC) {
        // the filter copy is an accelerator, not a requirement: without HBM for it the index
        // keeps working on the f32 scan
        const size_t hb = (size_t)want * idx->ds * 2;
        if (hipMalloc(&nh, hb) != hipSuccess) {
            (void)hipGetLastError();
            nh = nullptr;
        } else {
            const size_t used = idx->xh ? (size_t)round_up(idx->n, kTileRows) * idx->ds * 2 : 0;
            if (used) MX_HIP(hipMemcpyAsync(nh, idx->xh, used, hipMemcpyDeviceToDevice, idx->stream));
            MX_HIP(hipMemsetAsync(static_cast<char *>(nh) + used, 0, hb - used, idx->stream));
            if (!idx->xh && idx->n)  // (re)enabled on a populated index
                MX_HIP(launch_shadow(idx->stream, nx, nsc, idx->ds, 0, (uint32_t)((idx->n + kTileRows - 1) / kTileRows), nh));
        }
    }
    MX_HIP(hipStreamSynchronize(idx->stream));
    if (idx->x) (void)hipFree(idx->x);
    if (idx->scale) (void)hipFree(idx->scale);
    if (idx->xh) (void)hipFree(idx->xh);
    idx->x = nx;
    idx->scale = nsc;
    idx->xh = nh;
    idx->cap = want;
    return MX_OK;
}

// rows already on the device ([n, dim]); appends and validates
int add_device_locked(mx_index *idx, const float *d_rows, uint64_t n, uint64_t *first_id) {
    if (n == 0) {
        if (first_id) *first_id = idx->id_offset + idx->n + 1;
        return MX_OK;
    }
    if (idx->n + n > 0xfffffff0ull) return fail(MX_EINSERT, "index shard limited to 2^32 rows");
    int rc = ensure_capacity(idx, idx->n + n);
    if (rc != MX_OK) return rc;
    MX_HIP(hipMemsetAsync(idx->flags, 0, 2 * sizeof(uint32_t), idx->stream));
    MX_HIP(launch_ingest(idx->stream, d_rows, n, idx->dim, idx->x, idx->scale, idx->n, idx->ds, idx->flags));
    if (idx->xh)  // tiles touched by this append (the first one may already be partly filled)
        MX_HIP(launch_shadow(idx->stream, idx->x, idx->scale, idx->ds, (uint32_t)(idx->n / kTileRows),
                             (uint32_t)((idx->n + n + kTileRows - 1) / kTileRows), idx->xh));
    uint32_t fl[2] = {0, 0};
    MX_HIP(hipMemcpyAsync(fl, idx->flags, sizeof(fl), hipMemcpyDeviceToHost, idx->stream));
    MX_HIP(hipStreamSynchronize(idx->stream));
    if (fl[0] != 0) return fail(MX_EINVAL, "%u row(s) contain non-finite values; nothing inserted", fl[0]);
    idx->wild_rows += fl[1];
    if (first_id) *first_id = idx->id_offset + idx->n + 1;  // local.rs:63: next_id = len + 1
    idx->n += n;
    return MX_OK;
}

struct Stage {
    uint32_t t0, t1;
    bool main;
};

std::vector<Stage> plan_stages(uint64_t n, int nwg, int k) {
    std::vector<Stage> st;
    const uint64_t tiles = (n + kTileRows - 1) / kTileRows;
    if (tiles == 0) return st;
    const uint64_t head = std::min<uint64_t>(tiles, (uint64_t)nwg);
    st.push_back({0u, (uint32_t)head, false});
    if (tiles > head) {
        // rows passing a stage ~ k * (growth-1) * tail(margin); keep that a few per lane buffer
        const double R = (double)tiles / (double)head;
        const double G = std::min(48.0, std::max(2.0, 1.0 + 400.0 / (double)std::max(k, 1)));
        const int s = std::max(1, (int)std::ceil(std::log(R) / std::log(G) - 1e-9));
        uint64_t prev = head;
        for (int i = 1; i <= s; ++i) {
            uint64_t b = i == s ? tiles : (uint64_t)std::llround((double)head * std::pow(R, (double)i / s));
            b = std::min<uint64_t>(std::max<uint64_t>(b, prev + 1), tiles);
            st.push_back({(uint32_t)prev, (uint32_t)b, false});
            prev = b;
            if (b == tiles) break;
        }
    }
    Stage &last = st.back();
    if ((uint64_t)(last.t1 - last.t0) * 2 >= tiles) last.main = true;
    return st;
}

// one batch (B <= 256) with queries and outputs on the device
int search_batch(mx_index *idx, const float *d_q, int B, int k, uint64_t *d_ids, float *d_scores, float *d_dists,
                 int32_t *d_nfound) {
    int rc = ensure_scratch(idx);
    if (rc != MX_OK) return rc;
    Scratch &s = idx->s;
    hipStream_t st = idx->stream;
    MX_HIP(launch_prep_queries(st, d_q, B, idx->dim, idx->ds, s.qfrag, s.qpad, s.qnorm2, s.theta, s.overflow,
                               s.pool_cnt));
    const bool fast = idx->mode == MX_SEARCH_AUTO && idx->kc <= kMaxKC && idx->wild_rows == 0 && k <= 256 && k > 0;
    std::vector<int> redo;
    bool timed = false;
    if (fast || idx->n == 0 || k == 0) {
        int cur = 0;
        if (fast) {
            for (const Stage &sg : plan_stages(idx->n, idx->nwg, k)) {
                ScanParams p;
                p.x = idx->x;
                p.xh = idx->xh;
                p.scale = idx->scale;
                p.qfrag = s.qfrag;
                p.theta = s.theta;
                p.n_rows = idx->n;
                p.tile_begin = sg.t0;
                p.tile_end = sg.t1;
                p.ds = (uint32_t)idx->ds;
                p.lane_buf = s.lane_buf;
                p.lane_cnt = s.lane_cnt;
                p.overflow = s.overflow;
                if (sg.main && idx->profiling) MX_HIP(hipEventRecord(idx->ev0, st));
                if (idx->xh)
                    MX_HIP(launch_scan16(st, idx->kc, sg.main, idx->nwg, p));
                else
                    MX_HIP(launch_scan(st, idx->kc, sg.main, idx->nwg, p));
                if (sg.main) {
                    if (idx->profiling) {
                        MX_HIP(hipEventRecord(idx->ev1, st));
                        timed = true;
                    }
                    idx->stats.scan_launches += 1;
                    idx->stats.scan_bytes += (uint64_t)(sg.t1 - sg.t0) * kTileRows * idx->ds * (idx->xh ? 2ull : 4ull);
                }
                MX_HIP(launch_update(st, B, k, idx->nwg, s.lane_buf, s.lane_cnt, s.pool[cur], s.pool[cur ^ 1],
                                     s.pool_cnt, s.theta, s.overflow));
                cur ^= 1;
            }
        }
        MX_HIP(launch_final(st, B, k, idx->dim, idx->ds, idx->x, idx->n, idx->id_offset, s.qpad, s.qnorm2,
                            s.pool[cur], s.pool_cnt, s.overflow, d_ids, d_scores, d_dists, d_nfound,
                            idx->profiling ? s.max_err : nullptr));
        if (fast) {
            const uint32_t *ovf = s.host_flags, *cnt = s.host_flags + kMaxBatch;
            MX_HIP(hipMemcpyAsync(s.host_flags, s.dev_flags, 2 * kMaxBatch * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            MX_HIP(hipStreamSynchronize(st));
            for (int b = 0; b < B; ++b) {
                if (ovf[b]) redo.push_back(b);
                else idx->stats.candidates += cnt[b];
            }
        }
    } else {
        for (int b = 0; b < B; ++b) redo.push_back(b);
    }
    if (!redo.empty()) {
        rc = ensure_exact(idx, k);
        if (rc != MX_OK) return rc;
        for (int b : redo) {
            MX_HIP(launch_exact_query(st, k, idx->dim, idx->ds, idx->x, idx->n, idx->id_offset,
                                      s.qpad + (size_t)b * idx->ds, s.exact_keys, s.sel_state,
                                      d_ids + (size_t)b * k, d_scores + (size_t)b * k,
                                      d_dists ? d_dists + (size_t)b * k : nullptr, d_nfound + b));
        }
        if (idx->mode == MX_SEARCH_AUTO) idx->stats.fallback_queries += redo.size();
    }
    MX_HIP(hipStreamSynchronize(st));
    if (timed) {
        float ms = 0.f;
        MX_HIP(hipEventElapsedTime(&ms, idx->ev0, idx->ev1));
        idx->stats.scan_ms += ms;
    }
    idx->stats.searches += 1;
    idx->stats.queries += (uint64_t)B;
    return MX_OK;
}

const char kMagic[8] = {'M', 'X', 'F', 'L', 'A', 'T', '0', '1'};
std::string store_file(const char *dir) { return std::string(dir) + "/vectors.mxflat"; }

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char *mx_last_error(void) { return last_error_slot().c_str(); }
const char *mx_version(void) { return "memex-hip 0.1.0 (gfx950)"; }

int mx_device_count(int *n) {
    if (!n) return fail(MX_EINVAL, "null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return fail(MX_EDEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n = c;
    return MX_OK;
}

int mx_index_open(const char *key, int dim, int device, mx_index **out) {
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
            if (idx->device != device)
                return fail(MX_EINVAL, "index '%s' lives on device %d, not %d", k.c_str(), idx->device, device);
            idx->refs += 1;
            *out = idx;
            return MX_OK;
        }
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(MX_EDEVICE, "no HIP device available (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(MX_EDEVICE, "device %d out of range (have %d)", device, ndev);
    DeviceGuard g(device);
    if (!g.ok) return fail(MX_EDEVICE, "hipSetDevice(%d) failed", device);
    std::call_once(g_scan_once, [] {
        g_scan_setup_err = scan_setup();
        if (g_scan_setup_err == hipSuccess) g_scan_setup_err = scan16_setup();
    });
    if (g_scan_setup_err != hipSuccess)
        return fail(MX_EDEVICE, "scan kernel setup failed: %s (is this a gfx950 device?)", hipGetErrorString(g_scan_setup_err));
    std::unique_ptr<mx_index> idx(new mx_index());
    idx->key = k;
    idx->dim = dim;
    idx->ds = (int)round_up((uint64_t)dim, kChunkFloats);
    idx->kc = idx->ds / kChunkFloats;
    idx->device = device;
    hipDeviceProp_t prop;
    MX_HIP(hipGetDeviceProperties(&prop, device));
    idx->n_cu = prop.multiProcessorCount;
    idx->nwg = std::max(1, std::min(idx->n_cu, kMaxScanWGs));
    MX_HIP(hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking));
    MX_HIP(hipEventCreate(&idx->ev0));
    MX_HIP(hipEventCreate(&idx->ev1));
    MX_HIP(hipMalloc(&idx->flags, 2 * sizeof(uint32_t)));
    mx_index *raw = idx.release();
    if (!k.empty()) g_registry[k] = raw;
    *out = raw;
    return MX_OK;
}

void mx_index_close(mx_index *idx) {
    if (!idx) return;
    std::lock_guard<std::mutex> lk(g_reg_mu);
    if (--idx->refs > 0) return;
    if (!idx->key.empty()) g_registry.erase(idx->key);
    free_index(idx);
}

int mx_index_dim(mx_index *idx, int *dim) {
    if (!idx || !dim) return fail(MX_EINVAL, "null argument");
    *dim = idx->dim;
    return MX_OK;
}

int mx_index_size(mx_index *idx, uint64_t *n) {
    if (!idx || !n) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    *n = idx->n;
    return MX_OK;
}

int mx_index_reserve(mx_index *idx, uint64_t rows) {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->device);
    return ensure_capacity(idx, rows);
}

int mx_index_set_id_offset(mx_index *idx, uint64_t off) {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->id_offset = off;
    return MX_OK;
}

int mx_index_add_device(mx_index *idx, const float *d_rows, uint64_t n, uint64_t *first_id) {
    if (!idx || (!d_rows && n)) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->device);
    return add_device_locked(idx, d_rows, n, first_id);
}

int mx_index_add(mx_index *idx, const float *rows, uint64_t n, uint64_t *first_id) {
    if (!idx || (!rows && n)) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
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
    if (rc == MX_OK && first_id) *first_id = first;
    return rc;
}

int mx_index_clear(mx_index *idx) {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->n = 0;  // ids restart at 1 (local.rs:50,63); HBM is kept for reuse
    idx->wild_rows = 0;
    return MX_OK;
}

int mx_index_set_search_mode(mx_index *idx, int mode) {
    if (!idx) return fail(MX_EINVAL, "null index");
    if (mode != MX_SEARCH_AUTO && mode != MX_SEARCH_EXACT) return fail(MX_EINVAL, "unknown search mode %d", mode);
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->mode = mode;
    return MX_OK;
}

int mx_index_search_device(mx_index *idx, const float *d_q, int B, int k, uint64_t *d_ids, float *d_scores,
                           float *d_dists, int32_t *d_nfound) {
    if (!idx) return fail(MX_ESEARCH, "null index");
    if (B < 0 || k < 0) return fail(MX_EINVAL, "negative batch or k");
    if (B == 0) return MX_OK;
    if (!d_q || !d_nfound || (k > 0 && (!d_ids || !d_scores))) return fail(MX_EINVAL, "null argument");
    if (k > 4096) return fail(MX_EUNSUPPORTED, "k = %d > 4096", k);
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->device);
    for (int b0 = 0; b0 < B; b0 += kMaxBatch) {
        const int nb = std::min(kMaxBatch, B - b0);
        int rc = search_batch(idx, d_q + (size_t)b0 * idx->dim, nb, k, d_ids + (size_t)b0 * k,
                              d_scores + (size_t)b0 * k, d_dists ? d_dists + (size_t)b0 * k : nullptr, d_nfound + b0);
        if (rc != MX_OK) return rc;
    }
    return MX_OK;
}

namespace {

// one GPU batch (sum of B <= 256, same k) for a group of host requests: queries are packed into
// pinned memory, one H2D, search_batch, one D2H per output array, results scattered to the callers
int run_combined(mx_index *idx, const std::vector<SearchReq *> &batch) {
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->device);
    const int k = batch[0]->k;
    int rc = ensure_scratch(idx);
    if (rc != MX_OK) return rc;
    rc = ensure_out(idx, k);
    if (rc != MX_OK) return rc;
    Scratch &s = idx->s;
    const size_t dim = (size_t)idx->dim;
    int nb = 0;
    for (const SearchReq *r : batch) {
        memcpy(s.h_q + (size_t)nb * dim, r->q, (size_t)r->B * dim * sizeof(float));
        nb += r->B;
    }
    MX_HIP(hipMemcpyAsync(s.qstage, s.h_q, (size_t)nb * dim * sizeof(float), hipMemcpyHostToDevice, idx->stream));
    rc = search_batch(idx, s.qstage, nb, k, s.out_ids, s.out_scores, s.out_dists, s.out_nfound);
    if (rc != MX_OK) return rc;
    if (k > 0) {
        MX_HIP(hipMemcpyAsync(s.h_ids, s.out_ids, (size_t)nb * k * sizeof(uint64_t), hipMemcpyDeviceToHost, idx->stream));
        MX_HIP(hipMemcpyAsync(s.h_scores, s.out_scores, (size_t)nb * k * sizeof(float), hipMemcpyDeviceToHost, idx->stream));
        MX_HIP(hipMemcpyAsync(s.h_dists, s.out_dists, (size_t)nb * k * sizeof(float), hipMemcpyDeviceToHost, idx->stream));
    }
    MX_HIP(hipMemcpyAsync(s.h_nf, s.out_nfound, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, idx->stream));
    MX_HIP(hipStreamSynchronize(idx->stream));
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
                    int32_t *n_found) {
    if (!idx) return fail(MX_ESEARCH, "null index");
    if (B < 0 || k < 0) return fail(MX_EINVAL, "negative batch or k");
    if (B == 0) return MX_OK;
    if (!q || !n_found || (k > 0 && (!ids || !scores))) return fail(MX_EINVAL, "null argument");
    if (k > 4096) return fail(MX_EUNSUPPORTED, "k = %d > 4096", k);
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
            const int rc = run_combined(idx, batch);
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
}

int mx_index_set_filter_copy(mx_index *idx, int on) {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->device);
    idx->want_filter = on != 0;
    if (!on) {
        if (idx->xh) {
            MX_HIP(hipStreamSynchronize(idx->stream));
            (void)hipFree(idx->xh);
            idx->xh = nullptr;
        }
        return MX_OK;
    }
    if (idx->xh || idx->cap == 0 || idx->kc > kMaxKC) return MX_OK;  // present, or built with the first rows
    void *nh = nullptr;
    const size_t hb = (size_t)idx->cap * idx->ds * 2;
    hipError_t e = hipMalloc(&nh, hb);
    if (e != hipSuccess) return fail(MX_ENOMEM, "hipMalloc(filter copy, %zu bytes): %s", hb, hipGetErrorString(e));
    MX_HIP(hipMemsetAsync(nh, 0, hb, idx->stream));
    MX_HIP(launch_shadow(idx->stream, idx->x, idx->scale, idx->ds, 0, (uint32_t)((idx->n + kTileRows - 1) / kTileRows), nh));
    MX_HIP(hipStreamSynchronize(idx->stream));
    idx->xh = nh;
    return MX_OK;
}

int mx_index_set_profiling(mx_index *idx, int on) {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->profiling = on != 0;
    return MX_OK;
}

int mx_index_get_stats(mx_index *idx, mx_index_stats *out) {
    if (!idx || !out) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (idx->s.max_err) {  // device-side running maximum (profiling mode); fetched on demand
        DeviceGuard g(idx->device);
        float e = 0.f;
        MX_HIP(hipMemcpy(&e, idx->s.max_err, sizeof(float), hipMemcpyDeviceToHost));
        idx->stats.max_abs_err = std::max(idx->stats.max_abs_err, (double)e);
    }
    idx->stats.filter_copy_bytes = idx->xh ? (uint64_t)idx->cap * idx->ds * 2ull : 0;
    *out = idx->stats;
    return MX_OK;
}

int mx_index_reset_stats(mx_index *idx) {
    if (!idx) return fail(MX_EINVAL, "null index");
    std::lock_guard<std::mutex> lk(idx->mu);
    idx->stats = mx_index_stats{};
    if (idx->s.max_err) {
        DeviceGuard g(idx->device);
        (void)hipMemset(idx->s.max_err, 0, sizeof(float));
    }
    return MX_OK;
}

// ---- persistence (replaces hnsw file_dump / load_hnsw, local.rs:115-165) ----------------------
int mx_index_save(mx_index *idx, const char *dir) {
    if (!idx || !dir) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->device);
    struct stat sb;
    if (stat(dir, &sb) != 0 && mkdir(dir, 0755) != 0) return fail(MX_EIO, "cannot create directory %s", dir);
    const std::string tmp = store_file(dir) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(MX_EIO, "cannot open %s for writing", tmp.c_str());
    uint32_t hdr[2] = {(uint32_t)idx->dim, 0};
    uint64_t n = idx->n;
    bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(hdr, sizeof(hdr), 1, f) == 1 && fwrite(&n, sizeof(n), 1, f) == 1;
    const uint64_t chunk = std::max<uint64_t>(1, (32ull << 20) / ((uint64_t)idx->ds * 4));
    std::vector<float> host((size_t)std::min<uint64_t>(chunk, std::max<uint64_t>(n, 1)) * idx->ds);
    for (uint64_t r = 0; ok && r < n; r += chunk) {
        const uint64_t m = std::min(chunk, n - r);
        hipError_t e = hipMemcpy(host.data(), idx->x + (size_t)r * idx->ds, (size_t)m * idx->ds * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            fclose(f);
            unlink(tmp.c_str());
            return fail(MX_EDEVICE, "hipMemcpy D2H: %s", hipGetErrorString(e));
        }
        for (uint64_t i = 0; ok && i < m; ++i)
            ok = fwrite(host.data() + (size_t)i * idx->ds, sizeof(float), (size_t)idx->dim, f) == (size_t)idx->dim;
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), store_file(dir).c_str()) != 0) {
        unlink(tmp.c_str());
        return fail(MX_EIO, "write to %s failed", store_file(dir).c_str());
    }
    return MX_OK;
}

int mx_index_load(mx_index *idx, const char *dir) {
    if (!idx || !dir) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    DeviceGuard g(idx->device);
    FILE *f = fopen(store_file(dir).c_str(), "rb");
    if (!f) return fail(MX_EIO, "cannot open %s", store_file(dir).c_str());
    char magic[8];
    uint32_t hdr[2];
    uint64_t n = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kMagic, 8) != 0 || fread(hdr, sizeof(hdr), 1, f) != 1 ||
        fread(&n, sizeof(n), 1, f) != 1) {
        fclose(f);
        return fail(MX_EIO, "%s: bad header", store_file(dir).c_str());
    }
    if ((int)hdr[0] != idx->dim) {
        fclose(f);
        return fail(MX_EIO, "%s holds dim %u, index has dim %d", store_file(dir).c_str(), hdr[0], idx->dim);
    }
    idx->n = 0;
    idx->wild_rows = 0;
    const uint64_t chunk = std::max<uint64_t>(1, (32ull << 20) / ((uint64_t)idx->dim * 4));
    std::vector<float> host((size_t)std::min<uint64_t>(chunk, std::max<uint64_t>(n, 1)) * idx->dim);
    float *stage = nullptr;
    hipError_t e = hipMalloc(&stage, host.size() * sizeof(float));
    if (e != hipSuccess) {
        fclose(f);
        return fail(MX_ENOMEM, "hipMalloc: %s", hipGetErrorString(e));
    }
    int rc = MX_OK;
    for (uint64_t r = 0; r < n && rc == MX_OK; r += chunk) {
        const uint64_t m = std::min(chunk, n - r);
        if (fread(host.data(), sizeof(float), (size_t)m * idx->dim, f) != (size_t)m * idx->dim) {
            rc = fail(MX_EIO, "%s: truncated", store_file(dir).c_str());
            break;
        }
        e = hipMemcpy(stage, host.data(), (size_t)m * idx->dim * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            rc = fail(MX_EDEVICE, "hipMemcpy H2D: %s", hipGetErrorString(e));
            break;
        }
        rc = add_device_locked(idx, stage, m, nullptr);
    }
    (void)hipFree(stage);
    fclose(f);
    if (rc != MX_OK) idx->n = 0;
    return rc;
}

int mx_index_has_store(const char *dir, int *exists) {
    if (!dir || !exists) return fail(MX_EINVAL, "null argument");
    struct stat sb;
    *exists = stat(store_file(dir).c_str(), &sb) == 0 ? 1 : 0;
    return MX_OK;
}

int mx_index_store_info(const char *dir, int *dim, uint64_t *n_rows) {
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
}

int mx_index_remove_files(const char *dir) {
    if (!dir) return fail(MX_EINVAL, "null argument");
    const std::string p = store_file(dir);
    struct stat sb;
    if (stat(p.c_str(), &sb) == 0 && unlink(p.c_str()) != 0) return fail(MX_EIO, "cannot remove %s", p.c_str());
    return MX_OK;
}

int mx_topk_merge_device(int device, const uint64_t *d_ids, const float *d_dists, int G, int B, int k,
                         uint64_t *d_out_ids, float *d_out_dists, float *d_out_scores) {
    if (G < 1 || B < 0 || k < 0) return fail(MX_EINVAL, "bad merge shape");
    if (B == 0 || k == 0) return MX_OK;
    if (!d_ids || !d_dists || !d_out_ids || !d_out_dists) return fail(MX_EINVAL, "null argument");
    DeviceGuard g(device);
    if (!g.ok) return fail(MX_EDEVICE, "hipSetDevice(%d) failed", device);
    MX_HIP(launch_merge(hipStreamPerThread, d_ids, (size_t)B * k * sizeof(uint64_t), d_dists, (size_t)B * k * sizeof(float),
                        G, B, k, d_out_ids, d_out_dists, d_out_scores));
    MX_HIP(hipStreamSynchronize(hipStreamPerThread));
    return MX_OK;
}

int mx_topk_merge_packed_device(int device, const void *d_packed, int G, int B, int k, uint64_t *d_out_ids,
                                float *d_out_dists, float *d_out_scores) {
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
}

}  // extern "C"
