/*
 * oracle/hnsw_baseline.cpp -- TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's ACTUAL search algorithm, for bench.py's `cpu_baseline` leg
 * (SURVEY.md section 8(d)(ii)(b)): memex's `hnsw://` store is hnsw_rs 0.1.20 (un-vendored crate,
 * Cargo.lock:1716-1718) constructed as
 *     Hnsw::<f32, DistCosine>::new(max_nb_connection = 16, max_elements = 100, max_layer = 16,
 *                                  ef_construction = 200, DistCosine)   lib/libmemex/src/storage/local.rs:101
 * filled by `hnsw.insert((&vector, id))` (local.rs:65) and queried by
 *     hnsw.search(vec, limit, ef = 32)                                  local.rs:76
 * single-threaded per call.  This file restates the published algorithm (Malkov & Yashunin, "Efficient
 * and robust approximate nearest neighbor search using Hierarchical Navigable Small World graphs",
 * Alg. 1-5) with those parameters:
 *   - level of a new point: floor(-ln(U) / ln(M)), capped at max_layer - 1
 *   - insert: greedy descent (ef = 1) to the point's level, then SEARCH-LAYER with ef_construction per
 *     layer, SELECT-NEIGHBORS-HEURISTIC (Alg. 4, no candidate extension, pruned candidates kept to
 *     fill up), bidirectional links, shrink lists above Mmax (M on upper layers, 2M on layer 0)
 *   - search: greedy descent to layer 1, SEARCH-LAYER on layer 0 with ef = max(ef_arg, k), best k
 *   - distance: DistCosine::eval exactly as oracle/cosine_oracle.c restates it -- BOTH norms are
 *     recomputed for every pair (f32 products, sequential f64 sums), as the crate does; no SIMD
 *     (hnsw_rs is built without its simd features, Cargo.lock:1716-1733).
 * The graph is NOT bit-comparable with hnsw_rs (levels come from an entropy-seeded RNG there); what
 * the baseline reports is the cost and the recall@k of the algorithm under the reference's
 * parameters.  The build is parallel over points (per-node locks) because it is not the thing being
 * timed; searches are timed on ONE thread, like the reference's.
 *
 * C interface (ctypes, oracle/hnsw_baseline.py):
 *   mxh_build(x, n, d, M, efc, max_layer, seed, threads) -> handle
 *   mxh_search(handle, q, nq, k, ef, ids_out[nq*k] (1-based, 0 = none), dists_out[nq*k]) -> seconds (1 thread)
 *   mxh_free(handle)
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <queue>
#include <random>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// hnsw_rs DistCosine::eval (f32 slices): f32 products accumulated in f64, both norms recomputed
inline float dist_cosine(const float *a, const float *b, int d) {
    double dot = 0.0, na = 0.0, nb = 0.0;
    for (int i = 0; i < d; ++i) {
        dot += (double)(a[i] * b[i]);
        na += (double)(a[i] * a[i]);
        nb += (double)(b[i] * b[i]);
    }
    if (na > 0.0 && nb > 0.0) {
        double v = 1.0 - dot / std::sqrt(na * nb);
        return (float)(v < 0.0 ? 0.0 : v);
    }
    return 0.0f;
}

struct Hnsw {
    const float *x = nullptr;
    int64_t n = 0;
    int d = 0, M = 16, efc = 200, max_layer = 16;
    std::vector<int> level;
    std::vector<std::vector<std::vector<int32_t>>> links;  // [node][layer] -> neighbours
    std::vector<std::mutex> locks;
    std::mutex top_mu;
    int64_t entry = -1;
    int top = -1;

    const float *row(int64_t i) const { return x + (size_t)i * d; }
    int mmax(int layer) const { return layer == 0 ? 2 * M : M; }

    typedef std::pair<float, int32_t> DI;

    // Alg. 2: ef closest to q on `layer` starting from `eps`; result as a max-heap on distance
    std::priority_queue<DI> search_layer(const float *q, const std::vector<DI> &eps, int ef, int layer,
                                         std::vector<uint32_t> &visited, uint32_t stamp, bool locked) {
        std::priority_queue<DI> best;                                   // farthest on top
        std::priority_queue<DI, std::vector<DI>, std::greater<DI>> cand;  // nearest on top
        for (const DI &e : eps) {
            visited[e.second] = stamp;
            best.push(e);
            cand.push(e);
        }
        while (!cand.empty()) {
            const DI c = cand.top();
            if (c.first > best.top().first && (int)best.size() >= ef) break;
            cand.pop();
            std::vector<int32_t> nb;
            if (locked) {
                std::lock_guard<std::mutex> lk(locks[c.second]);
                nb = links[c.second][layer];
            } else {
                nb = links[c.second][layer];
            }
            for (int32_t e : nb) {
                if (visited[e] == stamp) continue;
                visited[e] = stamp;
                const float de = dist_cosine(q, row(e), d);
                if ((int)best.size() < ef || de < best.top().first) {
                    cand.push({de, e});
                    best.push({de, e});
                    if ((int)best.size() > ef) best.pop();
                }
            }
        }
        return best;
    }

    // Alg. 4 (no extension, keep pruned): candidates ascending by distance to the base point
    std::vector<int32_t> select(std::vector<DI> c, int m) {
        std::sort(c.begin(), c.end());
        std::vector<int32_t> out;
        std::vector<DI> pruned;
        for (const DI &e : c) {
            if ((int)out.size() >= m) break;
            bool good = true;
            for (int32_t r : out)
                if (dist_cosine(row(e.second), row(r), d) < e.first) {
                    good = false;
                    break;
                }
            if (good) out.push_back(e.second);
            else pruned.push_back(e);
        }
        for (const DI &e : pruned) {
            if ((int)out.size() >= m) break;
            out.push_back(e.second);
        }
        return out;
    }

    void insert(int64_t p, std::vector<uint32_t> &visited, uint32_t &stamp) {
        const int lp = level[p];
        int64_t ep;
        int L;
        {
            std::lock_guard<std::mutex> lk(top_mu);
            ep = entry;
            L = top;
            if (entry < 0) {
                entry = p;
                top = lp;
                return;
            }
        }
        const float *q = row(p);
        std::vector<DI> eps{{dist_cosine(q, row(ep), d), (int32_t)ep}};
        for (int lc = L; lc > lp; --lc) {
            auto w = search_layer(q, eps, 1, lc, visited, ++stamp, true);
            while (w.size() > 1) w.pop();
            eps.assign(1, w.top());
        }
        for (int lc = std::min(L, lp); lc >= 0; --lc) {
            auto w = search_layer(q, eps, efc, lc, visited, ++stamp, true);
            std::vector<DI> c;
            while (!w.empty()) {
                c.push_back(w.top());
                w.pop();
            }
            const std::vector<int32_t> nbrs = select(c, M);
            {
                std::lock_guard<std::mutex> lk(locks[p]);
                links[p][lc] = nbrs;
            }
            for (int32_t e : nbrs) {
                std::lock_guard<std::mutex> lk(locks[e]);
                auto &le = links[e][lc];
                le.push_back((int32_t)p);
                if ((int)le.size() > mmax(lc)) {
                    std::vector<DI> ce;
                    for (int32_t r : le) ce.push_back({dist_cosine(row(e), row(r), d), r});
                    le = select(ce, mmax(lc));
                }
            }
            eps = c;
        }
        if (lp > L) {
            std::lock_guard<std::mutex> lk(top_mu);
            if (lp > top) {
                top = lp;
                entry = p;
            }
        }
    }

    // Alg. 5
    void search(const float *q, int k, int ef_arg, std::vector<uint32_t> &visited, uint32_t &stamp, uint64_t *ids, float *dists) {
        for (int j = 0; j < k; ++j) {
            ids[j] = 0;
            dists[j] = INFINITY;
        }
        if (entry < 0) return;
        std::vector<DI> eps{{dist_cosine(q, row(entry), d), (int32_t)entry}};
        for (int lc = top; lc >= 1; --lc) {
            auto w = search_layer(q, eps, 1, lc, visited, ++stamp, false);
            while (w.size() > 1) w.pop();
            eps.assign(1, w.top());
        }
        auto w = search_layer(q, eps, std::max(ef_arg, k), 0, visited, ++stamp, false);
        while ((int)w.size() > k) w.pop();
        for (int j = (int)w.size() - 1; j >= 0; --j) {
            ids[j] = (uint64_t)w.top().second + 1;  // local.rs:63: ids are 1-based
            dists[j] = w.top().first;
            w.pop();
        }
    }
};

}  // namespace

extern "C" {

void *mxh_build(const float *x, int64_t n, int d, int M, int efc, int max_layer, uint64_t seed, int threads) {
    Hnsw *h = new Hnsw();
    h->x = x;
    h->n = n;
    h->d = d;
    h->M = M;
    h->efc = efc;
    h->max_layer = max_layer;
    h->level.resize(n);
    h->links.resize(n);
    std::vector<std::mutex> lk(n);
    h->locks.swap(lk);
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const double scale = 1.0 / std::log((double)M);
    for (int64_t i = 0; i < n; ++i) {
        double u = U(rng);
        if (u <= 0.0) u = 1e-300;
        int l = (int)std::floor(-std::log(u) * scale);
        h->level[i] = std::min(l, max_layer - 1);
        h->links[i].resize(h->level[i] + 1);
    }
    // the first points go in serially so that the upper layers exist before the parallel phase
    {
        std::vector<uint32_t> visited(n, 0);
        uint32_t stamp = 0;
        const int64_t head = std::min<int64_t>(n, 1024);
        for (int64_t i = 0; i < head; ++i) h->insert(i, visited, stamp);
    }
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
#endif
    {
        std::vector<uint32_t> visited(n, 0);
        uint32_t stamp = 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
        for (int64_t i = 1024; i < n; ++i) h->insert(i, visited, stamp);
    }
    return h;
}

double mxh_search(void *handle, const float *q, int nq, int k, int ef, uint64_t *ids, float *dists) {
    Hnsw *h = static_cast<Hnsw *>(handle);
    std::vector<uint32_t> visited(h->n, 0);
    uint32_t stamp = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < nq; ++i) h->search(q + (size_t)i * h->d, k, ef, visited, stamp, ids + (size_t)i * k, dists + (size_t)i * k);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void mxh_free(void *handle) { delete static_cast<Hnsw *>(handle); }

}  // extern "C"
