/*
 * oracle/cosine_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the arithmetic on memex's vector-search path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (memex_amd/, libmemex_hip.so) never links or calls it.
 *
 * What it restates (reference = /root/reference, Rust; cannot be built here):
 *   - HnswStore::insert   lib/libmemex/src/storage/local.rs:62-69   ids are dense, 1-based
 *   - HnswStore::search   lib/libmemex/src/storage/local.rs:71-91   neighbours ascending by
 *     distance, similarity = 1.0 - (1.0 / (1.0 / distance)) in f32   (local.rs:86)
 *   - the distance itself lives in the un-vendored crate hnsw_rs 0.1.20 @ git 52a7f917
 *     (lib/libmemex/Cargo.toml:14, Cargo.lock:1716-1718): DistCosine::eval for f32 is
 *         (dot, na, nb) = fold over i of ((a_i*b_i) as f64, (a_i*a_i) as f64, (b_i*b_i) as f64)
 *         if na > 0 && nb > 0 { max(1 - dot/sqrt(na*nb), 0) as f32 } else { 0 }
 *     i.e. f32 products, sequential f64 accumulation, one f64 sqrt and divide, clamp, round to f32.
 *
 * HNSW itself is approximate and seeded from entropy, so it cannot be a bit-exact oracle; the
 * contract is exact brute force under the same distance, ordered by
 * (dist_f32 ascending, internal id ascending).  HNSW agrees with it wherever HNSW's recall is 1
 * (always for the reference's own 3-vector known-answer test, local.rs:201-214).
 *
 * Pinning: the only known-answer test the reference holds on this path is test_hnsw
 * (local.rs:175-214): tests/test_oracle.py checks it plus the derived dist/score table of
 * SURVEY.md section 4.  Scores are asserted by no reference test.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* hnsw_rs DistCosine::eval (f32).  volatile-free: -ffp-contract=off keeps a*b a rounded f32. */
float mxo_dist_cosine(const float *a, const float *b, int d) {
    double dot = 0.0, na = 0.0, nb = 0.0;
    for (int i = 0; i < d; ++i) {
        float pab = a[i] * b[i];
        float paa = a[i] * a[i];
        float pbb = b[i] * b[i];
        dot += (double)pab;
        na += (double)paa;
        nb += (double)pbb;
    }
    if (na > 0.0 && nb > 0.0) {
        double dist = 1.0 - dot / sqrt(na * nb);
        if (dist < 0.0) dist = 0.0;
        return (float)dist;
    }
    return 0.0f;
}

/* local.rs:86 -- `1.0 - (1.0 / (1.0 / x.distance))`, every operation rounded to f32. */
float mxo_score(float dist) {
    volatile float t = 1.0f / dist; /* dist == 0 -> +inf */
    volatile float u = 1.0f / t;    /* +inf -> 0 */
    return 1.0f - u;
}

typedef struct {
    float dist;
    uint64_t id;
} mxo_hit;

static int hit_less(const mxo_hit *x, const mxo_hit *y) {
    if (x->dist != y->dist) return x->dist < y->dist;
    return x->id < y->id;
}

/* keep the k smallest (dist, id) in a sorted array */
static void topk_push(mxo_hit *heap, int *n, int k, mxo_hit h) {
    if (*n == k && !hit_less(&h, &heap[k - 1])) return;
    int pos = (*n < k) ? (*n)++ : k - 1;
    while (pos > 0 && hit_less(&h, &heap[pos - 1])) {
        heap[pos] = heap[pos - 1];
        --pos;
    }
    heap[pos] = h;
}

/*
 * Exact brute-force search.  corpus: [n, d] row-major f32, ids = row + 1 + id_offset
 * (local.rs:63: next_id = len + 1).  queries: [B, d].  Outputs row-major [B, k]:
 * ids (0 where not found), dists, scores; n_found[b] = min(k, n).
 */
int mxo_search(const float *corpus, uint64_t n, int d, uint64_t id_offset, const float *queries,
               int B, int k, uint64_t *ids, float *dists, float *scores, int *n_found) {
    if (d <= 0 || k < 0 || B < 0) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        mxo_hit *best = (mxo_hit *)malloc(sizeof(mxo_hit) * (size_t)(k > 0 ? k : 1));
        int cnt = 0;
        const float *q = queries + (size_t)b * d;
        for (uint64_t r = 0; r < n && k > 0; ++r) {
            mxo_hit h;
            h.dist = mxo_dist_cosine(q, corpus + (size_t)r * d, d);
            h.id = r + 1 + id_offset;
            topk_push(best, &cnt, k, h);
        }
        for (int j = 0; j < k; ++j) {
            size_t o = (size_t)b * k + j;
            if (j < cnt) {
                ids[o] = best[j].id;
                dists[o] = best[j].dist;
                scores[o] = mxo_score(best[j].dist);
            } else {
                ids[o] = 0;
                dists[o] = INFINITY;
                scores[o] = 0.0f;
            }
        }
        n_found[b] = cnt;
        free(best);
    }
    return 0;
}

/* all distances of one query (used by size-independent property tests and recall checks) */
void mxo_all_dists(const float *corpus, uint64_t n, int d, const float *q, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)n; ++r) out[r] = mxo_dist_cosine(q, corpus + (size_t)r * d, d);
}

/* merge G per-shard top-k lists (each ascending by (dist,id)) into one; shard lists [G, B, k] */
int mxo_merge(const uint64_t *ids, const float *dists, int G, int B, int k, uint64_t *out_ids,
              float *out_dists) {
    for (int b = 0; b < B; ++b) {
        mxo_hit *best = (mxo_hit *)malloc(sizeof(mxo_hit) * (size_t)(k > 0 ? k : 1));
        int cnt = 0;
        for (int g = 0; g < G; ++g)
            for (int j = 0; j < k; ++j) {
                size_t o = ((size_t)g * B + b) * k + j;
                if (ids[o] == 0) continue;
                mxo_hit h = {dists[o], ids[o]};
                topk_push(best, &cnt, k, h);
            }
        for (int j = 0; j < k; ++j) {
            size_t o = (size_t)b * k + j;
            out_ids[o] = j < cnt ? best[j].id : 0;
            out_dists[o] = j < cnt ? best[j].dist : INFINITY;
        }
        free(best);
    }
    return 0;
}

int mxo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
