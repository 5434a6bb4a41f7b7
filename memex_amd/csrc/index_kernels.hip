// index_kernels.hip -- everything around the streaming scan: row ingest (K8), query preparation,
// candidate-pool maintenance, exact DistCosine rescoring + ordering (K7), the all-f64 EXACT path
// and the multi-GPU merge.  Compiled with -ffp-contract=off: the exact arithmetic below must
// round exactly like the reference's scalar Rust code.
//
// Reference arithmetic being reproduced (hnsw_rs 0.1.20 DistCosine, called from
// lib/libmemex/src/storage/local.rs:65,76; score formula local.rs:86):
//   dot = sum_i f64(f32(q_i*c_i)), na = sum_i f64(f32(q_i*q_i)), nb = sum_i f64(f32(c_i*c_i))
//   dist = (na>0 && nb>0) ? f32(max(1 - dot/sqrt(na*nb), 0)) : 0 ;  score = 1 - (1/(1/dist)) in f32
#include <cmath>

#include "index_kernels.h"

namespace mx {

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_key(float f) {  // order-preserving map f32 -> u32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ float score_from_dist(float d) {
    const float t = __fdiv_rn(1.0f, d);  // d == 0 -> +inf
    const float u = __fdiv_rn(1.0f, t);  // +inf -> 0
    return __fsub_rn(1.0f, u);
}
__device__ __forceinline__ float dist_from_sums(double dot, double na, double nb) {
    if (na > 0.0 && nb > 0.0) {
        double d = __dsub_rn(1.0, __ddiv_rn(dot, __dsqrt_rn(__dmul_rn(na, nb))));
        if (d < 0.0) d = 0.0;
        return (float)d;  // round-to-nearest-even, like Rust `as f32`
    }
    return 0.0f;
}

// block-wide sum of a small unsigned value; every thread gets the result.  blockDim = 256.
__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t *lds4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

// ---------------------------------------------------------------------------------------------
// ingest: copy rows into the padded store and compute 1/|c|   (replaces hnsw.insert, local.rs:65)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ingest_kernel(const float *__restrict__ src, uint64_t n, int d,
                                                     float *__restrict__ x, float *__restrict__ scale,
                                                     uint64_t first, int ds, uint32_t *flags) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t r = wave0; r < n; r += nwaves) {
        const float *s = src + r * (uint64_t)d;
        float *o = x + (first + r) * (uint64_t)ds;
        double acc = 0.0;
        bool bad = false;
        for (int c = lane; c < ds; c += 64) {
            const float v = c < d ? s[c] : 0.0f;
            o[c] = v;
            bad |= !isfinite(v);
            acc += (double)v * (double)v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        const bool anybad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) {
            float sc;
            if (acc > 0.0) {
                sc = (float)(1.0 / sqrt(acc));
                if (acc < 1e-30 || acc > 1e30) atomicAdd(&flags[1], 1u);
            } else {
                sc = INFINITY;  // zero-norm row: exact dist is 0 for every query (DistCosine else-branch)
            }
            if (anybad || !isfinite(acc)) atomicAdd(&flags[0], 1u);
            scale[first + r] = sc;
        }
    }
}

hipError_t launch_ingest(hipStream_t s, const float *src, uint64_t n, int d, float *x, float *scale,
                         uint64_t first, int ds, uint32_t *flags) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ingest_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, n, d, x, scale, first, ds, flags);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// query preparation: one block per query slot (256 slots)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_queries_kernel(const float *__restrict__ q, int B, int d, int ds,
                                                           __bf16 *__restrict__ qfrag, float *__restrict__ qpad,
                                                           double *__restrict__ qnorm2, float *__restrict__ theta,
                                                           uint32_t *__restrict__ overflow,
                                                           uint32_t *__restrict__ pool_cnt) {
    extern __shared__ __attribute__((aligned(16))) char psm[];
    float *s_row = reinterpret_cast<float *>(psm);  // [d] this query (coalesced load; the chain reads LDS)
    __shared__ float s_inv;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const bool live = b < B;
    for (int i = tid; i < d; i += 256) s_row[i] = live ? q[(size_t)b * d + i] : 0.0f;
    __syncthreads();
    if (tid == 0) {
        double na = 0.0;
        for (int i = 0; i < d; ++i) {  // sequential, f32 products: DistCosine's query-norm chain
            const float v = s_row[i];
            na += (double)__fmul_rn(v, v);
        }
        qnorm2[b] = na;
        const bool usable = live && na > 0.0 && isfinite(na);
        s_inv = usable ? (float)(1.0 / sqrt(na)) : 0.0f;
        theta[b] = usable ? -INFINITY : INFINITY;  // zero / padded queries never pass the scan filter
        overflow[b] = 0;
        pool_cnt[b] = 0;
    }
    __syncthreads();
    const float inv = s_inv;
    const int ksteps = ds / 16;
    const int w = b >> 5, col = b & 31;
    for (int dim = tid; dim < ds; dim += 256) {
        const float v = dim < d ? s_row[dim] : 0.0f;
        qpad[(size_t)b * ds + dim] = v;
        // MFMA 32x32x16 B-operand: lane l holds B[k = 8*(l>>5)+i][n = l&31]
        const int ks = dim >> 4, hh = (dim >> 3) & 1, i = dim & 7;
        const int lane = hh * 32 + col;
        qfrag[(((size_t)w * ksteps + ks) * 64 + lane) * 8 + i] = (__bf16)(v * inv);
    }
}

hipError_t launch_prep_queries(hipStream_t s, const float *q, int B, int d, int ds, void *qfrag, float *qpad,
                               double *qnorm2, float *theta, uint32_t *overflow, uint32_t *pool_cnt) {
    hipLaunchKernelGGL(prep_queries_kernel, dim3(kMaxBatch), dim3(256), sizeof(float) * (size_t)d, s, q, B, d, ds, (__bf16 *)qfrag, qpad,
                       qnorm2, theta, overflow, pool_cnt);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// pool update: gather a stage's lane buffers, select the k-th best approximate score, prune
// ---------------------------------------------------------------------------------------------
constexpr int kPer = kPoolCap / 256;  // pool entries held per thread

__global__ __launch_bounds__(256) void update_kernel(int k, int nwg, const Cand *__restrict__ lane_buf,
                                                     const uint32_t *__restrict__ lane_cnt, Cand *pool_in,
                                                     Cand *pool_out, uint32_t *pool_cnt, float *theta,
                                                     uint32_t *overflow) {
    __shared__ uint32_t s_cnt;
    __shared__ uint32_t s_sel[2][4];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    Cand *pin = pool_in + (size_t)q * kPoolCap;
    Cand *pout = pool_out + (size_t)q * kPoolCap;
    const uint32_t m_old = pool_cnt[q];
    if (tid == 0) s_cnt = m_old;
    __syncthreads();

    // ---- gather: scan workgroup w kept this query's candidates in lanes L0 and L0+32 of wave q/32.
    // All loads of a lane buffer are issued before the first store (16-byte vectors, 2 entries each):
    // a load-store-load chain through HBM latency is what made this kernel slow.
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    for (int w = tid; w < nwg; w += 256) {
        // [thread-in-workgroup][workgroup] layout (scan.hip): consecutive threads read consecutive words
        const uint32_t t0 = (uint32_t)(q >> 5) * 64 + (uint32_t)(q & 31);
        const uint32_t l0 = t0 * (uint32_t)nwg + (uint32_t)w, l1 = (t0 + 32) * (uint32_t)nwg + (uint32_t)w;
        const uint32_t c0 = lane_cnt[l0], c1 = lane_cnt[l1];
        for (int half = 0; half < 2; ++half) {
            const uint32_t c = half ? c1 : c0;
            if (c == 0) continue;
            const uint32_t pos = atomicAdd(&s_cnt, c);
            const u32x4 *src = reinterpret_cast<const u32x4 *>(lane_buf + (size_t)(half ? l1 : l0) * kLaneCap);
            u32x4 v[kLaneCap / 2];
#pragma unroll
            for (int e = 0; e < kLaneCap / 2; ++e)
                if ((uint32_t)(2 * e) < c) v[e] = src[e];
#pragma unroll
            for (int e = 0; e < kLaneCap / 2; ++e) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const uint32_t i = 2 * e + hh;
                    if (i < c && pos + i < (uint32_t)kPoolCap) {
                        Cand cd;
                        cd.score = __uint_as_float(v[e][2 * hh]);
                        cd.row = v[e][2 * hh + 1];
                        if (!(cd.score == cd.score)) cd.score = 2.0f;  // NaN = zero-norm row: rank first
                        pin[pos + i] = cd;
                    }
                }
            }
        }
    }
    __syncthreads();
    uint32_t m = s_cnt;
    if (m > (uint32_t)kPoolCap) {
        if (tid == 0) overflow[q] = 1;
        m = kPoolCap;
    }
    __syncthreads();

    // ---- load my entries (entry e*256 + tid)
    uint32_t key[kPer];
    uint32_t row[kPer];
    const int per = (int)((m + 255) / 256);
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
        const uint32_t i = (uint32_t)e * 256 + tid;
        if (e < per && i < m) {
            const Cand cd = pin[i];
            key[e] = f32_key(cd.score);
            row[e] = cd.row;
        } else {
            key[e] = 0;  // below every real key
            row[e] = 0xffffffffu;
        }
    }

    const float th_old = theta[q];
    float th_new = th_old;
    uint32_t keep_key = 0;  // keep everything
    if (m >= (uint32_t)k && k > 0 && th_old != INFINITY) {
        // k-th largest key.  Only the bits below the highest bit in which the keys differ need a
        // decision; counts are wave ballots + popcounts (scalar), one LDS exchange and one barrier
        // per bit.
        uint32_t kmax = 0, kmin = 0xffffffffu;
#pragma unroll
        for (int e = 0; e < kPer; ++e)
            if (e < per && row[e] != 0xffffffffu) {
                kmax = key[e] > kmax ? key[e] : kmax;
                kmin = key[e] < kmin ? key[e] : kmin;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t a = __shfl_xor(kmax, o), b = __shfl_xor(kmin, o);
            kmax = a > kmax ? a : kmax;
            kmin = b < kmin ? b : kmin;
        }
        if ((tid & 63) == 0) {
            s_sel[0][tid >> 6] = kmax;
            s_sel[1][tid >> 6] = kmin;
        }
        __syncthreads();
        kmax = max(max(s_sel[0][0], s_sel[0][1]), max(s_sel[0][2], s_sel[0][3]));
        kmin = min(min(s_sel[1][0], s_sel[1][1]), min(s_sel[1][2], s_sel[1][3]));
        __syncthreads();
        const uint32_t diff = kmax ^ kmin;
        uint32_t prefix = kmax;
        if (diff != 0) {
            const int top = 31 - __clz((int)diff);
            prefix = top == 31 ? 0u : (kmax & ~((2u << top) - 1u));  // bits above `top` are common
            for (int bit = top; bit >= 0; --bit) {
                const uint32_t trial = prefix | (1u << bit);
                uint32_t c = 0;
#pragma unroll
                for (int e = 0; e < kPer; ++e)
                    if (e < per) c += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(key[e] >= trial));
                uint32_t *slot = s_sel[bit & 1];
                if ((tid & 63) == 0) slot[tid >> 6] = c;
                __syncthreads();  // slots alternate per bit, so one barrier per bit suffices
                c = slot[0] + slot[1] + slot[2] + slot[3];
                if (c >= (uint32_t)k) prefix = trial;
            }
        }
        const float kth = key_f32(prefix);
        th_new = kth - kMargin;
        if (th_new > th_old || th_old == -INFINITY) {
            keep_key = f32_key(th_new);
        } else {
            th_new = th_old;
        }
    }

    // ---- prune into pool_out
    __syncthreads();
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    uint32_t mine = 0;
#pragma unroll
    for (int e = 0; e < kPer; ++e) mine += (e < per && row[e] != 0xffffffffu && key[e] >= keep_key) ? 1u : 0u;
    uint32_t pos = mine ? atomicAdd(&s_cnt, mine) : 0;
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
        if (e < per && row[e] != 0xffffffffu && key[e] >= keep_key) {
            Cand cd;
            cd.score = key_f32(key[e]);
            cd.row = row[e];
            pout[pos++] = cd;
        }
    }
    __syncthreads();
    if (tid == 0) {
        pool_cnt[q] = s_cnt;
        theta[q] = th_new;
    }
}

hipError_t launch_update(hipStream_t s, int B, int k, int nwg, const Cand *lane_buf, const uint32_t *lane_cnt,
                         Cand *pool_in, Cand *pool_out, uint32_t *pool_cnt, float *theta, uint32_t *overflow) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(update_kernel, dim3(B), dim3(256), 0, s, k, nwg, lane_buf, lane_cnt, pool_in, pool_out,
                       pool_cnt, theta, overflow);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// exact rescoring + ordering of the candidate pool (K7)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float exact_dist_row(const float *__restrict__ qv, const float *__restrict__ row, int ds,
                                                double na, double *cos_out) {
    // sequential f64 accumulation of f32 products, element order 0..d-1 (zero padding adds +0.0)
    double dot = 0.0, nb = 0.0;
    const float4 *r4 = reinterpret_cast<const float4 *>(row);
#pragma unroll 8
    for (int i = 0; i < ds / 4; ++i) {  // ds is a multiple of 128: 8 independent loads in flight per trip
        const float4 c = r4[i];
        const float4 a = *reinterpret_cast<const float4 *>(qv + 4 * i);
        dot += (double)__fmul_rn(a.x, c.x); nb += (double)__fmul_rn(c.x, c.x);
        dot += (double)__fmul_rn(a.y, c.y); nb += (double)__fmul_rn(c.y, c.y);
        dot += (double)__fmul_rn(a.z, c.z); nb += (double)__fmul_rn(c.z, c.z);
        dot += (double)__fmul_rn(a.w, c.w); nb += (double)__fmul_rn(c.w, c.w);
    }
    if (cos_out) *cos_out = (na > 0.0 && nb > 0.0) ? dot / sqrt(na * nb) : 1.0;
    return dist_from_sums(dot, na, nb);
}

__global__ __launch_bounds__(256) void final_kernel(int k, int ds, const float *__restrict__ x, uint64_t n_rows,
                                                    uint64_t id_offset, const float *__restrict__ qpad,
                                                    const double *__restrict__ qnorm2, const Cand *__restrict__ pool,
                                                    const uint32_t *__restrict__ pool_cnt, uint32_t *overflow,
                                                    uint64_t *ids, float *scores, float *dists, int32_t *n_found,
                                                    float *max_err) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    uint64_t *keys = reinterpret_cast<uint64_t *>(fsm);                        // [kFinalCap]
    float *qv = reinterpret_cast<float *>(fsm + sizeof(uint64_t) * kFinalCap);  // [ds]
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const uint64_t want64 = n_rows < (uint64_t)k ? n_rows : (uint64_t)k;
    const int want = (int)want64;
    uint64_t *oid = ids + (size_t)q * k;
    float *osc = scores + (size_t)q * k;
    float *odi = dists ? dists + (size_t)q * k : nullptr;
    if (tid == 0) n_found[q] = want;
    for (int j = tid; j < k; j += 256) {  // defaults for unused slots
        oid[j] = 0;
        osc[j] = 0.0f;
        if (odi) odi[j] = INFINITY;
    }
    const double na = qnorm2[q];
    if (!(na > 0.0)) {
        // zero-norm query: DistCosine returns 0 for every row -> ties broken by id (local.rs:63 ids)
        for (int j = tid; j < want; j += 256) {
            oid[j] = id_offset + (uint64_t)j + 1;
            osc[j] = 1.0f;
            if (odi) odi[j] = 0.0f;
        }
        return;
    }
    const uint32_t m = pool_cnt[q];
    if (m > (uint32_t)kFinalCap || m < (uint32_t)want) {
        if (tid == 0) overflow[q] = 1;  // host re-answers this query on the EXACT path
        return;
    }
    for (int i = tid; i < ds; i += 256) qv[i] = qpad[(size_t)q * ds + i];
    __syncthreads();
    const Cand *pl = pool + (size_t)q * kPoolCap;
    float err = 0.0f;
    for (uint32_t c = tid; c < m; c += 256) {
        const Cand cd = pl[c];
        double cosv;
        const float d = exact_dist_row(qv, x + (size_t)cd.row * ds, ds, na, max_err ? &cosv : nullptr);
        keys[c] = ((uint64_t)__float_as_uint(d) << 32) | cd.row;
        if (max_err && cd.score < 1.5f) err = fmaxf(err, fabsf((float)cosv - cd.score));
    }
    if (max_err) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) err = fmaxf(err, __shfl_xor(err, o));
        if ((tid & 63) == 0) atomicMax(reinterpret_cast<unsigned int *>(max_err), __float_as_uint(err));
    }
    __syncthreads();
    for (uint32_t c = tid; c < m; c += 256) {
        const uint64_t me = keys[c];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; ++j) rank += keys[j] < me ? 1u : 0u;
        if (rank < (uint32_t)want) {
            const float d = __uint_as_float((uint32_t)(me >> 32));
            oid[rank] = id_offset + (uint64_t)(uint32_t)me + 1;
            osc[rank] = score_from_dist(d);
            if (odi) odi[rank] = d;
        }
    }
}

hipError_t launch_final(hipStream_t s, int B, int k, int d, int ds, const float *x, uint64_t n_rows,
                        uint64_t id_offset, const float *qpad, const double *qnorm2, const Cand *pool,
                        const uint32_t *pool_cnt, uint32_t *overflow, uint64_t *ids, float *scores, float *dists,
                        int32_t *n_found, float *max_err) {
    (void)d;
    if (B <= 0) return hipSuccess;
    const size_t lds = sizeof(uint64_t) * kFinalCap + sizeof(float) * (size_t)ds;
    hipLaunchKernelGGL(final_kernel, dim3(B), dim3(256), lds, s, k, ds, x, n_rows, id_offset, qpad, qnorm2, pool,
                       pool_cnt, overflow, ids, scores, dists, n_found, max_err);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// EXACT path: f64 DistCosine on every row, then a 64-step MSB-first select on (dist,row) keys
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void exact_keys_kernel(int ds, const float *__restrict__ x, uint64_t n_rows,
                                                         const float *__restrict__ qrow, uint64_t *__restrict__ keys) {
    // all LDS in the dynamic region (a static __shared__ in front would misalign the float4 reads)
    extern __shared__ __attribute__((aligned(16))) char esm[];
    float *qv = reinterpret_cast<float *>(esm);
    double *s_na = reinterpret_cast<double *>(esm + sizeof(float) * (size_t)ds);
    for (int i = threadIdx.x; i < ds; i += 256) qv[i] = qrow[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        double na = 0.0;
        for (int i = 0; i < ds; ++i) na += (double)__fmul_rn(qv[i], qv[i]);
        *s_na = na;
    }
    __syncthreads();
    const double na = *s_na;
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * 256) {
        const float d = exact_dist_row(qv, x + r * (uint64_t)ds, ds, na, nullptr);
        keys[r] = ((uint64_t)__float_as_uint(d) << 32) | (uint32_t)r;
    }
}

// sel_state: [0] prefix, [1] count, [2] output cursor
__global__ __launch_bounds__(256) void exact_count_kernel(const uint64_t *__restrict__ keys, uint64_t n, int bit,
                                                          uint64_t *state) {
    const uint64_t trial = state[0] | (1ull << bit);
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        c += keys[i] < trial ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(reinterpret_cast<unsigned long long *>(&state[1]), (unsigned long long)c);
}
__global__ void exact_decide_kernel(int bit, uint64_t kk, uint64_t *state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (state[1] < kk) state[0] |= (1ull << bit);  // fewer than k keys below trial: k-th smallest >= trial
        state[1] = 0;
    }
}
__global__ __launch_bounds__(256) void exact_collect_kernel(const uint64_t *__restrict__ keys, uint64_t n,
                                                            uint64_t kk, uint64_t *state, uint64_t *sel) {
    const uint64_t kth = state[0];
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t key = keys[i];
        if (key <= kth) {
            const unsigned long long p = atomicAdd(reinterpret_cast<unsigned long long *>(&state[2]), 1ull);
            if (p < kk) sel[p] = key;
        }
    }
}
__global__ __launch_bounds__(256) void exact_emit_kernel(int k, uint64_t kk, uint64_t id_offset,
                                                         const uint64_t *__restrict__ sel, uint64_t *ids,
                                                         float *scores, float *dists, int32_t *n_found) {
    const int tid = threadIdx.x;
    if (tid == 0) *n_found = (int32_t)kk;
    for (int j = tid; j < k; j += 256) {
        ids[j] = 0;
        scores[j] = 0.0f;
        if (dists) dists[j] = INFINITY;
    }
    __syncthreads();
    for (uint64_t c = tid; c < kk; c += 256) {
        const uint64_t me = sel[c];
        uint64_t rank = 0;
        for (uint64_t j = 0; j < kk; ++j) rank += sel[j] < me ? 1u : 0u;
        const float d = __uint_as_float((uint32_t)(me >> 32));
        ids[rank] = id_offset + (uint64_t)(uint32_t)me + 1;
        scores[rank] = score_from_dist(d);
        if (dists) dists[rank] = d;
    }
}

hipError_t launch_exact_query(hipStream_t s, int k, int d, int ds, const float *x, uint64_t n_rows,
                              uint64_t id_offset, const float *qpad_row, uint64_t *keys, uint64_t *sel_state,
                              uint64_t *ids, float *scores, float *dists, int32_t *n_found) {
    (void)d;
    const uint64_t kk = n_rows < (uint64_t)k ? n_rows : (uint64_t)k;
    hipError_t e = hipMemsetAsync(sel_state, 0, 3 * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    uint64_t *sel = sel_state + 4;  // [k] selected keys
    if (kk > 0) {
        unsigned blocks = (unsigned)((n_rows + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(exact_keys_kernel, dim3(blocks), dim3(256), sizeof(float) * (size_t)ds + 16, s, ds, x, n_rows,
                           qpad_row, keys);
        unsigned cblocks = blocks > 1024 ? 1024 : blocks;
        for (int bit = 63; bit >= 0; --bit) {
            hipLaunchKernelGGL(exact_count_kernel, dim3(cblocks), dim3(256), 0, s, keys, n_rows, bit, sel_state);
            hipLaunchKernelGGL(exact_decide_kernel, dim3(1), dim3(64), 0, s, bit, kk, sel_state);
        }
        hipLaunchKernelGGL(exact_collect_kernel, dim3(cblocks), dim3(256), 0, s, keys, n_rows, kk, sel_state, sel);
    }
    hipLaunchKernelGGL(exact_emit_kernel, dim3(1), dim3(256), 0, s, k, kk, id_offset, sel, ids, scores, dists, n_found);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// multi-GPU merge of per-shard top-k lists (after the RCCL all-gather)
// ---------------------------------------------------------------------------------------------
// shard g's lists: ids at ids_base + g*ids_stride, dists at dists_base + g*dists_stride (bytes), each [B, k]
__global__ __launch_bounds__(256) void merge_kernel(const char *__restrict__ ids_base, size_t ids_stride,
                                                    const char *__restrict__ dists_base, size_t dists_stride,
                                                    int G, int B, int k, uint64_t *out_ids, float *out_dists,
                                                    float *out_scores) {
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int total = G * k;
    for (int j = tid; j < k; j += 256) {
        out_ids[(size_t)b * k + j] = 0;
        out_dists[(size_t)b * k + j] = INFINITY;
        if (out_scores) out_scores[(size_t)b * k + j] = 0.0f;
    }
    __syncthreads();
    for (int c = tid; c < total; c += 256) {
        const int g = c / k, j = c % k;
        const size_t o = (size_t)b * k + j;
        const uint64_t id = reinterpret_cast<const uint64_t *>(ids_base + (size_t)g * ids_stride)[o];
        if (id == 0) continue;
        const float d = reinterpret_cast<const float *>(dists_base + (size_t)g * dists_stride)[o];
        int rank = 0;
        for (int c2 = 0; c2 < total; ++c2) {
            const int g2 = c2 / k;
            const size_t o2 = (size_t)b * k + (c2 % k);
            const uint64_t id2 = reinterpret_cast<const uint64_t *>(ids_base + (size_t)g2 * ids_stride)[o2];
            if (id2 == 0) continue;
            const float d2 = reinterpret_cast<const float *>(dists_base + (size_t)g2 * dists_stride)[o2];
            rank += (d2 < d || (d2 == d && id2 < id)) ? 1 : 0;
        }
        if (rank < k) {
            out_ids[(size_t)b * k + rank] = id;
            out_dists[(size_t)b * k + rank] = d;
            if (out_scores) out_scores[(size_t)b * k + rank] = score_from_dist(d);
        }
    }
}

hipError_t launch_merge(hipStream_t s, const void *ids, size_t ids_stride, const void *dists, size_t dists_stride,
                        int G, int B, int k, uint64_t *out_ids, float *out_dists, float *out_scores) {
    if (B <= 0 || k <= 0) return hipSuccess;
    hipLaunchKernelGGL(merge_kernel, dim3(B), dim3(256), 0, s, static_cast<const char *>(ids), ids_stride,
                       static_cast<const char *>(dists), dists_stride, G, B, k, out_ids, out_dists, out_scores);
    return hipGetLastError();
}

}  // namespace mx
