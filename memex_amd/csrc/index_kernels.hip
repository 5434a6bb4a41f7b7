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
#include "mx_rotate.h"

namespace mx {

typedef __attribute__((ext_vector_type(4))) float f32x4;

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


// k-th largest of the u32 keys a block holds in registers (PER per thread; valid[e] marks real
// entries).  MSB-first radix select over the bits in which the keys actually differ (scores of one
// query share sign, exponent and leading mantissa bits), up to 8 bits per pass: LDS histogram of the
// digit among the keys that still match the prefix, one wave walks the bins from the top.
// hist: 256 words of LDS; s_pick: 2 words.  Requires 1 <= k <= number of valid keys, blockDim <= 1024.
template <int PER>
__device__ __forceinline__ uint32_t block_kth_largest(const uint32_t (&key)[PER], const bool (&valid)[PER], uint32_t k,
                                                      uint32_t *hist, uint32_t *s_pick) {
    const int tid = threadIdx.x;
    uint32_t kmax = 0, kmin = 0xffffffffu;
#pragma unroll
    for (int e = 0; e < PER; ++e)
        if (valid[e]) {
            kmax = key[e] > kmax ? key[e] : kmax;
            kmin = key[e] < kmin ? key[e] : kmin;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(kmax, o), b = __shfl_xor(kmin, o);
        kmax = a > kmax ? a : kmax;
        kmin = b < kmin ? b : kmin;
    }
    __syncthreads();
    if ((tid & 63) == 0) {
        hist[tid >> 6] = kmax;
        hist[16 + (tid >> 6)] = kmin;
    }
    __syncthreads();
    const int nw = (int)(blockDim.x >> 6);
    for (int w = 0; w < nw; ++w) {
        kmax = hist[w] > kmax ? hist[w] : kmax;
        kmin = hist[16 + w] < kmin ? hist[16 + w] : kmin;
    }
    const uint32_t diff = kmax ^ kmin;
    if (diff == 0) return kmax;
    int hi = 31 - __clz((int)diff);  // highest bit in which two keys differ
    uint32_t mask = hi == 31 ? 0u : ~((2u << hi) - 1u);
    uint32_t prefix = kmax & mask;
#pragma unroll 1
    while (hi >= 0) {
        const int lo = hi >= 7 ? hi - 7 : 0;
        const uint32_t dmask = (1u << (hi - lo + 1)) - 1u;
        __syncthreads();  // the previous pass's picks / the reduction slots are consumed
        for (int i = tid; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < PER; ++e)
            if (valid[e] && (key[e] & mask) == prefix) atomicAdd(&hist[(key[e] >> lo) & dmask], 1u);
        __syncthreads();
        if (tid < 64) {
            // lane l owns bins 4l .. 4l+3; above(l) = keys in bins owned by higher lanes
            const uint32_t b0 = hist[4 * tid], b1 = hist[4 * tid + 1], b2 = hist[4 * tid + 2], b3 = hist[4 * tid + 3];
            const uint32_t mine = b0 + b1 + b2 + b3;
            uint32_t inc = mine;  // inclusive suffix sum over lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_down(inc, o);
                if (tid + o < 64) inc += t;
            }
            const uint32_t above = inc - mine;
            if (above < k && k <= inc) {  // the k-th largest key has its digit in one of my bins
                uint32_t a = above;
                int dg;
                if (a + b3 >= k) dg = 3;
                else if ((a += b3) + b2 >= k) dg = 2;
                else if ((a += b2) + b1 >= k) dg = 1;
                else { a += b1; dg = 0; }
                s_pick[0] = (uint32_t)(4 * tid + dg);
                s_pick[1] = k - a;  // rank inside that bin
            }
        }
        __syncthreads();
        prefix |= s_pick[0] << lo;
        mask |= dmask << lo;
        k = s_pick[1];
        hi = lo - 1;
    }
    return prefix;
}

// ---------------------------------------------------------------------------------------------
// ingest: copy rows into the padded store and compute 1/|c|   (replaces hnsw.insert, local.rs:65)
// ---------------------------------------------------------------------------------------------
// raw & 1: the rows are stored values of a compressed corpus coming back from disk (unit length up to
// bf16 rounding): they must be stored as they are, so 1/|c| is replaced by 1.  raw & 2 (f32 corpus): rows whose norm is
// outside [1e-15, 1e15] are stored as zeros and listed in wild_rows (below).
__global__ __launch_bounds__(256) void ingest_kernel(const float *__restrict__ src, uint64_t n, int d,
                                                     float *__restrict__ x, float *__restrict__ scale,
                                                     uint64_t first, int ds, uint32_t *flags, int raw,
                                                     uint32_t *__restrict__ zero_rows, uint32_t *__restrict__ wild_rows,
                                                     uint64_t row_base) {
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
            const bool wild = acc > 0.0 && (acc < 1e-30 || acc > 1e30) && isfinite(acc);
            if (wild) atomicAdd(&flags[1], 1u);
            if (wild && (raw & 2) && !anybad) {
                // a norm outside [1e-15, 1e15]: the f32 stages are not certified for it (1/|c| and c/|c| leave the
                // f32 range).  The row is stored as zeros in every filter copy (1/|c| kept as 0) and listed: the scan
                // and the f32 stages never see it, finish_kernel hands it to its f64 stage for EVERY query, beside
                // the survivors.  One such row costs one more f64 chain per query, not the whole collection its fast path.
                sc = 0.0f;
                const uint32_t at = atomicAdd(&flags[4], 1u);
                if (at < (uint32_t)kWildCap) wild_rows[at] = (uint32_t)(row_base + r);
            } else if (acc > 0.0) {
                sc = (raw & 1) ? 1.0f : (float)(1.0 / sqrt(acc));
            } else {
                // zero-norm row: exact dist is 0 for every query (DistCosine else-branch).  It is stored as
                // zeros (scan score 0) and remembered in the index's zero-row list, from which
                // finish_kernel adds it to every query's candidates.
                sc = 0.0f;
                if (!anybad && acc == 0.0) {
                    const uint32_t at = atomicAdd(&flags[3], 1u);
                    if (at < (uint32_t)kZeroCap) zero_rows[at] = (uint32_t)(row_base + r);
                }
            }
            if (anybad || !isfinite(acc)) atomicAdd(&flags[0], 1u);
            scale[first + r] = sc;
        }
    }
}

hipError_t launch_ingest(hipStream_t s, const float *src, uint64_t n, int d, float *x, float *scale,
                         uint64_t first, int ds, uint32_t *flags, int raw, uint32_t *zero_rows, uint32_t *wild_rows, uint64_t row_base) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ingest_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, n, d, x, scale, first, ds, flags, raw, zero_rows, wild_rows, row_base);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// query preparation: one block per query slot (256 slots)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_queries_kernel(const float *__restrict__ q, int B, int d, int ds,
                                                           __bf16 *__restrict__ qfrag, float *__restrict__ qpad,
                                                           double *__restrict__ qnorm2, float *__restrict__ theta,
                                                           float *__restrict__ e1, const uint32_t *__restrict__ ec_max,
                                                           uint32_t *__restrict__ overflow, uint32_t *__restrict__ flags,
                                                           int filt8, float *__restrict__ qscale,
                                                           float *__restrict__ qa, float *__restrict__ qb,
                                                           const float *__restrict__ mean, float *__restrict__ qmean,
                                                           const uint32_t *__restrict__ rc_max) {
    extern __shared__ __attribute__((aligned(16))) char psm[];
    float *s_row = reinterpret_cast<float *>(psm);  // [d] this query (coalesced load; the chain reads LDS)
    __shared__ float s_inv;
    __shared__ uint32_t s_red[4];
    __shared__ uint32_t s_bad[4];
    __shared__ double s_dot[4];
    __shared__ float s_rq[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const bool live = b < B;
    bool bad = false;
    for (int i = tid; i < d; i += 256) {
        const float v = live ? q[(size_t)b * d + i] : 0.0f;
        s_row[i] = v;
        bad |= !isfinite(v);
    }
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
    }
    __syncthreads();
    const float inv = s_inv;
    const int w = b >> 5, col = b & 31;
    float r2 = 0.0f;  // |stored(q/|q|) - q/|q||^2: this query's share of the scan's error bound
    if (filt8) {
        // 8-bit filter copy (scan8.hip): the normalised query goes through the same rotation as the rows
        // (mx_rotate.h), then q8 = rint(rotated / s_q), s_q = max |rotated_i| / 127
        float *rin = s_row + d, *rot = rin + ds, *mixq = rot + ds;  // behind the query: [ds] | [ds] | [144]
        // centred copy: a_q = (q/|q|) . mean in f64, the quantiser sees r_q = q/|q| - a_q mean (as the rows' builder, shadow8_kernel)
        float aq8 = 0.0f, rq2 = 0.0f;
        if (mean) {
            double dp = 0.0;
            for (int dim = tid; dim < d; dim += 256) dp += (double)(s_row[dim] * inv) * (double)mean[dim];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dp += __shfl_xor(dp, o);
            if ((tid & 63) == 0) s_dot[tid >> 6] = dp;
            __syncthreads();
            aq8 = (float)(s_dot[0] + s_dot[1] + s_dot[2] + s_dot[3]);
            if (tid == 0) qmean[b] = aq8;
        }
        for (int dim = tid; dim < ds; dim += 256) {
            float vn = dim < d ? s_row[dim] * inv : 0.0f;
            if (mean && dim < d) vn = __builtin_fmaf(-aq8, mean[dim], vn);
            rin[dim] = vn;
            rq2 += vn * vn;
        }
        if (mean) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) rq2 += __shfl_xor(rq2, o);
            if ((tid & 63) == 0) s_rq[tid >> 6] = rq2;
        }
        rot_fill_mix(mixq, ds >> 7, tid, 256);
        __syncthreads();
        if (tid < 64) rot_wave(rin, rot, ds >> 7, tid, mixq);
        __syncthreads();
        float mx = 0.0f;
        for (int dim = tid; dim < ds; dim += 256) mx = fmaxf(mx, fabsf(rot[dim]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((tid & 63) == 0) s_red[tid >> 6] = __float_as_uint(mx);
        __syncthreads();
        mx = fmaxf(fmaxf(__uint_as_float(s_red[0]), __uint_as_float(s_red[1])), fmaxf(__uint_as_float(s_red[2]), __uint_as_float(s_red[3])));
        __syncthreads();  // s_red is reused for the residual below
        // (centred copy: the step stays at or above kMinStep8, as the rows' does -- scan8.hip's accumulator start a_q a_c / (s_h s_q)
        // must fit an int32; a query whose residual is that short loses nothing it can measure: r2 below is its actual residual)
        const float sq = (mean && mx > 0.0f) ? fmaxf(mx / 127.0f, kMinStep8) : mx / 127.0f;
        const float qinv = mx > 0.0f ? (mean ? 1.0f / sq : 127.0f / mx) : 0.0f;
        if (tid == 0) qscale[b] = sq;
        const int ksteps = ds / 32;
        int8_t *q8 = reinterpret_cast<int8_t *>(qfrag);
        for (int dim = tid; dim < ds; dim += 256) {
            const float v = dim < d ? s_row[dim] : 0.0f;
            qpad[(size_t)b * ds + dim] = v;  // the rescoring stages read the query as it came
            // v_mfma_i32_32x32x32_i8 B-operand: lane l holds B[k = 16*(l>>5)+i][n = l&31], 16 int8
            const int ks = dim >> 5, hh = (dim >> 4) & 1, i = dim & 15;
            const int lane = hh * 32 + col;
            const float vn = rot[dim];
            const float qv = fminf(fmaxf(rintf(vn * qinv), -127.0f), 127.0f);
            q8[(((size_t)w * ksteps + ks) * 64 + lane) * 16 + i] = (int8_t)(int)qv;
            const float r = qv * sq - vn;
            r2 += r * r;
        }
    } else {
        // centred bf16 copy (ScanParams::amean): a_q = (q/|q|) . mean in f64, the fragments hold r_q = q/|q| - a_q mean
        float aq = 0.0f;
        if (mean) {
            double dp = 0.0;
            for (int dim = tid; dim < d; dim += 256) dp += (double)(s_row[dim] * inv) * (double)mean[dim];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dp += __shfl_xor(dp, o);
            if ((tid & 63) == 0) s_dot[tid >> 6] = dp;
            __syncthreads();
            aq = (float)(s_dot[0] + s_dot[1] + s_dot[2] + s_dot[3]);
            if (tid == 0) qmean[b] = aq;
        }
        const int ksteps = ds / 16;
        float rq2 = 0.0f;  // |r_q|^2 (centred) -- the factor of the rows' residual in the bound
        for (int dim = tid; dim < ds; dim += 256) {
            const float v = dim < d ? s_row[dim] : 0.0f;
            qpad[(size_t)b * ds + dim] = v;
            // MFMA 32x32x16 B-operand: lane l holds B[k = 8*(l>>5)+i][n = l&31]
            const int ks = dim >> 4, hh = (dim >> 3) & 1, i = dim & 7;
            const int lane = hh * 32 + col;
            float vn = v * inv;
            if (mean && dim < d) vn = __builtin_fmaf(-aq, mean[dim], vn);
            const __bf16 vb = (__bf16)vn;
            qfrag[(((size_t)w * ksteps + ks) * 64 + lane) * 8 + i] = vb;
            const float r = (float)vb - vn;
            r2 += r * r;
            rq2 += vn * vn;
        }
        if (mean) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) rq2 += __shfl_xor(rq2, o);
            if ((tid & 63) == 0) s_rq[tid >> 6] = rq2;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) r2 += __shfl_xor(r2, o);
    const bool anybad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if ((tid & 63) == 0) {
        s_red[tid >> 6] = __float_as_uint(r2);
        s_bad[tid >> 6] = anybad ? 1u : 0u;
    }
    __syncthreads();
    if (tid == 0) {
        // non-finite query: the call fails with MX_EINVAL.  One word per query slot, written by every launch:
        // nothing to clear between calls
        flags[b] = live ? (s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) : 0u;
        // |approx - cos| <= |c^ - c| + |q^ - q| + |c^ - c||q^ - q| + f32 accumulation (unit vectors,
        // Cauchy-Schwarz).  ec_max is the largest row residual the filter copy holds (shadow_kernel);
        // without a filter copy (f32 scan: rows are rounded unnormalised) the a-priori bound is used.
        const float eq = sqrtf(__uint_as_float(s_red[0]) + __uint_as_float(s_red[1]) + __uint_as_float(s_red[2]) +
                               __uint_as_float(s_red[3])) * 1.01f + 1e-6f;
        float e = kApproxErr;
        if (ec_max) {
            const float ec = __uint_as_float(*ec_max) * 1.01f + 1e-6f;
            e = ec + eq + ec * eq + kAccSlack;
            if (!filt8) e = fminf(kApproxErr + ec * ec, e);  // two bf16 roundings have an a-priori bound as well
            if (mean && !filt8) {
                // centred copy: cos = a_q a_c + r_q . r_c exactly (a_c, a_q are the stored f32 values, r := unit vector - a m),
                // |r^_q . r^_c - r_q . r_c| <= |r_q| Ec + (|r_c| + Ec) Eq <= |r_q| Ec + Eq + Ec Eq   (|r_c| <= 1 + 1e-6)
                const float rq = sqrtf(s_rq[0] + s_rq[1] + s_rq[2] + s_rq[3]) * 1.001f + 1e-6f;
                e = fminf(e, rq * ec + eq * 1.0001f + ec * eq + kAccSlack);
            }
        }
        // the bound of ONE row is a + b * (its residual): with a residual per half tile (8-bit copy, scan8.hip) the
        // rows of a well-conditioned half tile get a tighter bound than e1, which is the bound of the worst one;
        // the other scans know one residual for all rows: a = e1, b = 0
        float a8 = eq + kAccSlack, b8 = 1.0f + eq;
        if (mean && filt8) {
            // centred int8 copy: cos = a_q a_c + r_q . r_c exactly; |r^_q . r^_c - r_q . r_c| <= |r_q| e_h + (|r_c| + e_h) Eq
            //   = Eq |r_c| + (|r_q| + Eq) e_h,  |r_c| <= rc_max (measured by the builder; 1 + 1e-6 without it)
            const float rq = sqrtf(s_rq[0] + s_rq[1] + s_rq[2] + s_rq[3]) * 1.001f + 1e-6f;
            const float rc = rc_max ? fminf(__uint_as_float(*rc_max) * 1.001f + 1e-6f, 1.0f + 1e-6f) : 1.0f + 1e-6f;
            a8 = eq * rc * 1.0001f + kAccSlack;
            b8 = rq + eq;
            if (ec_max) e = a8 + b8 * (__uint_as_float(*ec_max) * 1.01f + 1e-6f);  // the bound of a row of the worst half tile
        }
        e1[b] = e;
        qa[b] = filt8 ? a8 : e;
        qb[b] = filt8 ? b8 : 0.0f;
    }
}

hipError_t launch_prep_queries(hipStream_t s, const float *q, int B, int d, int ds, void *qfrag, float *qpad,
                               double *qnorm2, float *theta, float *e1, const uint32_t *ec_max, uint32_t *overflow,
                               uint32_t *flags, float *qa, float *qb, bool filt8, float *qscale, const float *mean, float *qmean,
                               const uint32_t *rc_max) {
    if (!qmean) mean = nullptr;
    // one block per query slot: 256 slots (a pass computes all of them), 512 for a batch of more than 256
    hipLaunchKernelGGL(prep_queries_kernel, dim3(B > kPassBatch ? kMaxBatch : kPassBatch), dim3(256),
                       sizeof(float) * ((size_t)d + (filt8 ? 2 * (size_t)ds + kRotMaxBlocks * kRotMaxBlocks : 0)), s, q, B, d, ds,
                       (__bf16 *)qfrag, qpad, qnorm2, theta, e1, ec_max, overflow, flags, filt8 ? 1 : 0, qscale, qa, qb, mean, qmean, rc_max);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// mean direction of the normalised rows (the centre of a centred bf16 copy, launch_shadow)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_sum_kernel(const float *__restrict__ x, const float *__restrict__ scale, uint64_t n, int ds,
                                                       float *__restrict__ msum) {
    // a workgroup walks a contiguous share of the rows; thread t sums columns t, t + 256, ... (coalesced along a row)
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t r0 = (uint64_t)blockIdx.x * per, r1 = r0 + per < n ? r0 + per : n;
    for (int c = threadIdx.x; c < ds; c += 256) {
        float acc = 0.0f;
        for (uint64_t r = r0; r < r1; ++r) acc = __builtin_fmaf(x[r * (uint64_t)ds + c], scale[r], acc);
        atomicAdd(&msum[c], acc);
    }
}

__global__ __launch_bounds__(256) void mean_norm_kernel(float *__restrict__ msum, int ds, float *__restrict__ mean) {
    __shared__ double s_p[4];
    double p = 0.0;
    for (int c = threadIdx.x; c < ds; c += 256) p += (double)msum[c] * (double)msum[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = p;
    __syncthreads();
    const double n2 = s_p[0] + s_p[1] + s_p[2] + s_p[3];
    const double inv = n2 > 1e-30 && n2 < 1e30 ? 1.0 / sqrt(n2) : 0.0;  // a vanishing (or non-finite) sum: no centre
    for (int c = threadIdx.x; c < ds; c += 256) mean[c] = (float)((double)msum[c] * inv);
    __syncthreads();
    if (threadIdx.x == 0) msum[ds] = (float)sqrt(n2);  // |sum of the unit rows|: the host decides whether centring pays
}

hipError_t launch_mean_dir(hipStream_t s, const float *x, const float *scale, uint64_t n, int ds, float *msum, float *mean) {
    hipError_t e = hipMemsetAsync(msum, 0, (size_t)ds * sizeof(float), s);
    if (e != hipSuccess) return e;
    if (n > 0) {
        const unsigned blocks = (unsigned)(n < 2048 ? n : 2048);
        hipLaunchKernelGGL(mean_sum_kernel, dim3(blocks), dim3(256), 0, s, x, scale, n, ds, msum);
    }
    hipLaunchKernelGGL(mean_norm_kernel, dim3(1), dim3(256), 0, s, msum, ds, mean);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// theta: pass threshold of the collect launch from the sample launch's lane maxima
// ---------------------------------------------------------------------------------------------
// A query's 2*nwg lane maxima are approximate scores of 2*nwg DISTINCT rows, so the k-th largest of
// them, a_k, is a lower bound of the k-th best approximate score, the exact k-th best cosine is
// >= a_k - e1, and every row of the exact top-k has an approximate score >= a_k - 2*e1.
// Per-row bounds (scan8_kernel): the lane maxima already are lower bounds of cosines, L_k is a lower bound of the
// k-th best cosine, a row of the top-k has score + (qa + qb * residual) >= L_k, and the scan tests
// score >= theta - qb * residual with theta = L_k - qa.  `raw` = 1: lane maxima are plain scores, theta = a_k - 2*qa
// (qa = e1 there).
__global__ __launch_bounds__(256) void theta_kernel(int k, int nwg, const float *__restrict__ lane_max,
                                                    const float *__restrict__ qa, int raw, float *__restrict__ theta) {
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_pick[2];
    __shared__ uint32_t s_nvalid[4];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = 2 * nwg;  // <= 512: two values per thread
    const uint32_t t0 = (uint32_t)(q >> 5) * 64 + (uint32_t)(q & 31);
    uint32_t key[2];
    bool valid[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int i = e * 256 + tid;
        float v = -INFINITY;
        if (i < n) {
            const int hh = i / nwg, w = i - hh * nwg;
            v = lane_max[(size_t)(t0 + 32 * hh) * nwg + w];
        }
        valid[e] = v > -INFINITY;  // a lane that saw no row keeps -inf
        key[e] = f32_key(v);
    }
    uint32_t nv = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(valid[0])) + (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(valid[1]));
    if ((tid & 63) == 0) s_nvalid[tid >> 6] = nv;
    __syncthreads();
    nv = s_nvalid[0] + s_nvalid[1] + s_nvalid[2] + s_nvalid[3];
    float kth = -INFINITY;  // fewer than k lanes saw a row: no threshold
    if (nv >= (uint32_t)k && k > 0) kth = key_f32(block_kth_largest<2>(key, valid, (uint32_t)k, s_hist, s_pick));
    if (tid == 0 && theta[q] != INFINITY) theta[q] = kth - (raw ? 2.0f : 1.0f) * qa[q];
}

hipError_t launch_theta(hipStream_t s, int B, int k, int nwg, const float *lane_max, const float *qa, bool raw, float *theta) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(theta_kernel, dim3(B), dim3(256), 0, s, k, nwg, lane_max, qa, raw ? 1 : 0, theta);
    return hipGetLastError();
}

// retry launch: queries whose lane buffers overflowed are scanned again with the tight threshold
// finish_kernel derived from what was collected; every other query is parked at +inf
__global__ void retry_setup_kernel(float *theta, const float *theta_retry, uint32_t *overflow, uint32_t *todo) {
    const int q = threadIdx.x;
    const bool again = overflow[q] == 1;
    todo[q] = again ? 1u : 0u;
    theta[q] = again ? theta_retry[q] : INFINITY;
    if (again) overflow[q] = 0;
}

hipError_t launch_retry_setup(hipStream_t s, float *theta, const float *theta_retry, uint32_t *overflow, uint32_t *todo) {
    hipLaunchKernelGGL(retry_setup_kernel, dim3(1), dim3(kMaxBatch), 0, s, theta, theta_retry, overflow, todo);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// finish: gather a query's candidates, prune by approximate score, rescore in f32, prune again,
// exact DistCosine on the survivors, order by (dist, id), emit            (K7)
// ---------------------------------------------------------------------------------------------
// 4 consecutive dims (4*c4 ..) of a row.  CMP = false: the f32 store.  CMP = true (compressed corpus:
// the bf16 filter copy is the ONLY copy): the 8-byte half of the 16-byte MFMA fragment holding those
// dims (layout: launch_shadow), widened to f32 exactly; a zero-norm row is stored as zeros.
template <bool CMP>
__device__ __forceinline__ float4 row_load4(const float *__restrict__ x, const void *__restrict__ xh, int ds, uint32_t row,
                                            int c4) {
    if (!CMP) return reinterpret_cast<const float4 *>(x + (size_t)row * ds)[c4];
    const uint32_t t = row >> 5, ks = (uint32_t)c4 >> 2, hh = ((uint32_t)c4 >> 1) & 1u, l = (row & 31u) + 32u * hh;
    const uint2 v = reinterpret_cast<const uint2 *>(xh)[(((size_t)t * (uint32_t)(ds >> 4) + ks) * 64 + l) * 2 + (c4 & 1)];
    float4 f;
    f.x = __uint_as_float(v.x << 16);
    f.y = __uint_as_float(v.x & 0xffff0000u);
    f.z = __uint_as_float(v.y << 16);
    f.w = __uint_as_float(v.y & 0xffff0000u);
    f.x = f.x == f.x ? f.x : 0.0f;
    f.y = f.y == f.y ? f.y : 0.0f;
    f.z = f.z == f.z ? f.z : 0.0f;
    f.w = f.w == f.w ? f.w : 0.0f;
    return f;
}

// exact DistCosine of query qv (LDS) against a stored row, read through row_load4
template <bool CMP>
__device__ __forceinline__ float exact_dist_stored(const float *__restrict__ qv, const float *__restrict__ x,
                                                   const void *__restrict__ xh, int ds, uint32_t row, double na) {
    double dot = 0.0, nb = 0.0;
#pragma unroll 8
    for (int i = 0; i < ds / 4; ++i) {
        const float4 c = row_load4<CMP>(x, xh, ds, row, i);
        const float4 a = *reinterpret_cast<const float4 *>(qv + 4 * i);
        dot += (double)__fmul_rn(a.x, c.x); nb += (double)__fmul_rn(c.x, c.x);
        dot += (double)__fmul_rn(a.y, c.y); nb += (double)__fmul_rn(c.y, c.y);
        dot += (double)__fmul_rn(a.z, c.z); nb += (double)__fmul_rn(c.z, c.z);
        dot += (double)__fmul_rn(a.w, c.w); nb += (double)__fmul_rn(c.w, c.w);
    }
    return dist_from_sums(dot, na, nb);
}

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

constexpr int kFinThreads = 1024;
constexpr int kFinishTailBytes = 128 + (2 * kMaxScanWGs + 1 + 3) * 4;  // staged row ids [32] + scanned record counts
constexpr int kFinWaves = kFinThreads / 64;
constexpr int kFinPer = kCandCap / kFinThreads;  // candidates held per thread
static_assert(kCandCap % kFinThreads == 0, "candidate capacity must divide over the block");

// block-wide exclusive scan of a small unsigned value over 1024 threads; total in *total
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t v, uint32_t *lds_w, uint32_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) lds_w[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kFinWaves; ++w) {
        const uint32_t c = lds_w[w];
        base += w < wave ? c : 0u;
        tot += c;
    }
    *total = tot;
    return base + inc - v;
}

template <bool CMP>
__device__ __forceinline__ void finish_query(const FinishParams &p) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    Cand *ent = reinterpret_cast<Cand *>(fsm);                                        // [kCandCap]
    uint64_t *keys = reinterpret_cast<uint64_t *>(fsm);                               // same storage, later
    float *qv = reinterpret_cast<float *>(fsm + sizeof(Cand) * (size_t)kCandCap);     // [ds] raw query
    __shared__ uint32_t s_w[kFinWaves];
    __shared__ uint32_t s_sel[2][kFinWaves];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_pick[2];
    __shared__ uint32_t s_cnt;

    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if (p.todo && !p.todo[q]) return;
    const int k = p.k, ds = p.ds;
    const uint64_t want64 = p.n_rows < (uint64_t)k ? p.n_rows : (uint64_t)k;
    const int want = (int)want64;
    uint64_t *oid = p.ids + (size_t)q * k;
    float *osc = p.scores + (size_t)q * k;
    float *odi = p.dists ? p.dists + (size_t)q * k : nullptr;
    if (tid == 0) {
        p.n_found[q] = want;
        p.cand_cnt[q] = 0;
    }
    for (int j = tid; j < k; j += kFinThreads) {  // defaults for unused slots
        oid[j] = 0;
        osc[j] = 0.0f;
        if (odi) odi[j] = INFINITY;
    }
    if (want == 0) return;
    const double na = p.qnorm2[q];
    if (!(na > 0.0)) {
        // zero-norm query: DistCosine returns 0 for every row -> ties broken by id (local.rs:63 ids)
        for (int j = tid; j < want; j += kFinThreads) {
            oid[j] = p.idmap.id_of((uint32_t)j);
            osc[j] = 1.0f;
            if (odi) odi[j] = 0.0f;
        }
        return;
    }
    const uint32_t lane_ovf = p.overflow[q];
    // bound of a row's filter score: qa + qb * (residual of its half tile); one residual for all rows: qb = 0, qa = e1
    const float qa = p.qa[q], qb = p.qb[q];
    auto resid = [&](uint32_t row) { return p.terr ? p.terr[kTscaleFloats * (size_t)(row >> 6) + 2 + ((row >> 5) & 1u)] : 0.0f; };
    for (int i = tid; i < ds; i += kFinThreads) qv[i] = p.qpad[(size_t)q * ds + i];

    // ---- gather.  Scan workgroup w kept this query's records in lanes L0 and L0+32 of wave q/32
    // ([thread-in-workgroup][workgroup] layout): a record = the lane's 16 scores of one tile, rows
    // tile*32 + 4*hh + (r&3) + 8*(r>>2).  Thread i < 2*nwg owns one lane buffer: count the scores
    // >= theta (what the old per-row append code did on the stream), scan, then write (score, row).
    // The index's zero-norm rows score 0 in the scan (stored as zeros); their exact dist is 0, so they
    // join every query's candidates here with cosine 1 -- and are dropped from the records, should a
    // threshold <= 0 have let them through.
    const int nwg = p.nwg;
    const float th = p.theta[q];
    const uint32_t nz = p.n_zero;
    // rows with a norm outside the f32 stages' range (stored as zeros, like the zero-norm rows, but their cosine is
    // not known): they never become candidates; stage 3 evaluates them in f64 for every query, beside the survivors
    const uint32_t nw = p.n_wild;
    auto in_list = [&](const uint32_t *list, uint32_t n, uint32_t row) {  // ascending lists
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (list[mid] < row) lo = mid + 1;
            else hi = mid;
        }
        return lo < n && list[lo] == row;
    };
    auto is_zero_row = [&](uint32_t row) { return (nz && in_list(p.zero_rows, nz, row)) || (nw && in_list(p.wild_rows, nw, row)); };
    // (a) record counts of this query's 2*nwg lane buffers -> exclusive scan in LDS
    uint32_t *s_off = reinterpret_cast<uint32_t *>(qv + ds) + 32;  // [2*kMaxScanWGs + 1], behind the staged-row ids
    uint32_t nrec = 0;
    if (tid < 2 * nwg) {
        const uint32_t hh0 = (uint32_t)(tid / nwg);
        const int w = tid - (int)hh0 * nwg;
        nrec = p.lane_cnt[(size_t)((uint32_t)(q >> 5) * 64 + (uint32_t)(q & 31) + 32u * hh0) * nwg + w];
        nrec = nrec > (uint32_t)kRecCap ? (uint32_t)kRecCap : nrec;
    }
    uint32_t R;
    const uint32_t roff = block_scan_1024(nrec, s_w, &R);
    if (tid <= 2 * nwg) s_off[tid] = tid < 2 * nwg ? roff : R;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // (b) one record per thread and trip (all of a query's records are in flight together): find its
    // lane buffer by binary search in the scanned counts, test the 16 scores, append the passing rows
    auto put = [&](uint32_t at, float sc, uint32_t row) {
        if (at < (uint32_t)kCandCap) {
            Cand cd;
            cd.score = sc;
            cd.row = row;
            ent[at] = cd;
        }
    };
    for (uint32_t j0 = 0; j0 < R; j0 += kFinThreads) {  // block-uniform trip count
        const uint32_t j = j0 + tid;
        uint32_t mask = 0, rowb = 0;
        float v[16];
        if (j < R) {
            uint32_t lo = 0, hi = (uint32_t)(2 * nwg) - 1;  // last lane buffer i with s_off[i] <= j
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) >> 1;
                if (s_off[mid] <= j) lo = mid;
                else hi = mid - 1;
            }
            const uint32_t e = j - s_off[lo], hh = lo / (uint32_t)nwg, w = lo - hh * (uint32_t)nwg;
            const size_t l = (size_t)((uint32_t)(q >> 5) * 64 + (uint32_t)(q & 31) + 32u * hh) * nwg + w;
            const f32x4 *rec = reinterpret_cast<const f32x4 *>(p.lane_rec + l * (kRecCap * 16)) + e * 4;
            rowb = p.lane_tile[l * kRecCap + e] * kTileRows + 4u * hh;
            const f32x4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = a[r], v[4 + r] = b[r], v[8 + r] = c[r], v[12 + r] = d[r];
            const float er = resid(rowb);                  // one half tile per record
            const float thr = fmaf(-qb, er, th), eb = fmaf(qb, er, qa);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t row = rowb + (r & 3) + 8 * (r >> 2);
                const bool pass = v[r] >= thr && (uint64_t)row < p.n_rows && !(v[r] == 0.0f && (nz | nw) && is_zero_row(row));
                mask |= pass ? (1u << r) : 0u;
                v[r] -= eb;                                // candidates carry the LOWER bound of their cosine
            }
        }
        // one LDS atomic per WAVE (a per-thread atomicAdd on the one counter serialises: ~1800 of them
        // cost 19 us against 13 this way): wave-inclusive scan of the pass counts, lane 63 reserves the range
        const uint32_t cnt = (uint32_t)__popc(mask);
        uint32_t inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        uint32_t wbase = 0;
        if (lane == 63 && inc) wbase = atomicAdd(&s_cnt, inc);
        wbase = __shfl(wbase, 63);
        uint32_t at = wbase + inc - cnt;
        if (mask) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mask & (1u << r)) put(at++, v[r], rowb + (r & 3) + 8 * (r >> 2));
        }
    }
    // the zero-norm rows
    for (uint32_t z = tid; z < nz; z += kFinThreads)
        if ((uint64_t)p.zero_rows[z] < p.n_rows) put(atomicAdd(&s_cnt, 1u), 1.0f, p.zero_rows[z]);
    __syncthreads();
    uint32_t M = s_cnt;
    const bool too_many = M > (uint32_t)kCandCap;  // more than the block can hold: keep what fits (any subset yields a
                                                    // valid rescan threshold) and flag the query for the rescan
    if (too_many) M = kCandCap;
    __syncthreads();
    if (M < (uint32_t)want) {  // cannot happen unless candidates were lost: leave it to the host
        if (tid == 0) p.overflow[q] = lane_ovf ? 3u : 2u;
        return;
    }

    // ---- stage 1: L = k-th best lower bound; keep the rows whose upper bound (lower + 2 * own bound) reaches L;
    // the survivors carry their filter score again (lower + own bound)
    uint32_t key[kFinPer], row[kFinPer];
    bool valid[kFinPer];
#pragma unroll
    for (int e = 0; e < kFinPer; ++e) {
        const uint32_t i = (uint32_t)e * kFinThreads + tid;
        valid[e] = i < M;
        if (valid[e]) {
            const Cand cd = ent[i];
            key[e] = f32_key(cd.score);
            row[e] = cd.row;
        } else {
            key[e] = 0;
            row[e] = 0xffffffffu;
        }
    }
    {
        const uint32_t kth = block_kth_largest<kFinPer>(key, valid, (uint32_t)want, s_hist, s_pick);
        const float L1 = key_f32(kth);
        if (tid == 0) s_cnt = 0;
        __syncthreads();  // also: every thread has its entries in registers, ent[] may be overwritten
        uint32_t mine = 0;
        float eb[kFinPer];
#pragma unroll
        for (int e = 0; e < kFinPer; ++e) {
            eb[e] = valid[e] ? fmaf(qb, resid(row[e]), qa) : 0.0f;
            valid[e] = valid[e] && key_f32(key[e]) + 2.0f * eb[e] >= L1;
            mine += valid[e] ? 1u : 0u;
        }
        uint32_t at = mine ? atomicAdd(&s_cnt, mine) : 0;
#pragma unroll
        for (int e = 0; e < kFinPer; ++e)
            if (valid[e]) {
                Cand cd;
                cd.score = key_f32(key[e]) + eb[e];
                cd.row = row[e];
                ent[at++] = cd;
            }
        __syncthreads();
    }
    const uint32_t m1 = s_cnt;
    if (tid == 0) p.cand_cnt[q] = m1;

    // ---- stage 2: f32 rescoring of the m1 survivors (one wave per row, 4 rows in flight per wave):
    // s2 = sum fma(q_i/|q|, c_i/|c|), |s2 - cos| <= e2 (any summation order)
    const float invq = (float)(1.0 / sqrt(na));
    const int nc4 = ds >> 2;
    float err = 0.0f;
    for (uint32_t base = (uint32_t)wave * 4; base < m1; base += kFinWaves * 4) {
        float dot[4];
        float sc[4];   // f32 store: 1/|c| of the row; compressed: sum of squares of the stored (near-unit) row
        uint32_t rr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = base + j < m1 ? base + j : base;
            rr[j] = ent[i].row;
            sc[j] = CMP ? 0.0f : p.scale[rr[j]];
            dot[j] = 0.0f;
        }
        // per pass of 768 dims: at most 3 float4 per lane and row, all 12 row loads in flight together (the rows
        // are random 1.5-3 KB reads in a multi-GB array: every one is a TLB miss); wide rows take two passes
        for (int tb = 0; tb * 64 < nc4; tb += kMaxKC / 2) {
            float4 x[4][kMaxKC / 2];
#pragma unroll
            for (int t = 0; t < kMaxKC / 2; ++t) {
                const int c4 = lane + 64 * (tb + t);
                if (c4 < nc4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j][t] = row_load4<CMP>(p.x, p.xh, ds, rr[j], c4);
                }
            }
#pragma unroll
            for (int t = 0; t < kMaxKC / 2; ++t) {
                const int c4 = lane + 64 * (tb + t);
                if (c4 < nc4) {
                    const float4 a = *reinterpret_cast<const float4 *>(qv + 4 * c4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float m = CMP ? 1.0f : sc[j];
                        dot[j] = fmaf(a.x * invq, x[j][t].x * m, dot[j]);
                        dot[j] = fmaf(a.y * invq, x[j][t].y * m, dot[j]);
                        dot[j] = fmaf(a.z * invq, x[j][t].z * m, dot[j]);
                        dot[j] = fmaf(a.w * invq, x[j][t].w * m, dot[j]);
                        if (CMP) sc[j] = fmaf(x[j][t].x, x[j][t].x, fmaf(x[j][t].y, x[j][t].y, fmaf(x[j][t].z, x[j][t].z, fmaf(x[j][t].w, x[j][t].w, sc[j]))));
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                dot[j] += __shfl_xor(dot[j], o);
                if (CMP) sc[j] += __shfl_xor(sc[j], o);
            }
        }
        if (lane < 4 && base + lane < m1) {
            const float d0 = lane == 0 ? dot[0] : lane == 1 ? dot[1] : lane == 2 ? dot[2] : dot[3];
            const float s0 = lane == 0 ? sc[0] : lane == 1 ? sc[1] : lane == 2 ? sc[2] : sc[3];
            // zero-norm row (f32 store: its 1/|c| is kept as 0; compressed: every element is 0): dist 0
            const bool zero_row = !(s0 > 0.0f);
            const float s2 = zero_row ? 1.0f : (CMP ? d0 * __frsqrt_rn(s0) : d0);
            if (p.max_err && !zero_row) err = fmaxf(err, fabsf(s2 - ent[base + lane].score));
            ent[base + lane].score = s2;
        }
    }
    if (p.max_err) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) err = fmaxf(err, __shfl_xor(err, o));
        if (lane == 0 && err > 0.0f) atomicMax(reinterpret_cast<unsigned int *>(p.max_err), __float_as_uint(err));
    }
    __syncthreads();

    // ---- k-th best f32 score L; keep [L - 2*e2, +inf); publish the retry threshold
#pragma unroll
    for (int e = 0; e < kFinPer; ++e) {
        const uint32_t i = (uint32_t)e * kFinThreads + tid;
        valid[e] = i < m1;
        if (valid[e]) {
            const Cand cd = ent[i];
            key[e] = f32_key(cd.score);
            row[e] = cd.row;
        } else {
            key[e] = 0;
            row[e] = 0xffffffffu;
        }
    }
    {
        uint32_t kth;
        if (m1 <= 64u) {
            // the usual case (a few dozen survivors): wave 0 ranks them with shuffles, no histogram passes
            if (wave == 0) {
                const uint32_t mk = key[0];  // entry `lane` (kFinThreads >= 64: entry e = 0 of threads 0..63)
                uint32_t gt = 0, ge = 0;
                for (int j = 0; j < 64; ++j) {
                    const uint32_t o = __shfl(mk, j);
                    const bool ov = (uint32_t)j < m1;
                    gt += (ov && o > mk) ? 1u : 0u;
                    ge += (ov && o >= mk) ? 1u : 0u;
                }
                if (valid[0] && gt < (uint32_t)want && (uint32_t)want <= ge) s_pick[0] = mk;  // ties write the same value
            }
            __syncthreads();
            kth = s_pick[0];
        } else {
            kth = block_kth_largest<kFinPer>(key, valid, (uint32_t)want, s_hist, s_pick);
        }
        const float L = key_f32(kth);
        if (tid == 0) {
            p.theta_retry[q] = L - p.e2 - qa - 1e-6f;  // a row of the top-k: score + qa + qb * residual >= L - e2
            s_cnt = 0;
        }
        if (lane_ovf || too_many) {  // incomplete candidate set: the host rescans this query with theta_retry
            if (tid == 0) p.overflow[q] = 1;
            return;
        }
        const uint32_t keep = f32_key(L - 2.0f * p.e2);
        __syncthreads();
        uint32_t mine = 0;
#pragma unroll
        for (int e = 0; e < kFinPer; ++e) mine += (valid[e] && key[e] >= keep) ? 1u : 0u;
        uint32_t at = mine ? atomicAdd(&s_cnt, mine) : 0;
#pragma unroll
        for (int e = 0; e < kFinPer; ++e)
            if (valid[e] && key[e] >= keep) {
                Cand cd;
                cd.score = key_f32(key[e]);
                cd.row = row[e];
                ent[at++] = cd;
            }
        __syncthreads();
    }
    const uint32_t m2 = s_cnt;

    // ---- stage 3: exact DistCosine: sequential f64 chain, one thread per row, key = (dist_bits << 32 |
    // row).  Up to kStageRows survivors (the usual case: k plus a few) are first copied to LDS by the
    // whole block -- one coalesced fetch instead of a dozen dependent ones per chain; beyond that the
    // chains read global memory (the rows are L2-warm from stage 2).
    constexpr int kStageRows = 32;
    uint32_t *srow = reinterpret_cast<uint32_t *>(qv + ds);                     // [kStageRows] row ids
    float *stage = reinterpret_cast<float *>(fsm) + 2 * kStageRows;            // behind kStageRows keys
    const int pitch = ds + 4;                                                   // +16 B: conflict-free ds_read_b128 across rows
    // rows that fit behind the keys in ent[]'s storage: 32 up to 1020 dims, 21 at 1536
    const uint32_t stage_rows = min((uint32_t)kStageRows, (uint32_t)((sizeof(Cand) * kCandCap - 2 * kStageRows * sizeof(float)) / (pitch * sizeof(float))));
    // the listed wide-norm rows join here: mt keys in all
    uint32_t nwv = 0;  // those of them below n_rows (a rolled-back append may have left later ones)
    for (uint32_t i = 0; i < nw; ++i) nwv += (uint64_t)p.wild_rows[i] < p.n_rows ? 1u : 0u;  // ascending: a prefix
    const uint32_t mt = m2 + nwv;
    if (mt > (uint32_t)kCandCap) {
        // keys[] shares ent[]'s kCandCap entries and the raw query sits right behind them: a duplicate-heavy corpus can
        // keep (nearly) kCandCap survivors, and the wide-norm rows on top would run into qv[].  Workgroup-uniform: the
        // host rescans the query with theta_retry and, if that overflows as well, answers it on the EXACT path.
        if (tid == 0) p.overflow[q] = 1;
        return;
    }
    if (mt <= stage_rows) {
        if (tid < (int)m2) srow[tid] = ent[tid].row;
        else if (tid < (int)mt) srow[tid] = p.wild_rows[tid - (int)m2];
        __syncthreads();  // row ids are out of ent[]: its storage becomes keys + staged rows
        const uint32_t nc4s = (uint32_t)ds >> 2;
        for (uint32_t i = tid; i < mt * nc4s; i += kFinThreads) {
            const uint32_t r = i / nc4s, c4 = i - r * nc4s;
            *reinterpret_cast<float4 *>(stage + (size_t)r * pitch + 4 * c4) = row_load4<CMP>(p.x, p.xh, ds, srow[r], (int)c4);
        }
        __syncthreads();
        if (tid < (int)mt) {
            const float d = exact_dist_row(qv, stage + (size_t)tid * pitch, ds, na, nullptr);
            keys[tid] = ((uint64_t)__float_as_uint(d) << 32) | srow[tid];
        }
    } else {
        // (keys[] shares ent[]'s storage, 8 bytes per entry both: thread cI reads ent[cI] and writes keys[cI] itself)
        for (uint32_t cI = tid; cI < mt; cI += kFinThreads) {
            const uint32_t r = cI < m2 ? ent[cI].row : p.wild_rows[cI - m2];
            const float d = exact_dist_stored<CMP>(qv, p.x, p.xh, ds, r, na);
            keys[cI] = ((uint64_t)__float_as_uint(d) << 32) | r;
        }
    }
    __syncthreads();

    // ---- order by (dist, row) and emit the first `want`
    uint64_t lim = ~0ull;  // keys above the want-th smallest need no rank
    if (mt > 2048u) {
        // many exact ties (duplicated rows): select the want-th smallest key first (64-bit radix,
        // MSB first), so that the quadratic ranking below runs on `want` keys only
        uint64_t prefix = 0;
        for (int bit = 63; bit >= 0; --bit) {
            const uint64_t trial = prefix | (1ull << bit);
            uint32_t cc = 0;
            for (uint32_t cI = tid; cI < mt; cI += kFinThreads) cc += keys[cI] < trial ? 1u : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cc += __shfl_xor(cc, o);
            uint32_t *slot = s_sel[bit & 1];
            if (lane == 0) slot[wave] = cc;
            __syncthreads();
            cc = 0;
#pragma unroll
            for (int w = 0; w < kFinWaves; ++w) cc += slot[w];
            if (cc < (uint32_t)want) prefix = trial;  // fewer than `want` keys below trial: the want-th is >= trial
        }
        lim = prefix;
    }
    for (uint32_t cI = tid; cI < mt; cI += kFinThreads) {
        const uint64_t me = keys[cI];
        if (me > lim) continue;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < mt; ++j) rank += keys[j] < me ? 1u : 0u;
        if (rank < (uint32_t)want) {
            const float d = __uint_as_float((uint32_t)(me >> 32));
            oid[rank] = p.idmap.id_of((uint32_t)me);
            osc[rank] = score_from_dist(d);
            if (odi) odi[rank] = d;
        }
    }
}

// One workgroup per query (finish_query above), then the completion signal of FinishParams.
template <bool CMP>
__global__ __launch_bounds__(kFinThreads) void finish_kernel(const FinishParams p) {
    finish_query<CMP>(p);  // every exit of it is workgroup-uniform
    if (!p.host_flags) return;
    __shared__ uint32_t s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        // this workgroup's results and flag words before its tick (results in mapped host memory: visible to the host, which
        // reads them as soon as it sees the sequence number)
        if (p.host_out) __threadfence_system();
        else __threadfence();
        s_last = atomicAdd(p.done_ctr, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // summary of the batch for the host (4 words + seq: one small write over PCIe; the per-query words stay
    // in HBM and are only fetched when the summary says some query overflowed)
    __shared__ uint32_t s_sum[4];
    if (threadIdx.x < 4) s_sum[threadIdx.x] = 0;
    __threadfence();
    __syncthreads();
    if ((int)threadIdx.x < p.n_queries) {
        const uint32_t ovf = __hip_atomic_load(&p.dev_flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t cnt = __hip_atomic_load(&p.dev_flags[kMaxBatch + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t bad = __hip_atomic_load(&p.dev_flags[3 * kMaxBatch + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ovf) atomicMax(&s_sum[0], ovf);
        atomicAdd(&s_sum[1], cnt);
        if (bad) atomicOr(&s_sum[2], 1u);
        // e1 is a positive float: the order of its bits is its order
        atomicMax(&s_sum[3], __hip_atomic_load(&p.dev_flags[2 * kMaxBatch + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        p.host_flags[0] = s_sum[0];  // 0: nobody overflowed; 1: some need the retry pass; >= 2: some need EXACT
        p.host_flags[1] = s_sum[1];  // candidates rescored in f32, whole batch
        p.host_flags[2] = s_sum[2];  // a query held non-finite values
        p.host_flags[3] = s_sum[3];  // largest e1 of the batch (the bound the profiling maximum is checked against)
        *p.done_ctr = 0;
        __hip_atomic_store(&p.host_flags[4], p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t finish_setup() {
    const int lds = (int)(sizeof(Cand) * (size_t)kCandCap + sizeof(float) * (size_t)kMaxKC16 * kChunkFloats + kFinishTailBytes);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&finish_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&finish_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

hipError_t launch_finish(hipStream_t s, int B, const FinishParams &p) {
    if (B <= 0) return hipSuccess;
    const size_t lds = sizeof(Cand) * (size_t)kCandCap + sizeof(float) * (size_t)p.ds + kFinishTailBytes;  // candidates | query | staged row ids | record offsets
    if (p.x) hipLaunchKernelGGL(finish_kernel<false>, dim3(B), dim3(kFinThreads), lds, s, p);
    else hipLaunchKernelGGL(finish_kernel<true>, dim3(B), dim3(kFinThreads), lds, s, p);  // compressed corpus: rows come from xh
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// EXACT path: f64 DistCosine on every row, then a 64-step MSB-first select on (dist,row) keys
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// EXACT path, batched: every row against a GROUP of up to 32 queries in one pass (k > 256, more listed rows than
// finish_kernel takes, rows wider than the scans, MX_SEARCH_EXACT, and queries whose neighbourhood overflowed twice).
// The reference answers any vector and any limit at one price (local.rs:71-91); so must this: a group costs about one
// f32 scan of the corpus plus five light passes over its 4-byte distances, whatever k is.
//   exact_dist_batch_kernel   dist[g][row] = DistCosine(query g, row): f32 products, sequential f64 sums in element
//                             order (one chain per pair), as the oracle; a workgroup = 64 rows x 32 queries, the rows and
//                             the queries of a 128-dim chunk in LDS (rows padded: conflict-free), thread = 1 row x 8 queries
//   xsel_hist / xsel_pick     3-pass radix select (12 + 12 + 8 bits) of the kk-th smallest distance per query
//   xsel_count / xsel_collect rows below it, plus the lowest-numbered rows equal to it (ties are ordered by row: slices
//                             are contiguous and ascending, a tie's rank is a prefix count)
//   xsel_emit_kernel          order the kk keys by (dist, row), emit ids / scores / dists
// ---------------------------------------------------------------------------------------------
constexpr int kXRows = 64;      // rows per workgroup pass
constexpr int kXChunk = 128;    // dims per LDS chunk
constexpr int kXPitch = kXChunk + 4;  // row pitch in floats: lanes (rows) hit distinct bank groups with 16-byte reads

// QPT = queries per thread: the workgroup's four waves take QPT queries each (4 QPT per pass over the rows); the host picks
// the smallest QPT that holds the group, so that four queries do not pay for thirty-two
template <bool CMP, int QPT>
__global__ __launch_bounds__(256) void exact_dist_batch_kernel(int ds, const float *__restrict__ x, const void *__restrict__ xh,
                                                               uint64_t n_rows, const float *__restrict__ qpad,
                                                               const double *__restrict__ qnorm2, ExactGroup grp,
                                                               uint32_t *__restrict__ dist) {
    __shared__ __attribute__((aligned(16))) float s_rows[kXRows * kXPitch];
    __shared__ __attribute__((aligned(16))) float s_q[kExactGroup * kXChunk];
    const int tid = threadIdx.x, lr = tid & 63, qg = tid >> 6;  // local row, query group of the wave (QPT queries)
    const int nq = grp.n;
    double na[QPT];
#pragma unroll
    for (int j = 0; j < QPT; ++j) na[j] = qg * QPT + j < nq ? qnorm2[grp.q[qg * QPT + j]] : 0.0;
    const uint64_t tiles = (n_rows + kXRows - 1) / kXRows;
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t r0 = t * kXRows;
        double dot[QPT], nb = 0.0;
#pragma unroll
        for (int j = 0; j < QPT; ++j) dot[j] = 0.0;
        for (int c0 = 0; c0 < ds; c0 += kXChunk) {
            __syncthreads();  // the previous chunk has been consumed
            for (int i = tid; i < kXRows * (kXChunk / 4); i += 256) {  // coalesced: 32 consecutive float4 per row
                const int r = i / (kXChunk / 4), c4 = i % (kXChunk / 4);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r0 + r < n_rows) v = row_load4<CMP>(x, xh, ds, (uint32_t)(r0 + r), c0 / 4 + c4);
                *reinterpret_cast<float4 *>(s_rows + r * kXPitch + 4 * c4) = v;
            }
            for (int i = tid; i < 4 * QPT * (kXChunk / 4); i += 256) {
                const int g = i / (kXChunk / 4), c4 = i % (kXChunk / 4);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g < nq) v = *reinterpret_cast<const float4 *>(qpad + (size_t)grp.q[g] * ds + c0 + 4 * c4);
                *reinterpret_cast<float4 *>(s_q + g * kXChunk + 4 * c4) = v;
            }
            __syncthreads();
            const float *rw = s_rows + lr * kXPitch;
            const float *qw = s_q + qg * QPT * kXChunk;
#pragma unroll 2
            for (int i = 0; i < kXChunk / 4; ++i) {
                const float4 c = *reinterpret_cast<const float4 *>(rw + 4 * i);
                nb += (double)__fmul_rn(c.x, c.x);
                nb += (double)__fmul_rn(c.y, c.y);
                nb += (double)__fmul_rn(c.z, c.z);
                nb += (double)__fmul_rn(c.w, c.w);
#pragma unroll
                for (int j = 0; j < QPT; ++j) {
                    const float4 a = *reinterpret_cast<const float4 *>(qw + j * kXChunk + 4 * i);  // same address in every lane
                    dot[j] += (double)__fmul_rn(a.x, c.x);
                    dot[j] += (double)__fmul_rn(a.y, c.y);
                    dot[j] += (double)__fmul_rn(a.z, c.z);
                    dot[j] += (double)__fmul_rn(a.w, c.w);
                }
            }
        }
        if (r0 + lr < n_rows) {
#pragma unroll
            for (int j = 0; j < QPT; ++j)
                if (qg * QPT + j < nq) dist[(size_t)(qg * QPT + j) * n_rows + r0 + lr] = __float_as_uint(dist_from_sums(dot[j], na[j], nb));
        }
    }
}

// per query g of the group: state[g] = {prefix, need, -, cursor}; hist[g][4096]
constexpr int kXBins = 4096;
__global__ void xsel_init_kernel(uint32_t kk, uint32_t *state) {
    state[4 * threadIdx.x] = 0;
    state[4 * threadIdx.x + 1] = kk;
    state[4 * threadIdx.x + 2] = 0;
    state[4 * threadIdx.x + 3] = 0;
}
__global__ __launch_bounds__(256) void xsel_hist_kernel(const uint32_t *__restrict__ dist, uint64_t n_rows, int pass,
                                                        const uint32_t *__restrict__ state, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[kXBins];
    const int g = blockIdx.y;
    for (int i = threadIdx.x; i < kXBins; i += 256) h[i] = 0;
    __syncthreads();
    const uint32_t prefix = state[4 * g];
    const int shift = pass == 0 ? 20 : pass == 1 ? 8 : 0;
    const uint32_t bmask = pass == 2 ? 0xffu : 0xfffu;
    const uint32_t pmask = pass == 0 ? 0u : pass == 1 ? 0xfff00000u : 0xffffff00u;
    const uint64_t per = (n_rows + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n_rows ? lo + per : n_rows;
    const uint32_t *d = dist + (size_t)g * n_rows;
    for (uint64_t r = lo + threadIdx.x; r < hi; r += 256) {
        const uint32_t key = d[r];
        if ((key & pmask) == prefix) atomicAdd(&h[(key >> shift) & bmask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kXBins; i += 256)
        if (h[i]) atomicAdd(&hist[(size_t)g * kXBins + i], h[i]);
}
// one workgroup per query: the bin that holds the need-th smallest remaining key; clears the histogram for the next pass
__global__ __launch_bounds__(256) void xsel_pick_kernel(int pass, uint32_t *__restrict__ state, uint32_t *__restrict__ hist) {
    __shared__ uint32_t s_part[256];
    __shared__ uint32_t s_res[2];
    const int g = blockIdx.x, tid = threadIdx.x;
    uint32_t *h = hist + (size_t)g * kXBins;
    const int nb = pass == 2 ? 256 : kXBins, per = nb / 256;  // bins per thread, ascending
    uint32_t mine = 0;
    for (int i = 0; i < per; ++i) mine += h[tid * per + i];
    s_part[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        const uint32_t need = state[4 * g + 1];
        uint32_t below = 0;
        int t = 0;
        for (; t < 255 && below + s_part[t] < need; ++t) below += s_part[t];
        int bin = t * per;
        for (; bin < t * per + per - 1 && below + h[bin] < need; ++bin) below += h[bin];
        const int shift = pass == 0 ? 20 : pass == 1 ? 8 : 0;
        s_res[0] = state[4 * g] | ((uint32_t)bin << shift);
        s_res[1] = need - below;
    }
    __syncthreads();
    for (int i = tid; i < kXBins; i += 256) h[i] = 0;
    if (tid == 0) {
        state[4 * g] = s_res[0];           // pass 2: the kk-th smallest key T itself
        state[4 * g + 1] = s_res[1];       // pass 2: how many rows equal to T are wanted
    }
}
// rows equal to T per slice (slices = contiguous ascending row ranges)
__global__ __launch_bounds__(256) void xsel_count_kernel(const uint32_t *__restrict__ dist, uint64_t n_rows,
                                                         const uint32_t *__restrict__ state, uint32_t *__restrict__ eq_cnt) {
    __shared__ uint32_t s4[4];
    const int g = blockIdx.y;
    const uint32_t T = state[4 * g];
    const uint64_t per = (n_rows + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n_rows ? lo + per : n_rows;
    const uint32_t *d = dist + (size_t)g * n_rows;
    uint32_t c = 0;
    for (uint64_t r = lo + threadIdx.x; r < hi; r += 256) c += d[r] == T ? 1u : 0u;
    c = block_sum_256(c, s4);
    if (threadIdx.x == 0) eq_cnt[(size_t)g * gridDim.x + blockIdx.x] = c;
}
// sel[g][kk]: keys (dist << 32 | row) of the rows below T (any order) and of the first `need` rows equal to T
__global__ __launch_bounds__(256) void xsel_collect_kernel(const uint32_t *__restrict__ dist, uint64_t n_rows, uint32_t kk,
                                                           uint32_t *__restrict__ state, const uint32_t *__restrict__ eq_cnt,
                                                           uint64_t *__restrict__ sel) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_base;
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t T = state[4 * g], need = state[4 * g + 1], n_lt = kk - need;
    const uint64_t per = (n_rows + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n_rows ? lo + per : n_rows;
    const uint32_t *d = dist + (size_t)g * n_rows;
    uint64_t *out = sel + (size_t)g * kk;
    if (tid == 0) {
        uint32_t b = 0;
        for (unsigned i = 0; i < blockIdx.x; ++i) b += eq_cnt[(size_t)g * gridDim.x + i];
        s_base = b;
    }
    __syncthreads();
    uint32_t eq_base = s_base;  // ties in lower-numbered rows
    for (uint64_t r0 = lo; r0 < hi; r0 += 256) {  // block-uniform trip count
        const uint64_t r = r0 + tid;
        const uint32_t key = r < hi ? d[r] : 0xffffffffu;
        const bool lt = r < hi && key < T, eq = r < hi && key == T;
        if (lt) {
            const uint32_t at = atomicAdd(&state[4 * g + 3], 1u);
            if (at < n_lt) out[at] = ((uint64_t)key << 32) | (uint32_t)r;
        }
        // rank of a tie among the ties of this slice, in row order
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(eq);
        const uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull)), wtot = (uint32_t)__popcll(bal);
        __syncthreads();
        if (lane == 0) s_w[wave] = wtot;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            wbase += w < wave ? s_w[w] : 0u;
            tot += s_w[w];
        }
        if (eq) {
            const uint32_t rank = eq_base + wbase + before;
            if (rank < need) out[n_lt + rank] = ((uint64_t)key << 32) | (uint32_t)r;
        }
        eq_base += tot;
    }
}
__global__ __launch_bounds__(256) void xsel_emit_kernel(int k, uint32_t kk, IdMap idmap, ExactGroup grp,
                                                        const uint64_t *__restrict__ sel_all, uint64_t *ids_all,
                                                        float *scores_all, float *dists_all, int32_t *n_found) {
    const int g = blockIdx.x, tid = threadIdx.x;
    const int q = grp.q[g];
    const uint64_t *sel = sel_all + (size_t)g * kk;
    uint64_t *ids = ids_all + (size_t)q * k;
    float *scores = scores_all + (size_t)q * k;
    float *dists = dists_all ? dists_all + (size_t)q * k : nullptr;
    if (tid == 0) n_found[q] = (int32_t)kk;
    for (int j = tid; j < k; j += 256) {
        ids[j] = 0;
        scores[j] = 0.0f;
        if (dists) dists[j] = INFINITY;
    }
    __syncthreads();
    for (uint32_t c = tid; c < kk; c += 256) {
        const uint64_t me = sel[c];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < kk; ++j) rank += sel[j] < me ? 1u : 0u;
        const float d = __uint_as_float((uint32_t)(me >> 32));
        ids[rank] = idmap.id_of((uint32_t)me);
        scores[rank] = score_from_dist(d);
        if (dists) dists[rank] = d;
    }
}

// scratch layout: hist | state | eq_cnt | sel (the fixed part, sized for a full group) | dist [gcap][n_rows]
static size_t exact_fixed_bytes(uint64_t kk) {
    size_t b = (size_t)kExactGroup * kXBins * sizeof(uint32_t)               // hist
               + (size_t)kExactGroup * 4 * sizeof(uint32_t)                  // state
               + (size_t)kExactGroup * kExactSlices * sizeof(uint32_t);      // eq_cnt
    b = (b + 15) & ~(size_t)15;
    b += (size_t)kExactGroup * (kk ? kk : 1) * sizeof(uint64_t);             // sel
    return (b + 255) & ~(size_t)255;
}

size_t exact_group_scratch_bytes(uint64_t n_rows, int k, int gcap) {
    const uint64_t kk = n_rows < (uint64_t)k ? n_rows : (uint64_t)k;
    return exact_fixed_bytes(kk) + (size_t)gcap * n_rows * sizeof(uint32_t) + 64;
}

hipError_t launch_exact_group(hipStream_t s, int k, int ds, const float *x, const void *xh, uint64_t n_rows, const IdMap &idmap,
                              const float *qpad, const double *qnorm2, const ExactGroup &grp, void *scratch, uint64_t *ids,
                              float *scores, float *dists, int32_t *n_found) {
    if (grp.n <= 0) return hipSuccess;
    const uint32_t kk = (uint32_t)(n_rows < (uint64_t)k ? n_rows : (uint64_t)k);
    char *base = static_cast<char *>(scratch);
    uint32_t *hist = reinterpret_cast<uint32_t *>(base);
    uint32_t *state = hist + (size_t)kExactGroup * kXBins;
    uint32_t *eq_cnt = state + kExactGroup * 4;
    uint64_t *sel = reinterpret_cast<uint64_t *>(((uintptr_t)(eq_cnt + (size_t)kExactGroup * kExactSlices) + 15) & ~(uintptr_t)15);
    uint32_t *dist = reinterpret_cast<uint32_t *>(base + exact_fixed_bytes(kk));  // [grp.n][n_rows]: the caller sized it for >= grp.n queries
    if (kk > 0) {
        hipError_t e = hipMemsetAsync(hist, 0, ((size_t)kExactGroup * kXBins + kExactGroup * 4) * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(xsel_init_kernel, dim3(1), dim3(kExactGroup), 0, s, kk, state);
        const uint64_t tiles = (n_rows + kXRows - 1) / kXRows;
        const unsigned blocks = (unsigned)(tiles < 2048 ? tiles : 2048);
#define MX_XDIST(CMP_, QPT_) hipLaunchKernelGGL((exact_dist_batch_kernel<CMP_, QPT_>), dim3(blocks), dim3(256), 0, s, ds, x, xh, n_rows, qpad, qnorm2, grp, dist)
        if (x) {
            if (grp.n <= 4) MX_XDIST(false, 1);
            else if (grp.n <= 8) MX_XDIST(false, 2);
            else if (grp.n <= 16) MX_XDIST(false, 4);
            else MX_XDIST(false, 8);
        } else {
            if (grp.n <= 4) MX_XDIST(true, 1);
            else if (grp.n <= 8) MX_XDIST(true, 2);
            else if (grp.n <= 16) MX_XDIST(true, 4);
            else MX_XDIST(true, 8);
        }
#undef MX_XDIST
        const unsigned slices = (unsigned)((n_rows + 4095) / 4096 < (uint64_t)kExactSlices ? (n_rows + 4095) / 4096 : (uint64_t)kExactSlices);
        const dim3 sg(slices ? slices : 1, grp.n);
        for (int pass = 0; pass < 3; ++pass) {
            hipLaunchKernelGGL(xsel_hist_kernel, sg, dim3(256), 0, s, dist, n_rows, pass, state, hist);
            hipLaunchKernelGGL(xsel_pick_kernel, dim3(grp.n), dim3(256), 0, s, pass, state, hist);
        }
        hipLaunchKernelGGL(xsel_count_kernel, sg, dim3(256), 0, s, dist, n_rows, state, eq_cnt);
        hipLaunchKernelGGL(xsel_collect_kernel, sg, dim3(256), 0, s, dist, n_rows, kk, state, eq_cnt, sel);
    }
    hipLaunchKernelGGL(xsel_emit_kernel, dim3(grp.n), dim3(256), 0, s, k, kk, idmap, grp, sel, ids, scores, dists, n_found);
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

// compressed corpus -> rows [row0, row0 + n) as f32 [n, d] (exact widening; zero-norm rows come out as zeros)
__global__ __launch_bounds__(256) void unshadow_kernel(const void *__restrict__ xh, int ds, int d, uint64_t row0, uint64_t n,
                                                       float *__restrict__ out) {
    const int nc4 = (d + 3) / 4;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n * (uint64_t)nc4; i += (uint64_t)gridDim.x * 256) {
        const uint64_t r = i / nc4;
        const int c4 = (int)(i - r * nc4);
        const float4 v = row_load4<true>(nullptr, xh, ds, (uint32_t)(row0 + r), c4);
        float *o = out + r * (uint64_t)d + 4 * c4;
        if (4 * c4 + 0 < d) o[0] = v.x;
        if (4 * c4 + 1 < d) o[1] = v.y;
        if (4 * c4 + 2 < d) o[2] = v.z;
        if (4 * c4 + 3 < d) o[3] = v.w;
    }
}

hipError_t launch_unshadow(hipStream_t s, const void *xh, int ds, int d, uint64_t row0, uint64_t n, float *out) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n * (uint64_t)((d + 3) / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(unshadow_kernel, dim3((unsigned)blocks), dim3(256), 0, s, xh, ds, d, row0, n, out);
    return hipGetLastError();
}

__global__ void fill_nfound_kernel(int32_t *nf, int B, int32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) nf[i] = v;
}

hipError_t launch_fill_nfound(hipStream_t s, int32_t *nf, int B, int32_t v) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_nfound_kernel, dim3((B + 255) / 256), dim3(256), 0, s, nf, B, v);
    return hipGetLastError();
}

hipError_t launch_merge(hipStream_t s, const void *ids, size_t ids_stride, const void *dists, size_t dists_stride,
                        int G, int B, int k, uint64_t *out_ids, float *out_dists, float *out_scores) {
    if (B <= 0 || k <= 0) return hipSuccess;
    hipLaunchKernelGGL(merge_kernel, dim3(B), dim3(256), 0, s, static_cast<const char *>(ids), ids_stride,
                       static_cast<const char *>(dists), dists_stride, G, B, k, out_ids, out_dists, out_scores);
    return hipGetLastError();
}

}  // namespace mx
