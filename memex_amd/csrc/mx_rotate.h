// mx_rotate.h -- the fixed orthonormal rotation applied to unit vectors before they are quantised to int8
// (scan8.hip's filter copy and prep_queries_kernel's query fragments; nothing else sees rotated values).
//
// Why: one quantisation step per 32 rows (and per query) is max |element| / 127, and the residual of a row is
// ~ step * sqrt(dim / 12).  On i.i.d. Gaussian rows the largest element is ~4 / sqrt(dim) and the residual
// 0.009; sentence embeddings are not like that -- a decaying spectrum or a common mean direction puts most of a
// unit vector's energy into a few dimensions, the step follows the largest element, and the measured residual
// grows to 0.06-0.09 (scripts/gpu_realistic_rows.py: the int8 certificate then keeps thousands of rows per
// query).  A rotation T (orthonormal) leaves every dot product as it is, (Tq).(Tc) = q.c, and a pseudo-random one
// spreads any vector's energy over all dimensions, so that what is quantised always looks like the Gaussian
// case.  T = (M_m (x) I_128) * (I_m (x) H_128 / sqrt(128)) * D  for dim_pad = 128 m:
//   D      diagonal of fixed pseudo-random signs (a vector that happens to be a Hadamard row stays spread),
//   H_128  Walsh-Hadamard transform inside every block of 128 dims (7 butterfly stages: 6 wave shuffles + 1),
//   M_m    the orthonormal DCT-II matrix across the m blocks, position by position (m = 1: nothing to mix).
// f32 arithmetic: |T x| = |x| and (Tq).(Tc) = q.c to ~1e-6, inside the certificate's 2.7e-4 slack.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mx {

constexpr int kRotMaxBlocks = 12;  // dim_pad <= 1536

__device__ __forceinline__ float rot_sign(uint32_t j) {  // +-1, fixed per dimension
    uint32_t h = j * 2654435761u + 0x9e3779b9u;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    return (h & 1u) ? -1.0f : 1.0f;
}

// mix[k * m + j] = M_m[k][j], the orthonormal DCT-II matrix; written by the first m*m threads of a group
__device__ __forceinline__ void rot_fill_mix(float *mix, int m, int t, int nthreads) {
    for (int i = t; i < m * m; i += nthreads) {
        const int k = i / m, j = i - k * m;
        mix[i] = sqrtf((k ? 2.0f : 1.0f) / (float)m) * cospif((float)((2 * j + 1) * k) / (float)(2 * m));
    }
}

// One wave rotates one vector: in[0 .. 128 m) (LDS, owned by this wave, overwritten) -> out[0 .. 128 m) (LDS).
// Lane l touches dims 128 b + l and 128 b + 64 + l of every block only, in both phases: no barrier needed.
__device__ __forceinline__ void rot_wave(float *in, float *out, int m, int lane, const float *mix) {
    for (int b = 0; b < m; ++b) {
        const uint32_t j0 = 128u * (uint32_t)b + (uint32_t)lane, j1 = j0 + 64u;
        float a0 = in[j0] * rot_sign(j0), a1 = in[j1] * rot_sign(j1);
        const float t = a0 + a1;  // stride 64
        a1 = a0 - a1;
        a0 = t;
#pragma unroll
        for (int bit = 1; bit < 64; bit <<= 1) {
            const float o0 = __shfl_xor(a0, bit), o1 = __shfl_xor(a1, bit);
            a0 = (lane & bit) ? o0 - a0 : a0 + o0;
            a1 = (lane & bit) ? o1 - a1 : a1 + o1;
        }
        in[j0] = a0 * 0.08838834764831845f;  // 1 / sqrt(128)
        in[j1] = a1 * 0.08838834764831845f;
    }
    if (m == 1) {
        out[lane] = in[lane];
        out[64 + lane] = in[64 + lane];
        return;
    }
    for (int hh = 0; hh < 2; ++hh) {
        const int p = 64 * hh + lane;
        for (int k = 0; k < m; ++k) {
            float acc = 0.0f;
            for (int j = 0; j < m; ++j) acc = fmaf(mix[k * m + j], in[128 * j + p], acc);
            out[128 * k + p] = acc;
        }
    }
}

}  // namespace mx
