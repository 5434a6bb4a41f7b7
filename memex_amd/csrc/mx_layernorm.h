// mx_layernorm.h -- the row statistics and the affine step of the Add & LayerNorm epilogues (BERT
// self-output / output blocks; oracle/bert_oracle.py), shared by gemm_kernel's EPI_BIAS_RES_LN epilogue
// and tail_kernel so that the fused and the GEMM-by-GEMM paths stay bit-identical: contraction is switched
// off inside and the two intended fma's are spelled out, otherwise the compiler fuses differently in
// different surroundings.  A row of N = TPR * NV values is held by TPR adjacent lanes, NV values each.
#pragma once
#include <hip/hip_runtime.h>

namespace mx {

template <int TPR, int NV>
__device__ __forceinline__ void ln_row_stats(const float (&y)[NV], float eps, float &mean, float &rstd) {
#pragma clang fp contract(off)
    constexpr float kN = (float)(TPR * NV);
    float sum = 0.0f;
#pragma unroll
    for (int e = 0; e < NV; ++e) sum += y[e];
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o);
    mean = sum / kN;
    float sq = 0.0f;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const float dlt = y[e] - mean;
        sq = __builtin_fmaf(dlt, dlt, sq);
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) sq += __shfl_xor(sq, o);
    rstd = 1.0f / sqrtf(sq / kN + eps);
}

__device__ __forceinline__ float ln_affine(float y, float mean, float rstd, float g, float b) {
#pragma clang fp contract(off)
    return __builtin_fmaf((y - mean) * rstd, g, b);
}

}  // namespace mx
