// mx_gelu.h -- the erf GELU of the BERT intermediate layer (reference: rust-bert's `gelu`, oracle/bert_oracle.py), written for
// the VALU budget of the MFMA epilogues it runs in (a hidden-768 W1 launch spent 95 of 655 us in it, the fused hidden-384
// tail 56 of 395: no MFMA shadow hides an epilogue):
//     gelu(x) = x Phi(x),   Phi(x) - 1/2 = erf(x / sqrt 2) / 2 = xc P(xc^2),   xc = x clamped to [-4.25, 4.25]
// P: degree 8, minimax for |Phi error| over the clamp range (8.5e-6; 1.2e-5 with f32 Horner rounding), beyond it Phi is 0 or 1
// to 1.1e-5 (the constant term is nudged by two ulps so that Phi(-4.25) = 3e-8 and Phi(4.25) = 1 exactly in f32: beyond the
// clamp the error grows by 3e-8 |x|).  |gelu - exact| <= 5e-5 + 1e-7 |x| (largest at the clamp points; 1.3e-5 |x| inside),
// relative error <= 2.4e-5 for x >= 0 -- every caller rounds the result to bf16 (2^-9 relative) next; the embeddings of the f64 oracle move by
// 1 - cos <= 1.3e-10 when its exact erf is swapped for this polynomial (three model shapes).
// No quarter-rate instruction (the Abramowitz-Stegun 7.1.28 form this replaces paid one v_rcp_f32, 4 issue slots, per value),
// everything but the clamp on PAIRS (v_pk_mul_f32 / v_pk_fma_f32): 6.5 issue slots per value against 13.
// All users (gemm_kernel's and pgemm_kernel's GELU epilogues, tail_kernel) call this one function with explicit fma's, so
// the fused and the GEMM-by-GEMM paths stay bit-identical.
#pragma once
#include <hip/hip_runtime.h>

namespace mx {

typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gelu_f32x2 gelu_erf2(gelu_f32x2 x) {
    typedef gelu_f32x2 v2;
    const v2 xc = {__builtin_amdgcn_fmed3f(x[0], -4.25f, 4.25f), __builtin_amdgcn_fmed3f(x[1], -4.25f, 4.25f)};
    const v2 t = xc * xc;
    v2 P = __builtin_elementwise_fma(t, (v2)5.564912767e-11f, (v2)-5.327799091e-09f);
    P = __builtin_elementwise_fma(P, t, (v2)2.255440705e-07f);
    P = __builtin_elementwise_fma(P, t, (v2)-5.626448910e-06f);
    P = __builtin_elementwise_fma(P, t, (v2)9.341890109e-05f);
    P = __builtin_elementwise_fma(P, t, (v2)-1.108561992e-03f);
    P = __builtin_elementwise_fma(P, t, (v2)9.815974161e-03f);
    P = __builtin_elementwise_fma(P, t, (v2)-6.634449214e-02f);
    P = __builtin_elementwise_fma(P, t, (v2)3.989023864e-01f);
    const v2 phi = __builtin_elementwise_fma(xc, P, (v2)0.5f);
    return x * phi;
}

}  // namespace mx
