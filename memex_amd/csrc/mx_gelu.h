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

// The same function to f32 accuracy, for MX_PREC_BF16X3 (encoder_precise.hip keeps 16 significant bits through every product:
// the 1.2e-5 of the polynomial above would be its largest error).  erf in two pieces, both on PAIRS:
//   |a| < 1:   erf a = a P5(a^2)                                      (|err| <= 1.1e-7)
//   |a| >= 1:  erfc |a| = exp2(|a| Q6(|a|)),  |a| clamped to 4.2       (relative 4e-6 of erfc, i.e. <= 1.1e-7 of erf)
// with a = x / sqrt 2, and Phi = 1/2 + erf(a)/2 resp. erfc(|a|)/2 or 1 - erfc(|a|)/2 -- the negative tail keeps its relative accuracy.
// Coefficients: Chebyshev fits in f64, evaluated in f32 Horner against math.erf over [-9, 9] (scripts/gelu_precise_fit.py):
// |gelu - exact| <= 3.9e-7 (at x = 4.9: one ulp), relative <= 4.1e-6 wherever |gelu| > 1e-6.  ocml's erff, which this replaces,
// cost the W1 GEMM of the bf16x3 mode more than its 3 x MFMA loop; this one: 13 packed fma/mul, 2 v_exp_f32, a select.
__device__ __forceinline__ gelu_f32x2 gelu_erf2_precise(gelu_f32x2 x) {
#pragma clang fp contract(off)  // (1 - 0.5 e must not become an fma in one GEMM kernel and stay two operations in the other)
    typedef gelu_f32x2 v2;
    const v2 a = x * (v2)0.70710678118654752f;
    const v2 t = {__builtin_fminf(__builtin_fabsf(a[0]), 4.2f), __builtin_fminf(__builtin_fabsf(a[1]), 4.2f)};
    const v2 s = a * a;
    v2 p = __builtin_elementwise_fma(s, (v2)-0.0005654105916619301f, (v2)0.004923277534544468f);
    p = __builtin_elementwise_fma(p, s, (v2)-0.026716385036706924f);
    p = __builtin_elementwise_fma(p, s, (v2)0.11280364543199539f);
    p = __builtin_elementwise_fma(p, s, (v2)-0.37612348794937134f);
    p = __builtin_elementwise_fma(p, s, (v2)1.1283791065216064f);
    const v2 phi_small = __builtin_elementwise_fma(a * p, (v2)0.5f, (v2)0.5f);
    v2 q = __builtin_elementwise_fma(t, (v2)-2.20783258555457e-05f, (v2)0.0005113601218909025f);
    q = __builtin_elementwise_fma(q, t, (v2)-0.005364956334233284f);
    q = __builtin_elementwise_fma(q, t, (v2)0.03429649397730827f);
    q = __builtin_elementwise_fma(q, t, (v2)-0.15295468270778656f);
    q = __builtin_elementwise_fma(q, t, (v2)-0.9167695045471191f);
    q = __builtin_elementwise_fma(q, t, (v2)-1.62811279296875f);
    const v2 r = t * q;
    const v2 he = {0.5f * __builtin_amdgcn_exp2f(r[0]), 0.5f * __builtin_amdgcn_exp2f(r[1])};  // erfc(|a|) / 2
    v2 phi;
#pragma unroll
    for (int i = 0; i < 2; ++i) phi[i] = t[i] < 1.0f ? phi_small[i] : (a[i] < 0.0f ? he[i] : 1.0f - he[i]);
    return x * phi;
}

}  // namespace mx
