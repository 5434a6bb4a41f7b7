// mx_gelu.h -- the exact-erf GELU of the BERT intermediate layer (reference: rust-bert's `gelu`,
// oracle/bert_oracle.py), written for the VALU budget of the MFMA epilogues it runs in:
//     gelu(x) = x Phi(x) = 0.5 (x + |x|) - 0.5 |x| q,     q = 1 - erf(|x| / sqrt 2)
//     erf(z)  = 1 - (1 / (1 + a1 z + ... + a6 z^6))^16   (Abramowitz-Stegun 7.1.28, |err| <= 3e-7)
// One v_rcp_f32 per value and no v_exp_f32 (both quarter rate), no sign select (erf is odd, so
// x erf(x/sqrt 2) = |x| erf(|x|/sqrt 2)), and every other operation on PAIRS (v_pk_fma_f32 / v_pk_mul_f32):
// 13 issue slots per value against 20 for the 7.1.26 form (rcp + exp + select) it replaces.  |gelu - exact|
// <= 5e-7 + f32 rounding; the result is rounded to bf16 (2^-9 relative) by every caller.  For negative x the
// first term is exactly 0 and the result is -0.5 |x| q with q computed without cancellation.
// Both users (gemm_kernel's GELU epilogue and tail_kernel) call this one function with explicit fma's, so
// the fused and the GEMM-by-GEMM paths stay bit-identical.
#pragma once
#include <hip/hip_runtime.h>

namespace mx {

typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ gelu_f32x2 gelu_erf2(gelu_f32x2 x) {
    typedef gelu_f32x2 v2;
    const v2 u = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
    const v2 z = u * 0.70710678118654752f;
    v2 P = __builtin_elementwise_fma(z, (v2)0.0000430638f, (v2)0.0002765672f);
    P = __builtin_elementwise_fma(P, z, (v2)0.0001520143f);
    P = __builtin_elementwise_fma(P, z, (v2)0.0092705272f);
    P = __builtin_elementwise_fma(P, z, (v2)0.0422820123f);
    P = __builtin_elementwise_fma(P, z, (v2)0.0705230784f);
    P = __builtin_elementwise_fma(P, z, (v2)1.0f);
    v2 r = {__builtin_amdgcn_rcpf(P[0]), __builtin_amdgcn_rcpf(P[1])};
    r = r * r;
    r = r * r;
    r = r * r;
    r = r * r;
    const v2 hu = u * 0.5f;
    const v2 s = __builtin_elementwise_fma(x, (v2)0.5f, hu);
    return __builtin_elementwise_fma(-hu, r, s);
}

}  // namespace mx
