// scripts/encoder_tail2.h -- interface of the EXPERIMENTAL activation-stationary layer tail (scripts/encoder_tail2.hip;
// measured by scripts/tail_ubench.hip).  Not part of libmemex_hip.so: in three in-situ A/B runs it landed within +-3 % of
// tail_kernel with the sign depending on the box, so the product keeps one tail kernel.
#pragma once
#include "encoder_kernels.h"

namespace mx {

struct Tail2Params : TailParams {
    const bf16_t *wf2;  // Wo, W1, W2 as ONE fragment stream (tail2_stream_layout)
    const float *pf;    // bo g1 be1 b2 g2 be2 | b1 in one block (tail2_param_layout)
};
hipError_t tail2_setup();
bool tail2_supported(int hidden, int ffn);
hipError_t launch_tail2(hipStream_t s, const Tail2Params &p);
size_t tail2_stream_elems(int F);
void tail2_stream_layout(const float *wo, const float *w1, const float *w2, int F, uint16_t *out, uint16_t (*to_bf16)(float));
size_t tail2_param_floats();
void tail2_param_layout(const float *bo, const float *g1, const float *be1, const float *b1, const float *b2, const float *g2,
                        const float *be2, int F, float *out);

}  // namespace mx
