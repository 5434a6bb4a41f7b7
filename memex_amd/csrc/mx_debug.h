// mx_debug.h -- the ONE kernel-variant switch of libmemex_hip.so: MEMEX_HIP_DEBUG="key=value,key=value".
// Several kernels of the library exist in two forms that must agree (bit for bit, or within a stated bound): the fused layer
// tail and the three GEMMs it replaces, pgemm_kernel and gemm_kernel, the short-sequence attention and the staged one, the
// small-pass layer and the bulk kernels.  The parity tests (tests/test_encoder_gpu.py) run the same input through both by
// setting a key here; nothing else reads the environment for kernel selection.  The string is looked at on every use (a
// getenv and a strcmp against the cached copy), so a test can switch inside one process.  Keys (default in brackets):
//   unfused_tail=1   [0]  hidden-384 layers run out-projection / W1 / W2 as three GEMMs instead of tail_kernel
//   pgemm=0          [1]  large passes stay on gemm_kernel
//   small=0          [1]  one kernel set at every pass size (no small-pass layer, no split-k)
//   small_rows=N          where the hidden-384 small-pass layer hands over to the bulk kernels (multiple of 64)
//   splitk=0         [1]  hidden-768 small passes keep the fused Add & LayerNorm GEMMs
//   attn_f32=1       [0]  MX_PREC_BF16X3 attention on the f32 MFMA instead of three bf16 products
//   attn_short=0     [1]  sequences of <= 128 tokens stay on the staged attention_kernel
//   attn_short_lds=0 [1]  attention_short_kernel fetches K / V^T per wave from global memory
//   attn_pair=0|1    [per pass]  never / always two heads per item of the staged attention
//   attn_safe=1      [0]  running-maximum softmax loop only
//   sample_div=N          the search's sample pass visits 1/N of the scan tiles (a tuning knob: results do not depend on it)
// Unknown keys are ignored.  Operational switches (wait mode, exchange, filter copy) are separate, documented in
// include/memex_hip.h; there is no fault-injection switch in this library (libmemex_hip_testing.so, -DMEMEX_TESTING, has two).
#pragma once
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace mx {

inline int debug_flag(const char *key, int dflt) {
    static std::mutex mu;
    static std::string cached;
    static std::vector<std::pair<std::string, int>> kv;
    const char *ev = getenv("MEMEX_HIP_DEBUG");
    if (!ev) ev = "";
    std::lock_guard<std::mutex> lock(mu);
    if (cached != ev) {
        cached = ev;
        kv.clear();
        size_t i = 0;
        while (i < cached.size()) {
            size_t j = cached.find(',', i);
            if (j == std::string::npos) j = cached.size();
            const std::string item = cached.substr(i, j - i);
            const size_t eq = item.find('=');
            if (eq != std::string::npos && eq > 0) kv.emplace_back(item.substr(0, eq), atoi(item.c_str() + eq + 1));
            i = j + 1;
        }
    }
    for (const auto &p : kv)
        if (p.first == key) return p.second;
    return dflt;
}

}  // namespace mx
