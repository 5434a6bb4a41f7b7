// Host-side check of the fused layer-tail kernel's weight stream (memex_amd/csrc/encoder_tail.hip,
// tail_stream_layout): every element of Wo, W1 and W2 lands in the stream exactly once, and the fragments
// sit where the kernel's addressing expects them (wave stream base, segment order, lane-major fragments).
// No GPU needed: the layout function is plain host code inside libmemex_hip.so.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace mx {
size_t tail_stream_elems(int F);
void tail_stream_layout(const float *wo, const float *w1, const float *w2, int F, uint16_t *out, uint16_t (*to_bf16)(float));
}  // namespace mx

// the "weights" are indices: wo in [0, 2^17), w1 from 2^20, w2 from 2^21 -- to_id keeps 16 bits + a tag lookup
static std::vector<uint32_t> g_tags;
static uint16_t to_id(float f) {
    g_tags.push_back((uint32_t)f);
    return (uint16_t)((uint32_t)f & 0xffffu);
}

int main() {
    const int H = 384;
    for (int F : {256, 384, 1536}) {
        std::vector<float> wo((size_t)H * H), w1((size_t)F * H), w2((size_t)H * F);
        // values < 2^24 are exact in f32
        for (size_t i = 0; i < wo.size(); ++i) wo[i] = (float)i;
        for (size_t i = 0; i < w1.size(); ++i) w1[i] = (float)((1u << 20) + i);
        for (size_t i = 0; i < w2.size(); ++i) w2[i] = (float)((1u << 22) + i);
        const size_t n = mx::tail_stream_elems(F);
        if (n != wo.size() + w1.size() + w2.size()) { printf("FAIL size F=%d: %zu\n", F, n); return 1; }
        std::vector<uint16_t> out(n);
        g_tags.clear();
        mx::tail_stream_layout(wo.data(), w1.data(), w2.data(), F, out.data(), to_id);
        if (g_tags.size() != n) { printf("FAIL count F=%d\n", F); return 1; }
        // (1) permutation: every source element exactly once
        std::vector<uint8_t> seen_o(wo.size(), 0), seen_1(w1.size(), 0), seen_2(w2.size(), 0);
        for (uint32_t t : g_tags) {
            uint8_t *s = t >= (1u << 22) ? &seen_2[t - (1u << 22)] : t >= (1u << 20) ? &seen_1[t - (1u << 20)] : &seen_o[t];
            if (*s) { printf("FAIL duplicate F=%d tag=%u\n", F, t); return 1; }
            *s = 1;
        }
        // (2) addressing as the kernel does it: wave wn, stream fragment index fi, lane, element e
        const int nch = F / 128, frags_per_wave = 72 + 48 * nch;
        auto at = [&](int wn, int fi, int lane, int e) { return g_tags[(((size_t)wn * frags_per_wave + fi) * 64 + lane) * 8 + e]; };
        for (int wn = 0; wn < 4; ++wn)
            for (int lane : {0, 31, 32, 63}) {
                const int r = lane & 31, h = lane >> 5;
                // out-projection fragment 3 t + j: Wo row wn*96 + j*32 + r, k = 16 t + 8 h + e
                for (int t : {0, 23}) for (int j = 0; j < 3; ++j)
                    if (at(wn, 3 * t + j, lane, 5) != (uint32_t)((wn * 96 + j * 32 + r) * H + 16 * t + 8 * h + 5)) { printf("FAIL PO\n"); return 1; }
                // first G1 segment (chunk 0) right behind: W1 row wn*32 + r, k = 16 t + 8 h + e
                if (at(wn, 72 + 7, lane, 2) != (1u << 20) + (uint32_t)((wn * 32 + r) * H + 16 * 7 + 8 * h + 2)) { printf("FAIL G1(0)\n"); return 1; }
                // second segment is G1(1), third is G2(0): fragment 3 t2 + j: W2 row wn*96 + j*32 + r, k = 16 t2 + 8 h + e
                if (at(wn, 72 + 24 + 3, lane, 0) != (1u << 20) + (uint32_t)((128 + wn * 32 + r) * H + 16 * 3 + 8 * h)) { printf("FAIL G1(1)\n"); return 1; }
                if (at(wn, 72 + 48 + 3 * 5 + 2, lane, 7) != (1u << 22) + (uint32_t)((wn * 96 + 2 * 32 + r) * F + 16 * 5 + 8 * h + 7)) { printf("FAIL G2(0)\n"); return 1; }
                // the last segment is G2(nch - 1)
                if (at(wn, frags_per_wave - 24 + 3 * 1 + 1, lane, 1) != (1u << 22) + (uint32_t)((wn * 96 + 32 + r) * F + (nch - 1) * 128 + 16 + 8 * h + 1)) { printf("FAIL G2(last)\n"); return 1; }
            }
    }
    printf("OK tail stream layout\n");
    return 0;
}
