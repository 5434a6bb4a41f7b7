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
size_t tail2_stream_elems(int F);
void tail2_stream_layout(const float *wo, const float *w1, const float *w2, int F, uint16_t *out, uint16_t (*to_bf16)(float));
size_t tail2_param_floats();
void tail2_param_layout(const float *bo, const float *g1, const float *be1, const float *b1, const float *b2, const float *g2,
                        const float *be2, int F, float *out);
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
    // ---- tail2_kernel (encoder_tail2.hip): ONE stream, Wo | G1(0) | G1(1) G2(0) | ... | G1(nch-1) G2(nch-2) | G2(nch-1),
    // 64-feature ffn chunks, fragment = 32 weight rows x 16 k, lane (h, r) -> row r, k 8 h .. +7
    for (int F : {128, 384, 1536}) {
        std::vector<float> wo((size_t)H * H), w1((size_t)F * H), w2((size_t)H * F);
        for (size_t i = 0; i < wo.size(); ++i) wo[i] = (float)i;
        for (size_t i = 0; i < w1.size(); ++i) w1[i] = (float)((1u << 20) + i);
        for (size_t i = 0; i < w2.size(); ++i) w2[i] = (float)((1u << 22) + i);
        const size_t n = mx::tail2_stream_elems(F);
        if (n != wo.size() + w1.size() + w2.size()) { printf("FAIL tail2 size F=%d: %zu\n", F, n); return 1; }
        std::vector<uint16_t> out(n);
        g_tags.clear();
        mx::tail2_stream_layout(wo.data(), w1.data(), w2.data(), F, out.data(), to_id);
        std::vector<uint8_t> seen_o(wo.size(), 0), seen_1(w1.size(), 0), seen_2(w2.size(), 0);
        for (uint32_t t : g_tags) {
            uint8_t *sp = t >= (1u << 22) ? &seen_2[t - (1u << 22)] : t >= (1u << 20) ? &seen_1[t - (1u << 20)] : &seen_o[t];
            if (*sp) { printf("FAIL tail2 duplicate F=%d tag=%u\n", F, t); return 1; }
            *sp = 1;
        }
        const int nch = F / 64;
        auto at = [&](size_t frag, int lane, int e) { return g_tags[(frag * 64 + lane) * 8 + e]; };
        // start of the segments in fragments: Wo 0; G1(c): 288 (c = 0), else 288 + 48 + 96 (c - 1); G2(c): G1(c+1) + 48, last: end - 48
        auto g1_at = [&](int c) { return (size_t)(c == 0 ? 288 : 288 + 48 + 96 * (c - 1)); };
        auto g2_at = [&](int c) { return c == nch - 1 ? (size_t)(288 + 96 * nch - 48) : g1_at(c + 1) + 48; };
        for (int lane : {0, 31, 32, 63}) {
            const int r = lane & 31, h = lane >> 5;
            for (int s2 : {0, 23}) for (int fb : {0, 5, 11})
                if (at((size_t)s2 * 12 + fb, lane, 3) != (uint32_t)((fb * 32 + r) * H + 16 * s2 + 8 * h + 3)) { printf("FAIL tail2 Wo\n"); return 1; }
            for (int c : {0, 1, nch - 1}) for (int s2 : {0, 17}) for (int fb : {0, 1})
                if (at(g1_at(c) + (size_t)s2 * 2 + fb, lane, 6) != (1u << 20) + (uint32_t)((c * 64 + fb * 32 + r) * H + 16 * s2 + 8 * h + 6)) { printf("FAIL tail2 G1(%d) F=%d\n", c, F); return 1; }
            for (int c : {0, nch - 2, nch - 1}) for (int s2 : {0, 3}) for (int fb : {0, 7})
                if (at(g2_at(c) + (size_t)s2 * 12 + fb, lane, 1) != (1u << 22) + (uint32_t)((fb * 32 + r) * F + c * 64 + 16 * s2 + 8 * h + 1)) { printf("FAIL tail2 G2(%d) F=%d\n", c, F); return 1; }
        }
        // parameter block: bo g1 be1 b2 g2 be2 | b1
        std::vector<float> v7[7];
        for (int i = 0; i < 7; ++i) { v7[i].resize(i == 3 ? F : H); for (size_t j = 0; j < v7[i].size(); ++j) v7[i][j] = (float)(1000 * i + j); }
        std::vector<float> pp(mx::tail2_param_floats(), -1.0f);
        mx::tail2_param_layout(v7[0].data(), v7[1].data(), v7[2].data(), v7[3].data(), v7[4].data(), v7[5].data(), v7[6].data(), F, pp.data());
        const int order[6] = {0, 1, 2, 4, 5, 6};
        for (int i = 0; i < 6; ++i) if (pp[(size_t)i * H + 7] != v7[order[i]][7]) { printf("FAIL tail2 params %d\n", i); return 1; }
        if (pp[6 * H + F - 1] != v7[3][F - 1] || (F < 1536 && pp[6 * H + F] != 0.0f)) { printf("FAIL tail2 b1\n"); return 1; }
    }
    printf("OK tail stream layout\n");
    return 0;
}
