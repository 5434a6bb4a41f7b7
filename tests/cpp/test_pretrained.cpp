// memex::load_pretrained_dir (include/memex_pretrained.hpp) on a sentence-transformers directory written by
// tests/test_pretrained.py: prints the configuration and a checksum of the weight blob for the Python side to compare with
// memex_amd.pretrained + pack_weights; "unsupported" directories must throw.  No GPU needed (mx_encoder_weight_bytes is host code).
#include <cstdio>
#include <string>

#include "memex_pretrained.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    try {
        const memex::PretrainedModel pm = memex::load_pretrained_dir(argv[1], argc > 2 ? std::atoi(argv[2]) : MX_PREC_BF16);
        uint64_t h = 0;  // position-weighted sum of the blob's 32-bit words, mod 2^64
        for (size_t i = 0; i < pm.weights.size(); ++i) {
            uint32_t wbits;
            std::memcpy(&wbits, &pm.weights[i], 4);
            h += (uint64_t)wbits * (uint64_t)(i + 1);
        }
        const mx_encoder_cfg &c = pm.cfg;
        std::printf("OK %d %d %d %d %d %d %d %.3e %d %d %d %d %zu %d %zu %016llx %s\n", c.layers, c.hidden, c.heads, c.ffn, c.vocab, c.max_pos,
                    c.type_vocab, (double)c.ln_eps, c.pooling, c.normalize, c.pos_offset, c.precision, pm.max_seq_length, pm.do_lower_case ? 1 : 0,
                    pm.weights.size() * sizeof(float), (unsigned long long)h, pm.vocab_path.c_str());
        return 0;
    } catch (const memex::EmbeddingError &e) {
        std::printf("REFUSED %s\n", e.what());
        return 3;
    }
}
