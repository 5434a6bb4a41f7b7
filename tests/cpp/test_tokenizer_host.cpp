// memex::Tokenizer (include/memex_hip.hpp) over the C ABI's mx_tokenizer_*: prints what tests/test_cpp_host.py compares with the
// `tokenizers` package -- ids, decoded text, segment_text windows (one call per document and one batch call).  No GPU.
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "memex_hip.hpp"

static unsigned long long fnv(const std::string &s) {
    unsigned long long h = 1469598103934665603ull;
    for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    try {
        const std::string src = argv[1];  // vocab.txt, or tokenizer.json (what Tokenizer::from_pretrained reads, embedding.rs:163)
        const bool json = src.size() > 5 && src.compare(src.size() - 5, 5, ".json") == 0;
        auto tok = json ? memex::Tokenizer::from_file(src) : memex::Tokenizer::wordpiece(src, true);
        std::vector<std::string> docs;
        std::ifstream f(argv[2]);
        for (std::string line; std::getline(f, line);) docs.push_back(line);
        const auto batch = tok->windows_batch(docs, 256, 86);
        if (batch.size() != docs.size()) return 3;
        for (size_t d = 0; d < docs.size(); ++d) {
            const auto ids = tok->encode(docs[d]);
            unsigned long long hi = 1469598103934665603ull;
            for (int32_t id : ids) hi = (hi ^ (unsigned long long)(unsigned)id) * 1099511628211ull;
            const auto one = tok->windows(docs[d], 256, 86);
            if (one != batch[d]) return 4;
            unsigned long long hw = 0;
            for (const auto &w : one) hw = hw * 31 + fnv(w);
            std::printf("DOC %zu ids %zu %016llx dec %016llx win %zu %016llx\n", d, ids.size(), hi, fnv(tok->decode(ids)), one.size(), hw);
        }
        std::vector<int32_t> ids, lens;
        int S = 0;
        tok->encode_batch(docs, 128, ids, lens, S);
        long total = 0;
        for (int32_t l : lens) total += l;
        std::printf("BATCH S %d rows %zu tokens %ld vocab %d\n", S, lens.size(), total, tok->vocab_size());
        bool refused = false;
        try {
            memex::Tokenizer::wordpiece("/nonexistent/vocab.txt");
        } catch (const memex::EmbeddingError &e) {
            refused = e.kind() == memex::EmbeddingError::SetupError;
        }
        // the reference's ids (RFC 4122 v5 over lib.rs:6's NAMESPACE): printed for the Python side to compare with uuid.uuid5
        for (long task : {0L, 1L, 42L, 123456789L}) {
            const std::string doc = memex::document_uuid(task);
            std::printf("UUID %ld %s %s %s\n", task, doc.c_str(), memex::segment_uuid(doc, 0).c_str(), memex::segment_uuid(doc, 71).c_str());
        }
        std::printf("UUID5 %s\n", memex::uuid5("6ba7b810-9dad-11d1-80b4-00c04fd430c8", "www.example.org").c_str());
        std::printf(refused ? "OK tokenizer host\n" : "missing vocabulary not refused\n");
        return refused ? 0 : 5;
    } catch (const std::exception &e) {
        std::printf("FAILED %s\n", e.what());
        return 1;
    }
}
