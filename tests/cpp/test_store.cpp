// C++ restatement of the reference's own storage tests (lib/libmemex/src/storage/local.rs:168-243)
// against memex::HipFlatStore (include/memex_hip.hpp over the C ABI), plus the embedder actor.
// Built and run by tests/test_cpp_host.py.  Exit code 0 = all passed.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

#include "memex_hip.hpp"

using namespace memex;

#define CHECK(c)                                                              \
    do {                                                                      \
        if (!(c)) {                                                           \
            std::fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); \
            std::exit(1);                                                     \
        }                                                                     \
    } while (0)

static std::vector<VectorData> test_data() {  // local.rs:175-199
    return {{"test-one", "test-one", "", {0.0f, 0.1f, 0.2f}, 0},
            {"test-two", "test-two", "", {0.1f, 0.1f, 0.1f}, 0},
            {"test-three", "test-three", "", {0.3f, 0.2f, 0.1f}, 0}};
}

static void test_hnsw(const std::string &tmp) {  // local.rs:201-214
    HipFlatStore store(tmp + "/a");
    store.bulk_insert(test_data());
    auto results = store.search({0.1f, 0.1f, 0.1f}, 3);
    CHECK(results.size() == 3);
    CHECK(results[0].first == "test-two");  // the reference's assertion
    CHECK(results[1].first == "test-three" && results[2].first == "test-one");
    CHECK(results[0].second == 1.0f && results[1].second == 0.9258201f && results[2].second == 0.7745967f);
    store.delete_all();
}

static void test_save_load(const std::string &tmp) {  // local.rs:216-227
    HipFlatStore store(tmp + "/vectortest");
    store.bulk_insert(test_data());
    store.save();
    auto loaded = HipFlatStore::load(tmp + "/vectortest");
    CHECK(loaded->_id_map.size() == store._id_map.size());
    auto a = loaded->search({0.1f, 0.1f, 0.1f}, 3), b = store.search({0.1f, 0.1f, 0.1f}, 3);
    CHECK(a == b);
    store.delete_all();
}

static void test_delete_all(const std::string &tmp) {  // local.rs:229-242
    HipFlatStore store(tmp + "/d");
    store.bulk_insert(test_data());
    store.save();
    store.delete_all();
    CHECK(store._id_map.empty());
    CHECK(store.nb_point() == 0);
    bool failed = false;
    try {
        HipFlatStore::load(tmp + "/d");
    } catch (const VectorStoreError &) {
        failed = true;
    }
    CHECK(failed);  // load fails: files removed
    store.insert(test_data()[0]);
    CHECK(store._id_map.size() == 1 && store._id_map.begin()->first == 1);  // ids restart at 1
}

static void test_factory_and_errors(const std::string &tmp) {  // mod.rs:95-139
    auto vs = get_vector_storage("hnsw://" + tmp + "/coll", "test");
    vs.add_vectors(test_data());
    std::dynamic_pointer_cast<HipFlatStore>(vs.client)->save();
    auto vs2 = get_vector_storage("hip://" + tmp + "/coll", "test");
    auto r = vs2.search({0.3f, 0.2f, 0.1f}, 2);
    CHECK(r.size() == 2 && r[0].first == "test-three" && r[1].first == "test-two");
    vs2.delete_collection();
    CHECK(vs2.search({0.3f, 0.2f, 0.1f}, 2).empty());
    for (const char *bad : {"", "not a uri", "qdrant://x", "opensearch+https://admin@localhost:9200"}) {
        bool unsupported = false;
        try {
            get_vector_storage(bad, "c");
        } catch (const VectorStoreError &e) {
            unsupported = e.kind() == VectorStoreError::Unsupported;
        }
        CHECK(unsupported);
    }
    HipFlatStore s(tmp + "/e");
    bool threw = false;
    try {
        s.delete_("x");
    } catch (const std::logic_error &) {
        threw = true;
    }
    CHECK(threw);  // unimplemented!() in the reference
    s.insert({"a", "a", "", {1.0f, 0.0f}, 0});
    bool dim_err = false;
    try {
        s.insert({"b", "b", "", {1.0f, 0.0f, 0.0f}, 0});
    } catch (const VectorStoreError &e) {
        dim_err = e.kind() == VectorStoreError::InsertionError;
    }
    CHECK(dim_err);
}

// The reference's flow (tasks.rs:59 / handlers.rs:63,81): the worker adds vectors through ITS
// get_vector_storage() handle, the API handler searches through ANOTHER one; neither calls save().
static void test_per_request_handles(const std::string &tmp) {
    const std::string uri = "hnsw://" + tmp + "/flow";
    {
        auto worker = get_vector_storage(uri, "docs");
        worker.add_vectors(test_data());
    }
    auto api = get_vector_storage(uri, "docs");
    auto r = api.search({0.3f, 0.2f, 0.1f}, 1);
    CHECK(r.size() == 1 && r[0].first == "test-three");
    {
        auto worker2 = get_vector_storage(uri, "docs");  // next task appends (incremental save)
        worker2.add_vectors({{"test-four", "d", "", {0.9f, 0.0f, -0.1f}, 3}});
    }
    CHECK(get_vector_storage(uri, "docs").search({0.9f, 0.0f, -0.1f}, 1)[0].first == "test-four");
    HipFlatStore::evict_resident();  // "restart": everything comes back from vectors.mxflat + vectors.meta.json
    auto again = get_vector_storage(uri, "docs");
    CHECK(again.search({0.9f, 0.0f, -0.1f}, 1)[0].first == "test-four");
    CHECK(again.search({0.0f, 0.1f, 0.2f}, 1)[0].first == "test-one");
    CHECK(std::dynamic_pointer_cast<HipFlatStore>(again.client)->_id_map.size() == 4);
    again.delete_collection();
    // the same surface over the in-library sharded index (3 logical shards on device 0)
    auto sh = get_vector_storage("hip://" + tmp + "/sharded", "docs", 0, {0, 0, 0});
    sh.add_vectors(test_data());
    auto rs = sh.search({0.1f, 0.1f, 0.1f}, 3);
    CHECK(rs.size() == 3 && rs[0].first == "test-two" && rs[1].first == "test-three" && rs[2].first == "test-one");
    CHECK(rs[0].second == 1.0f && rs[1].second == 0.9258201f && rs[2].second == 0.7745967f);
    sh.delete_collection();
    HipFlatStore::evict_resident();
}

static void test_embedder_actor(const std::string &g_tmp) {  // embedding.rs:78-152 with a seeded 1-layer MiniLM-shaped encoder
    mx_encoder_cfg cfg{1, 384, 12, 1536, 30522, 512, 2, 1e-12f, MX_POOL_MEAN, 1};
    const size_t n = mx_encoder_weight_bytes(&cfg) / sizeof(float);
    std::vector<float> w(n);
    std::mt19937 rng(5);
    std::normal_distribution<float> nd(0.0f, 0.04f);
    for (auto &x : w) x = nd(rng);
    auto [th, emb] = SentenceEmbedder::spawn(ModelConfig{}, cfg, std::move(w), 128);
    std::string doc;
    for (int i = 0; i < 600; ++i) doc += "w" + std::to_string(i % 97) + " ";
    auto segs = emb->encode(doc);
    CHECK(segs.size() == 4);  // 600 tokens, windows of 256 advancing by 170
    for (auto &s : segs) {
        double nn = 0;
        for (float v : s.vector) nn += (double)v * v;
        CHECK(s.vector.size() == 384 && std::fabs(nn - 1.0) < 1e-3);
    }
    auto one = emb->encode_single("w1 w2 w3");
    CHECK(one.has_value() && one->content == "w1 w2 w3");
    auto again = emb->encode_single("w1 w2 w3");
    CHECK(again->vector == one->vector);
    bool gate = false;
    try {
        segment_text(ModelConfig{EmbeddingsModelType::SentenceT5Base, 256, 86}, "x");
    } catch (const EmbeddingError &e) {
        gate = e.kind() == EmbeddingError::SetupError;
    }
    CHECK(gate);
    emb->shutdown();
    th.join();

    // the same actor with the NATIVE tokenizer (mx_tokenizer_*): segments are the tokenizer's windows, concurrent ingest
    // requests are segmented together (mx_tokenizer_segment_batch) and answered like single ones
    std::vector<std::string> vocab{"[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "w", ".", ","};
    for (int d = 0; d < 10; ++d) vocab.push_back("##" + std::to_string(d));
    while (vocab.size() < 30522) vocab.push_back("[unused" + std::to_string(vocab.size()) + "]");
    auto tok = Tokenizer::wordpiece_from_tokens(vocab);
    std::vector<float> w2(n);
    for (auto &x : w2) x = nd(rng);
    auto [th2, emb2] = SentenceEmbedder::spawn(ModelConfig{}, cfg, std::move(w2), 128, 0, tok);
    std::vector<std::string> docs(6);
    for (int d = 0; d < 6; ++d)
        for (int i = 0; i < 150 + 90 * d; ++i) docs[d] += "W" + std::to_string((i * 7 + d) % 97) + (i % 11 == 0 ? ", " : " ");
    std::vector<std::vector<EmbeddingResult>> alone;
    for (auto &d : docs) alone.push_back(emb2->encode(d));
    for (int d = 0; d < 6; ++d) {
        const auto want = tok->windows(docs[d], 256, 86);
        CHECK(alone[d].size() == want.size() && want.size() >= 2);
        for (size_t i = 0; i < want.size(); ++i) CHECK(alone[d][i].content == want[i] && alone[d][i].vector.size() == 384);
        CHECK(want[0].rfind("w", 0) == 0);  // lower-cased, detokenised text
    }
    std::vector<std::vector<EmbeddingResult>> together(6);
    {
        std::vector<std::thread> ts;
        for (int d = 0; d < 6; ++d) ts.emplace_back([&, d] { together[d] = emb2->encode(docs[d]); });
        for (auto &t : ts) t.join();
    }
    for (int d = 0; d < 6; ++d) {
        CHECK(together[d].size() == alone[d].size());
        for (size_t i = 0; i < alone[d].size(); ++i) {
            CHECK(together[d][i].content == alone[d][i].content);
            double dot = 0;
            for (int j = 0; j < 384; ++j) dot += (double)together[d][i].vector[j] * alone[d][i].vector[j];
            CHECK(dot > 1.0 - 1e-4);  // a row's embedding does not depend on its batch (pass regimes: include/memex_hip.h)
        }
    }
    // the two callers of the path: tasks.rs:9-66 (document -> named segments -> add_vectors) and handlers.rs:55-109
    // (query -> encode_single -> search); ids are the reference's v5 UUIDs
    {
        auto store = get_vector_storage("hip://" + g_tmp + "/callers", "docs");
        std::vector<std::vector<VectorData>> written;
        for (int d = 0; d < 3; ++d) written.push_back(process_embeddings(store, *emb2, 100 + d, docs[d]));
        for (int d = 0; d < 3; ++d) {
            CHECK(written[d].size() == alone[d].size());
            const std::string doc = document_uuid(100 + d);
            for (size_t i = 0; i < written[d].size(); ++i)
                CHECK(written[d][i]._id == segment_uuid(doc, i) && written[d][i].document_id == doc && written[d][i].segment_id == i &&
                      written[d][i].text == alone[d][i].content);
        }
        // the text of a stored window finds that window
        auto hits = search_docs(store, *emb2, written[1][1].text, 3);
        CHECK(hits.size() == 3 && hits[0].first == written[1][1]._id && hits[0].second > 0.999f);
        store.delete_collection();
    }
    emb2->shutdown();
    th2.join();
}

int main(int argc, char **argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp/memex_cpp_test";
    mkdir(tmp.c_str(), 0755);
    test_hnsw(tmp);
    test_save_load(tmp);
    test_delete_all(tmp);
    test_factory_and_errors(tmp);
    test_per_request_handles(tmp);
    test_embedder_actor(tmp);
    std::printf("OK 6 tests\n");
    return 0;
}
