// memex_hip.hpp -- header-only C++17 host mirror of memex's Rust surface over the C ABI
// (memex_hip.h).  The reference is compiled code (Rust); with no Rust toolchain in the build image
// this is the host side a compiled caller links against, with the reference's names, argument
// meaning and error behaviour:
//
//   reference (lib/libmemex/src)                         here (namespace memex)
//   storage/mod.rs:17-28   VectorData                    VectorData
//   storage/mod.rs:31-48   VectorStoreError (8 variants) VectorStoreError{kind(), what()}
//   storage/mod.rs:55-66   trait VectorStore             class VectorStore (abstract)
//   storage/local.rs:21-166 HnswStore                    class HipFlatStore (exact GPU search)
//   storage/mod.rs:69-93   VectorStorage (Arc<Mutex<..>>) class VectorStorage
//   storage/mod.rs:95-139  get_vector_storage            get_vector_storage (hnsw:// and hip://)
//   llm/embedding.rs:11-22 EmbeddingError / Result       EmbeddingError, EmbeddingResult
//   llm/embedding.rs:58-73 ModelConfig                   ModelConfig (L12-v2, 256, 86)
//   llm/embedding.rs:78-152 SentenceEmbedder             SentenceEmbedder::spawn/encode/encode_single
//   llm/embedding.rs:155-198 segment_text                segment_text (native Tokenizer; a whitespace stand-in without one)
//   llm/embedding.rs:163-195 tokenizers crate calls      Tokenizer (mx_tokenizer_*: WordPiece / byte-level BPE)
//   worker/tasks.rs:9-66    process_embeddings           process_embeddings (+ uuid5 / document_uuid / segment_uuid)
//   api/.../handlers.rs:55-109 handle_search_docs        search_docs
//
// The reference's async fns are blocking calls here (the C ABI is synchronous); the Rust shim in
// INTEGRATION.md wraps them in `async fn` again.
#pragma once
#include <sys/stat.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <deque>
#include <fstream>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "memex_hip.h"

namespace memex {

// ---- storage/mod.rs -----------------------------------------------------------------------------
struct VectorData {  // mod.rs:17-28
    std::string _id, document_id, text;
    std::vector<float> vector;
    size_t segment_id = 0;
};

class VectorStoreError : public std::runtime_error {  // mod.rs:31-48
  public:
    enum Kind { ConnectionError, DeleteError, FileIOError, InsertionError, SearchError, SerdeError, SaveError, Unsupported };
    VectorStoreError(Kind k, const std::string &m) : std::runtime_error(m), kind_(k) {}
    Kind kind() const { return kind_; }

  private:
    Kind kind_;
};

inline VectorStoreError from_status(int rc, VectorStoreError::Kind dflt) {
    const std::string msg = mx_last_error() ? mx_last_error() : "";
    switch (rc) {
        case MX_EDEVICE: return {VectorStoreError::ConnectionError, msg};
        case MX_ESEARCH: return {VectorStoreError::SearchError, msg};
        case MX_EIO: return {VectorStoreError::FileIOError, msg};
        case MX_EUNSUPPORTED: return {VectorStoreError::Unsupported, msg};
        default: return {dflt, msg};
    }
}

using VectorSearchResult = std::pair<std::string, float>;  // (doc_id, score), mod.rs:51

class VectorStore {  // mod.rs:55-66
  public:
    virtual ~VectorStore() = default;
    virtual void delete_(const std::string &id) = 0;
    virtual void delete_all() = 0;
    virtual void bulk_insert(const std::vector<VectorData> &data) = 0;
    virtual void insert(const VectorData &data) = 0;
    virtual std::vector<VectorSearchResult> search(const std::vector<float> &vec, size_t limit) = 0;
};

// ---- storage/local.rs ---------------------------------------------------------------------------
constexpr const char *META_FILE = "vectors.meta.json";  // local.rs:19

class HipFlatStore : public VectorStore {
  public:
    std::string storage_path;
    std::map<size_t, std::string> _id_map;  // local.rs:24 (usize -> _id)

    // devices: empty = one index on `device`; several ordinals = the in-library sharded index
    // (mx_index_open_sharded: rows dealt to the GPUs in blocks, one exchange + merge per search)
    explicit HipFlatStore(const std::string &path, int device = 0, std::vector<int> devices = {})
        : storage_path(path), device_(device), devices_(std::move(devices)) {}  // ::new, local.rs:95
    ~HipFlatStore() override {
        if (idx_) mx_index_close(idx_);
    }
    HipFlatStore(const HipFlatStore &) = delete;
    HipFlatStore &operator=(const HipFlatStore &) = delete;

    // Process-wide registry of resident stores keyed by storage path.  The reference builds a store
    // per request / per task (handlers.rs:61-63, worker/lib.rs:190) and reloads the index each time
    // (storage/mod.rs:115-116); here every handle on a collection shares ONE store object -- GPU index
    // AND id map -- as long as the files it last wrote / read are unchanged on disk.
    static std::map<std::string, std::shared_ptr<HipFlatStore>> &registry() {
        static std::map<std::string, std::shared_ptr<HipFlatStore>> r;
        return r;
    }
    static std::mutex &registry_mu() {
        static std::mutex m;
        return m;
    }
    static void evict_resident() {
        std::lock_guard<std::mutex> lk(registry_mu());
        registry().clear();
    }
    static std::shared_ptr<HipFlatStore> create(const std::string &path, int device = 0, std::vector<int> devices = {}) {
        auto st = std::make_shared<HipFlatStore>(path, device, std::move(devices));
        std::lock_guard<std::mutex> lk(registry_mu());
        registry()[path] = st;
        return st;
    }

    static bool has_store(const std::string &path) {  // local.rs:110-113
        struct stat sb;
        return stat((path + "/" + META_FILE).c_str(), &sb) == 0;
    }

    static std::shared_ptr<HipFlatStore> load(const std::string &path, int device = 0, std::vector<int> devices = {}) {  // local.rs:115-141
        {   // resident and unchanged on disk: attach in O(1) (mx_index_load is O(1) too in that case)
            std::lock_guard<std::mutex> lk(registry_mu());
            auto it = registry().find(path);
            if (it != registry().end()) {
                auto &st = it->second;
                std::lock_guard<std::mutex> l2(st->mu_);
                if (st->meta_known_ && file_sig(path + "/" + META_FILE) == st->meta_sig_ && st->_id_map.size() == st->meta_ids_) {
                    if (st->_id_map.empty()) return st;
                    if (st->idx_ && mx_index_load(st->idx_, path.c_str()) == MX_OK && st->nb_point() == st->_id_map.size()) return st;
                }
            }
        }
        std::ifstream f(path + "/" + META_FILE);
        if (!f) throw VectorStoreError(VectorStoreError::FileIOError, "cannot open " + path + "/" + META_FILE);
        std::stringstream ss;
        ss << f.rdbuf();
        auto store = std::make_shared<HipFlatStore>(path, device, std::move(devices));
        store->_id_map = parse_id_map(ss.str());
        store->meta_sig_ = file_sig(path + "/" + META_FILE);
        store->meta_ids_ = store->_id_map.size();
        store->meta_known_ = true;
        if (!store->_id_map.empty()) {
            int dim = 0;
            uint64_t n = 0;
            int rc = mx_index_store_info(path.c_str(), &dim, &n);
            if (rc != MX_OK) throw from_status(rc, VectorStoreError::FileIOError);
            store->open(dim);
            rc = mx_index_load(store->idx_, path.c_str());
            if (rc != MX_OK) throw from_status(rc, VectorStoreError::FileIOError);
            if (n != store->_id_map.size()) throw VectorStoreError(VectorStoreError::FileIOError, "vector / id count mismatch");
        }
        std::lock_guard<std::mutex> lk(registry_mu());
        registry()[path] = store;
        return store;
    }

    // local.rs:143-165.  The reference calls this after EVERY insert (local.rs:67) and rewrites both
    // files; a save into the store's own directory appends instead: mx_index_save adds the new rows to
    // vectors.mxflat, and the new "id":"_id" pairs are spliced in before the JSON object's closing brace.
    void save(const std::string &path_in = "") {
        std::lock_guard<std::mutex> lk(mu_);
        const std::string path = path_in.empty() ? storage_path : path_in;
        make_dirs(path);
        if (idx_) {
            int rc = mx_index_save(idx_, path.c_str());
            if (rc != MX_OK) throw VectorStoreError(VectorStoreError::SaveError, mx_last_error());
        }
        const std::string meta = path + "/" + META_FILE;
        const bool own = path == storage_path;
        const size_t n = _id_map.size();
        bool appended = false;
        if (own && meta_known_ && file_sig(meta) == meta_sig_ && meta_ids_ > 0 && meta_ids_ <= n) {
            appended = meta_ids_ == n;
            if (!appended) {
                std::fstream f(meta, std::ios::in | std::ios::out | std::ios::binary);
                char last = 0;
                if (f && f.seekg(-1, std::ios::end) && f.get(last) && last == '}') {  // the file ends the way this store left it
                    f.seekp(-1, std::ios::end);                                        // over the closing brace
                    for (size_t i = meta_ids_ + 1; i <= n; ++i) f << ",\"" << i << "\":\"" << escape(_id_map[i]) << "\"";
                    f << "}";
                    f.flush();
                    appended = (bool)f;
                }
            }
        }
        if (!appended) {
            // full rewrite through a temporary file: the vectors are already on disk, the id map must never be
            // left shorter than them (also the way out of a file the splice does not recognise)
            meta_known_ = false;
            const std::string tmp = meta + ".tmp";
            {
                std::ofstream f(tmp);
                if (!f) throw VectorStoreError(VectorStoreError::FileIOError, "cannot write " + tmp);
                f << "{";
                bool first = true;
                for (auto &kv : _id_map) {
                    f << (first ? "" : ",") << "\"" << kv.first << "\":\"" << escape(kv.second) << "\"";
                    first = false;
                }
                f << "}";
                f.flush();
                if (!f) throw VectorStoreError(VectorStoreError::FileIOError, "cannot write " + tmp);
            }
            if (std::rename(tmp.c_str(), meta.c_str()) != 0) throw VectorStoreError(VectorStoreError::FileIOError, "cannot replace " + meta);
        }
        if (own) {
            meta_sig_ = file_sig(meta);
            meta_ids_ = n;
            meta_known_ = true;
        }
    }

    void delete_(const std::string &) override {  // local.rs:29-32: unimplemented!()
        throw std::logic_error("Currently removing a single point is not supported");
    }

    void delete_all() override {  // local.rs:34-53
        std::remove((storage_path + "/" + META_FILE).c_str());
        if (mx_index_remove_files(storage_path.c_str()) != MX_OK) throw VectorStoreError(VectorStoreError::DeleteError, mx_last_error());
        if (idx_ && mx_index_clear(idx_) != MX_OK) throw VectorStoreError(VectorStoreError::DeleteError, mx_last_error());
        _id_map.clear();
        meta_known_ = false;
        meta_ids_ = 0;
    }

    // local.rs:55-69 semantics (ids in order, store persisted before returning), one transfer and
    // one incremental save instead of a save per vector
    void bulk_insert(const std::vector<VectorData> &data) override {
        if (data.empty()) return;
        const size_t d = data[0].vector.size();
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!idx_) {
                open((int)d);
                if (nb_point() != _id_map.size()) mx_index_clear(idx_);  // a stale resident index under this key
            }
        }
        std::vector<float> flat;
        flat.reserve(data.size() * d);
        for (auto &v : data) {
            if (v.vector.size() != (size_t)dim_) throw VectorStoreError(VectorStoreError::InsertionError, "vector dimension mismatch");
            flat.insert(flat.end(), v.vector.begin(), v.vector.end());
        }
        uint64_t first = 0;
        int rc = mx_index_add(idx_, flat.data(), data.size(), &first);
        if (rc != MX_OK) throw from_status(rc, VectorStoreError::InsertionError);
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (size_t i = 0; i < data.size(); ++i) _id_map[(size_t)first + i] = data[i]._id;  // next_id = len + 1 (local.rs:63)
        }
        try {
            save();  // local.rs:67 `let _ = self.save(..)`: errors are ignored there too
        } catch (const VectorStoreError &) {
        }
    }

    void insert(const VectorData &data) override { bulk_insert({data}); }  // local.rs:62-69

    std::vector<VectorSearchResult> search(const std::vector<float> &vec, size_t limit) override {  // local.rs:71-91
        std::vector<VectorSearchResult> out;
        if (!idx_ || limit == 0) return out;
        if (vec.size() != (size_t)dim_) throw VectorStoreError(VectorStoreError::SearchError, "query dimension mismatch");
        std::vector<uint64_t> ids(limit);
        std::vector<float> scores(limit);
        int32_t nf = 0;
        int rc = mx_index_search(idx_, vec.data(), 1, (int)limit, ids.data(), scores.data(), nullptr, &nf);
        if (rc != MX_OK) throw from_status(rc, VectorStoreError::SearchError);
        for (int j = 0; j < nf; ++j) {
            auto it = _id_map.find((size_t)ids[j]);
            if (it == _id_map.end())  // local.rs:80-83 panics; we raise
                throw VectorStoreError(VectorStoreError::SearchError, "Internal inconsistency. Id from vector store not mapped.");
            out.emplace_back(it->second, scores[j]);
        }
        return out;
    }

    uint64_t nb_point() const {  // hnsw.get_nb_point() in the reference's test (local.rs:238)
        uint64_t n = 0;
        if (idx_) mx_index_size(idx_, &n);
        return n;
    }

  private:
    int device_ = 0, dim_ = 0;
    std::vector<int> devices_;
    mx_index *idx_ = nullptr;
    std::mutex mu_;
    // (mtime ns, size) of vectors.meta.json as last written / read, and how many ids it holds
    std::pair<long long, long long> meta_sig_{0, 0};
    size_t meta_ids_ = 0;
    bool meta_known_ = false;

    static std::pair<long long, long long> file_sig(const std::string &p) {
        struct stat sb;
        if (stat(p.c_str(), &sb) != 0) return {-1, -1};
        return {(long long)sb.st_mtim.tv_sec * 1000000000LL + sb.st_mtim.tv_nsec, (long long)sb.st_size};
    }
    static void make_dirs(const std::string &dir) {  // create_dir_all (local.rs:144)
        std::string partial;
        for (size_t i = 0; i <= dir.size(); ++i)
            if (i == dir.size() || dir[i] == '/') {
                if (!partial.empty()) mkdir(partial.c_str(), 0755);
                if (i < dir.size()) partial += '/';
            } else {
                partial += dir[i];
            }
    }

    // keyed by the collection's path: every handle on this collection shares ONE resident GPU index
    void open(int dim) {
        const std::string key = storage_path + "@" + (devices_.empty() ? std::to_string(device_) : std::string("sharded"));
        int rc = devices_.empty() ? mx_index_open(key.c_str(), dim, device_, &idx_)
                                  : mx_index_open_sharded(key.c_str(), dim, (int)devices_.size(), devices_.data(), 0, &idx_);
        if (rc != MX_OK) throw from_status(rc, VectorStoreError::ConnectionError);
        dim_ = dim;
    }
    static std::string escape(const std::string &s) {
        std::string o;
        for (char c : s) {
            if (c == '"' || c == '\\') o += '\\';
            o += c;
        }
        return o;
    }
    // {"<usize>":"<id>", ...} as written by serde_json for HashMap<usize,String> (local.rs:156-163)
    static std::map<size_t, std::string> parse_id_map(const std::string &s) {
        std::map<size_t, std::string> m;
        size_t i = s.find('{');
        if (i == std::string::npos) throw VectorStoreError(VectorStoreError::SerdeError, "expected a JSON object");
        auto read_str = [&](std::string &out) {
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '}') return false;
                ++i;
            }
            if (i >= s.size()) throw VectorStoreError(VectorStoreError::SerdeError, "unterminated JSON");
            ++i;
            out.clear();
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) ++i;
                out += s[i++];
            }
            if (i >= s.size()) throw VectorStoreError(VectorStoreError::SerdeError, "unterminated string");
            ++i;
            return true;
        };
        ++i;
        std::string k, v;
        while (read_str(k)) {
            if (!read_str(v)) throw VectorStoreError(VectorStoreError::SerdeError, "missing value");
            try {
                m[(size_t)std::stoull(k)] = v;
            } catch (const std::exception &) {
                throw VectorStoreError(VectorStoreError::SerdeError, "non-numeric key " + k);
            }
        }
        return m;
    }
};

class VectorStorage {  // mod.rs:69-93
  public:
    explicit VectorStorage(std::shared_ptr<VectorStore> c) : client(std::move(c)) {}
    std::shared_ptr<VectorStore> client;
    void add_vectors(const std::vector<VectorData> &points) {
        std::lock_guard<std::mutex> lk(*mu_);
        client->bulk_insert(points);
    }
    void delete_collection() {
        std::lock_guard<std::mutex> lk(*mu_);
        client->delete_all();
    }
    std::vector<VectorSearchResult> search(const std::vector<float> &query, size_t limit) {
        std::lock_guard<std::mutex> lk(*mu_);
        return client->search(query, limit);
    }

  private:
    std::shared_ptr<std::mutex> mu_ = std::make_shared<std::mutex>();
};

// mod.rs:95-139.  Called per request like the reference's; a collection that is already resident is
// attached, not reloaded.
inline VectorStorage get_vector_storage(const std::string &uri, const std::string &collection, int device = 0,
                                        std::vector<int> devices = {}) {
    const size_t p = uri.find("://");
    if (p == std::string::npos || p == 0) throw VectorStoreError(VectorStoreError::Unsupported, uri);
    const std::string scheme = uri.substr(0, p);
    if (scheme == "hnsw" || scheme == "hip") {  // the reference's file backend, now served from HBM
        const std::string storage = uri.substr(p + 3) + "/" + collection;
        std::string partial;
        for (size_t i = 0; i <= storage.size(); ++i)  // create_dir_all
            if (i == storage.size() || storage[i] == '/') {
                if (!partial.empty()) mkdir(partial.c_str(), 0755);
                if (i < storage.size()) partial += '/';
            } else {
                partial += storage[i];
            }
        std::shared_ptr<VectorStore> store;
        if (HipFlatStore::has_store(storage)) {
            store = HipFlatStore::load(storage, device, devices);
        } else {
            std::shared_ptr<HipFlatStore> res;
            {
                std::lock_guard<std::mutex> lk(HipFlatStore::registry_mu());
                auto it = HipFlatStore::registry().find(storage);
                if (it != HipFlatStore::registry().end() && it->second->_id_map.empty()) res = it->second;
            }
            store = res ? res : HipFlatStore::create(storage, device, devices);
        }
        return VectorStorage(store);
    }
    throw VectorStoreError(VectorStoreError::Unsupported, uri);  // opensearch+https is a remote client: out of scope
}

// ---- llm/embedding.rs ---------------------------------------------------------------------------
class EmbeddingError : public std::runtime_error {  // embedding.rs:11-16
  public:
    enum Kind { EncodingFailure, SetupError };
    EmbeddingError(Kind k, const std::string &m) : std::runtime_error(m), kind_(k) {}
    Kind kind() const { return kind_; }

  private:
    Kind kind_;
};

struct EmbeddingResult {  // embedding.rs:19-22
    std::string content;
    std::vector<float> vector;
};

enum class EmbeddingsModelType {  // embedding.rs:25-33
    DistiluseBaseMultilingualCased, BertBaseNliMeanTokens, AllMiniLmL12V2, AllMiniLmL6V2, AllDistilrobertaV1,
    ParaphraseAlbertSmallV2, SentenceT5Base
};

struct ModelConfig {  // embedding.rs:58-73
    EmbeddingsModelType model = EmbeddingsModelType::AllMiniLmL12V2;
    size_t max_length = 256;
    size_t stride = 86;
};

// STAND-IN tokenizer (the pretrained WordPiece vocabulary is not obtainable offline; a native
// WordPiece segmenter is SURVEY.md section 8 row f-1): lower-case, whitespace split, CRC-32 hash into
// [1000, vocab); [PAD]=0, [CLS]=101, [SEP]=102; windowing arithmetic as tokenizers 0.14.
struct WhitespaceHashTokenizer {
    int vocab = 30522;
    static std::vector<std::string> words(const std::string &text) {
        std::vector<std::string> w;
        std::string cur;
        for (unsigned char c : text) {
            if (std::isspace(c)) {
                if (!cur.empty()) w.push_back(cur), cur.clear();
            } else {
                cur += (char)std::tolower(c);
            }
        }
        if (!cur.empty()) w.push_back(cur);
        return w;
    }
    static uint32_t crc32(const std::string &s) {
        uint32_t c = 0xffffffffu;
        for (unsigned char ch : s) {
            c ^= ch;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
        }
        return ~c;
    }
    int word_id(const std::string &w) const { return 1000 + (int)(crc32(w) % (uint32_t)(vocab - 1000)); }
    std::vector<std::string> windows(const std::string &text, size_t max_length, size_t stride) const {
        const auto ws = words(text);
        std::vector<std::string> out;
        if (ws.empty()) return {""};
        const size_t step = max_length - stride;  // each overflow window starts max_length - stride later
        for (size_t start = 0;; start += step) {
            std::string seg;
            for (size_t i = start; i < ws.size() && i < start + max_length; ++i) seg += (i > start ? " " : "") + ws[i];
            out.push_back(seg);
            if (start + max_length >= ws.size()) break;
        }
        return out;
    }
    void encode_batch(const std::vector<std::string> &texts, size_t max_seq_length, std::vector<int32_t> &ids,
                      std::vector<int32_t> &lens, int &S) const {
        std::vector<std::vector<int32_t>> rows;
        S = 0;
        for (auto &t : texts) {
            std::vector<int32_t> r{101};
            for (auto &w : words(t)) {
                if (r.size() + 1 >= max_seq_length) break;
                r.push_back(word_id(w));
            }
            r.push_back(102);
            S = std::max<int>(S, (int)r.size());
            rows.push_back(std::move(r));
        }
        ids.assign(rows.size() * (size_t)S, 0);
        lens.clear();
        for (size_t i = 0; i < rows.size(); ++i) {
            std::copy(rows[i].begin(), rows[i].end(), ids.begin() + i * S);
            lens.push_back((int32_t)rows[i].size());
        }
    }
};

// The native tokenizer / segmenter (mx_tokenizer_*: WordPiece over a BERT vocab.txt, or byte-level BPE over vocab.json +
// merges.txt for the RoBERTa family): what Tokenizer::from_pretrained + with_truncation + encode / decode give
// segment_text (embedding.rs:163-195) and what rust-bert does to the segments before the forward.
class Tokenizer {
  public:
    static std::shared_ptr<Tokenizer> wordpiece(const std::string &vocab_txt, bool lowercase = true) {
        mx_tokenizer *h = nullptr;
        if (mx_tokenizer_create(vocab_txt.c_str(), lowercase ? 1 : 0, &h) != MX_OK)
            throw EmbeddingError(EmbeddingError::SetupError, mx_last_error());  // "Unable to load model", :166-169
        return std::shared_ptr<Tokenizer>(new Tokenizer(h));
    }
    static std::shared_ptr<Tokenizer> wordpiece_from_tokens(const std::vector<std::string> &tokens, bool lowercase = true) {
        std::string blob;
        for (const std::string &t : tokens) blob += t, blob += '\n';
        mx_tokenizer *h = nullptr;
        if (mx_tokenizer_create_from_memory(blob.data(), blob.size(), lowercase ? 1 : 0, &h) != MX_OK)
            throw EmbeddingError(EmbeddingError::SetupError, mx_last_error());
        return std::shared_ptr<Tokenizer>(new Tokenizer(h));
    }
    static std::shared_ptr<Tokenizer> bpe(const std::string &vocab_json, const std::string &merges_txt) {
        mx_tokenizer *h = nullptr;
        if (mx_tokenizer_create_bpe(vocab_json.c_str(), merges_txt.c_str(), &h) != MX_OK)
            throw EmbeddingError(EmbeddingError::SetupError, mx_last_error());
        return std::shared_ptr<Tokenizer>(new Tokenizer(h));
    }
    // tokenizer.json, the file Tokenizer::from_pretrained itself reads (embedding.rs:163): either kind, chosen by its "model"
    static std::shared_ptr<Tokenizer> from_file(const std::string &tokenizer_json) {
        mx_tokenizer *h = nullptr;
        if (mx_tokenizer_create_from_json(tokenizer_json.c_str(), &h) != MX_OK)
            throw EmbeddingError(EmbeddingError::SetupError, mx_last_error());
        return std::shared_ptr<Tokenizer>(new Tokenizer(h));
    }
    ~Tokenizer() { mx_tokenizer_destroy(h_); }
    Tokenizer(const Tokenizer &) = delete;
    Tokenizer &operator=(const Tokenizer &) = delete;

    std::vector<int32_t> encode(const std::string &text, bool add_special_tokens = false) const {
        std::vector<int32_t> ids(text.size() + 2);
        int n = 0;
        check(mx_tokenizer_encode(h_, text.c_str(), add_special_tokens ? 1 : 0, ids.data(), (int)ids.size(), &n), text);
        ids.resize((size_t)n);
        return ids;
    }
    std::string decode(const std::vector<int32_t> &ids, bool skip_special_tokens = true) const {
        size_t nb = 0;
        check(mx_tokenizer_decode(h_, ids.data(), (int)ids.size(), skip_special_tokens ? 1 : 0, nullptr, 0, &nb), "");
        std::string out(nb, '\0');
        check(mx_tokenizer_decode(h_, ids.data(), (int)ids.size(), skip_special_tokens ? 1 : 0, out.data(), nb, &nb), "");
        out.resize(nb ? nb - 1 : 0);
        return out;
    }
    // segment_text's windows for several documents in one call (documents are dealt to host threads)
    std::vector<std::vector<std::string>> windows_batch(const std::vector<std::string> &texts, size_t max_length, size_t stride) const {
        std::vector<const char *> ptrs;
        size_t bytes = 0;
        for (const std::string &t : texts) ptrs.push_back(t.c_str()), bytes += t.size();
        std::vector<int32_t> nseg(texts.size());
        std::string buf(bytes * 3 + 64 * texts.size() + 64, '\0');
        size_t nb = 0;
        for (int pass = 0; pass < 2; ++pass) {
            check(mx_tokenizer_segment_batch(h_, ptrs.data(), (int)texts.size(), (int)max_length, (int)stride, buf.data(), buf.size(), &nb,
                                             nseg.data()), "");
            if (nb <= buf.size()) break;
            buf.assign(nb, '\0');
        }
        std::vector<std::vector<std::string>> out(texts.size());
        const char *p = buf.data();
        for (size_t i = 0; i < texts.size(); ++i)
            for (int32_t k = 0; k < nseg[i]; ++k) {
                out[i].emplace_back(p);
                p += out[i].back().size() + 1;
            }
        return out;
    }
    std::vector<std::string> windows(const std::string &text, size_t max_length, size_t stride) const {
        return windows_batch({text}, max_length, stride)[0];
    }
    // [CLS] .. [SEP] rows, truncated to max_seq_length, [PAD]-padded to the batch maximum S
    void encode_batch(const std::vector<std::string> &texts, size_t max_seq_length, std::vector<int32_t> &ids, std::vector<int32_t> &lens,
                      int &S) const {
        std::vector<const char *> ptrs;
        for (const std::string &t : texts) ptrs.push_back(t.c_str());
        std::vector<int32_t> wide(texts.size() * max_seq_length);
        lens.assign(texts.size(), 0);
        S = 0;
        check(mx_tokenizer_encode_batch(h_, ptrs.data(), (int)texts.size(), (int)max_seq_length, wide.data(), (int)max_seq_length,
                                        lens.data(), &S), "");
        ids.resize(texts.size() * (size_t)S);
        for (size_t b = 0; b < texts.size(); ++b)
            std::copy(wide.begin() + b * max_seq_length, wide.begin() + b * max_seq_length + S, ids.begin() + b * (size_t)S);
    }
    int vocab_size() const {
        int n = 0;
        mx_tokenizer_vocab_size(h_, &n);
        return n;
    }

  private:
    explicit Tokenizer(mx_tokenizer *h) : h_(h) {}
    static void check(int rc, const std::string &text) {
        if (rc != MX_OK) throw EmbeddingError(EmbeddingError::EncodingFailure, text.empty() ? mx_last_error() : text);  // :176-179
    }
    mx_tokenizer *h_;
};

inline void check_segmentable(const ModelConfig &mc) {
    if (mc.model != EmbeddingsModelType::AllMiniLmL12V2 && mc.model != EmbeddingsModelType::AllMiniLmL6V2 &&
        mc.model != EmbeddingsModelType::AllDistilrobertaV1)
        throw EmbeddingError(EmbeddingError::SetupError, "Model not supported yet");  // :160
}

// embedding.rs:155-198.  `tok`: the model's tokenizer; without one the whitespace stand-in above (tests and benchmarks only)
inline std::vector<std::string> segment_text(const ModelConfig &mc, const std::string &text, const Tokenizer *tok = nullptr) {
    check_segmentable(mc);
    if (tok) return tok->windows(text, mc.max_length, mc.stride);
    return WhitespaceHashTokenizer{}.windows(text, mc.max_length, mc.stride);
}

// The actor of embedding.rs:78-152: a dedicated thread owns the encoder; callers post
// (text, segment?, reply) messages on a queue bounded at 100 (sync_channel(100), :87).
class SentenceEmbedder {
  public:
    // `weights`: f32 blob in the order documented in memex_hip.h for `cfg`.
    // `tok`: the model's tokenizer (Tokenizer::wordpiece / ::bpe); nullptr = the whitespace stand-in (tests, benchmarks).
    static std::pair<std::thread, std::shared_ptr<SentenceEmbedder>> spawn(const ModelConfig &mc, const mx_encoder_cfg &cfg,
                                                                           std::vector<float> weights, size_t max_seq_length,
                                                                           int device = 0, std::shared_ptr<Tokenizer> tok = nullptr) {
        auto self = std::shared_ptr<SentenceEmbedder>(new SentenceEmbedder());
        std::promise<std::string> ready;
        auto fut = ready.get_future();
        std::thread th([self, mc, cfg, w = std::move(weights), max_seq_length, device, tok, pr = std::move(ready)]() mutable {
            self->runner(mc, cfg, w, max_seq_length, device, tok, pr);
        });
        const std::string err = fut.get();
        if (!err.empty()) {
            th.join();
            throw EmbeddingError(EmbeddingError::SetupError, err);
        }
        return {std::move(th), self};
    }
    std::vector<EmbeddingResult> encode(const std::string &text) { return call(text, true); }  // :138-142
    std::optional<EmbeddingResult> encode_single(const std::string &text) {                   // :146-151
        auto r = call(text, false);
        if (r.empty()) return std::nullopt;
        return r.back();
    }
    void shutdown() { post({"", false, nullptr, true}); }

  private:
    struct Msg {
        std::string text;
        bool segment;
        std::shared_ptr<std::promise<std::vector<EmbeddingResult>>> reply;
        bool stop = false;
    };
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Msg> q_;

    void post(Msg m) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return q_.size() < 100; });
        q_.push_back(std::move(m));
        cv_.notify_all();
    }
    std::vector<EmbeddingResult> call(const std::string &text, bool segment) {
        auto pr = std::make_shared<std::promise<std::vector<EmbeddingResult>>>();
        auto fut = pr->get_future();
        post({text, segment, pr});
        return fut.get();
    }
    void runner(const ModelConfig &mc, const mx_encoder_cfg &cfg, const std::vector<float> &w, size_t max_seq_length, int device,
                const std::shared_ptr<Tokenizer> &native, std::promise<std::string> &ready) {
        mx_encoder *enc = nullptr;
        if (mx_encoder_cfg_size() != sizeof(mx_encoder_cfg)) {  // this header and the loaded library disagree about the struct
            ready.set_value("mx_encoder_cfg size mismatch: header and libmemex_hip.so are different releases");
            return;
        }
        int rc = mx_encoder_create(&cfg, w.data(), w.size() * sizeof(float), device, &enc);  // create_model(), :99-100
        if (rc != MX_OK) {
            ready.set_value(mx_last_error());
            return;
        }
        ready.set_value("");
        WhitespaceHashTokenizer tok{cfg.vocab};
        // Two stages: this thread segments and tokenises batch i+1 while `gpu` has batch i in mx_encoder_encode (host work per
        // document is about as long as its encoder pass).  A slot of one batch between them.
        using Reply = std::shared_ptr<std::promise<std::vector<EmbeddingResult>>>;
        struct Batch {
            std::vector<std::pair<Reply, std::vector<std::string>>> work;
            std::vector<int32_t> ids, lens;
            int S = 0;
            size_t rows = 0;
            bool last = false;
        };
        std::mutex smu;
        std::condition_variable scv;
        std::unique_ptr<Batch> slot;
        bool gpu_busy = false;
        std::thread gpu([&] {
            for (;;) {
                std::unique_ptr<Batch> b;
                {
                    std::unique_lock<std::mutex> lk(smu);
                    scv.wait(lk, [&] { return slot != nullptr; });
                    b = std::move(slot);
                    gpu_busy = true;
                    scv.notify_all();
                }
                if (b->last) return;
                try {
                    std::vector<float> out(b->rows * (size_t)cfg.hidden);
                    const int erc = mx_encoder_encode(enc, b->ids.data(), b->lens.data(), (int)b->rows, b->S, out.data());  // model.encode, :109
                    if (erc != MX_OK) throw EmbeddingError(EmbeddingError::EncodingFailure, mx_last_error());
                    size_t o = 0;
                    for (auto &wk : b->work) {
                        std::vector<EmbeddingResult> res;
                        for (size_t i = 0; i < wk.second.size(); ++i, ++o)
                            res.push_back({wk.second[i], std::vector<float>(out.begin() + o * cfg.hidden, out.begin() + (o + 1) * cfg.hidden)});
                        wk.first->set_value(std::move(res));
                    }
                } catch (...) {
                    for (auto &wk : b->work) {
                        try {
                            wk.first->set_exception(std::current_exception());
                        } catch (const std::future_error &) {  // already answered before the failure
                        }
                    }
                }
                std::lock_guard<std::mutex> lk(smu);
                gpu_busy = false;
            }
        });
        auto hand_over = [&](std::unique_ptr<Batch> b) {
            std::unique_lock<std::mutex> lk(smu);
            scv.wait(lk, [&] { return slot == nullptr; });
            slot = std::move(b);
            scv.notify_all();
        };
        bool stop = false;
        while (!stop) {
            // requests that queued up while the previous batch was on the GPU are embedded together
            // (the reference's runner takes one message per model.encode, :101-109; a row's embedding
            // does not depend on its batch).  With the GPU stage idle only half of what is waiting is taken:
            // synchronous callers come back with their next document only after a reply, so two smaller
            // batches keep both stages busy.
            std::vector<Msg> msgs;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !q_.empty(); });
                bool idle;
                {
                    std::lock_guard<std::mutex> slk(smu);  // (the GPU stage never takes mu_: no lock-order cycle)
                    idle = !gpu_busy && slot == nullptr;
                }
                const size_t limit = idle ? std::max<size_t>(1, (q_.size() + 1) / 2) : 64;
                while (!q_.empty() && msgs.size() < limit) {
                    msgs.push_back(std::move(q_.front()));
                    q_.pop_front();
                }
                cv_.notify_all();
            }
            auto batch = std::make_unique<Batch>();
            // the documents of the drained requests are segmented together (one host thread per document)
            std::vector<std::vector<std::string>> pre;
            std::vector<size_t> pre_of(msgs.size(), (size_t)-1);
            if (native) {
                std::vector<std::string> docs;
                for (size_t i = 0; i < msgs.size(); ++i)
                    if (!msgs[i].stop && msgs[i].segment) pre_of[i] = docs.size(), docs.push_back(msgs[i].text);
                try {
                    check_segmentable(mc);
                    if (docs.size() > 1) pre = native->windows_batch(docs, mc.max_length, mc.stride);
                } catch (...) {  // reported per request below
                }
            }
            for (size_t i = 0; i < msgs.size(); ++i) {
                Msg &m = msgs[i];
                if (m.stop) {
                    stop = true;
                    continue;
                }
                try {
                    if (m.segment && pre_of[i] < pre.size()) batch->work.emplace_back(m.reply, std::move(pre[pre_of[i]]));
                    else batch->work.emplace_back(m.reply, m.segment ? segment_text(mc, m.text, native.get()) : std::vector<std::string>{m.text});  // :103-107
                } catch (...) {
                    m.reply->set_exception(std::current_exception());
                }
            }
            if (batch->work.empty()) continue;
            try {
                std::vector<std::string> flat;
                for (auto &wk : batch->work) flat.insert(flat.end(), wk.second.begin(), wk.second.end());
                if (native) native->encode_batch(flat, max_seq_length, batch->ids, batch->lens, batch->S);
                else tok.encode_batch(flat, max_seq_length, batch->ids, batch->lens, batch->S);
                batch->rows = flat.size();
            } catch (...) {
                for (auto &wk : batch->work) wk.first->set_exception(std::current_exception());
                continue;
            }
            hand_over(std::move(batch));
        }
        {
            auto fin = std::make_unique<Batch>();
            fin->last = true;
            hand_over(std::move(fin));
        }
        gpu.join();
        mx_encoder_destroy(enc);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// The two callers of the hot path, as far as the path goes (no SQL, no HTTP): what the worker does with a document
// (lib/worker/src/tasks.rs:9-66) and what the API does with a query (lib/api/src/endpoints/collections/handlers.rs:55-109).
// Segment ids are the reference's own: RFC 4122 version-5 UUIDs over its NAMESPACE (lib/libmemex/src/lib.rs:6).
// ---------------------------------------------------------------------------------------------------------------------
namespace detail {
inline void sha1(const std::string &msg, uint8_t out[20]) {
    uint32_t h[5] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u, 0xc3d2e1f0u};
    std::string m = msg;
    const uint64_t bits = (uint64_t)msg.size() * 8;
    m.push_back((char)0x80);
    while (m.size() % 64 != 56) m.push_back('\0');
    for (int i = 7; i >= 0; --i) m.push_back((char)((bits >> (8 * i)) & 0xff));
    auto rol = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
    for (size_t off = 0; off < m.size(); off += 64) {
        uint32_t w[80];
        for (int i = 0; i < 16; ++i)
            w[i] = (uint32_t)(uint8_t)m[off + 4 * i] << 24 | (uint32_t)(uint8_t)m[off + 4 * i + 1] << 16 |
                   (uint32_t)(uint8_t)m[off + 4 * i + 2] << 8 | (uint32_t)(uint8_t)m[off + 4 * i + 3];
        for (int i = 16; i < 80; ++i) w[i] = rol(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
        for (int i = 0; i < 80; ++i) {
            uint32_t f, k;
            if (i < 20) f = (b & c) | (~b & d), k = 0x5a827999u;
            else if (i < 40) f = b ^ c ^ d, k = 0x6ed9eba1u;
            else if (i < 60) f = (b & c) | (b & d) | (c & d), k = 0x8f1bbcdcu;
            else f = b ^ c ^ d, k = 0xca62c1d6u;
            const uint32_t t = rol(a, 5) + f + e + k + w[i];
            e = d, d = c, c = rol(b, 30), b = a, a = t;
        }
        h[0] += a, h[1] += b, h[2] += c, h[3] += d, h[4] += e;
    }
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
}
}  // namespace detail

// Uuid::new_v5(namespace, name): SHA-1 over the namespace's 16 bytes and the name, version and variant bits set
inline std::string uuid5(const std::string &namespace_uuid, const std::string &name) {
    std::string bytes;
    int hi = -1;
    for (char c : namespace_uuid) {
        if (c == '-') continue;
        const int v = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
        if (v < 0) throw std::invalid_argument("uuid5: bad namespace");
        if (hi < 0) hi = v;
        else bytes.push_back((char)(hi << 4 | v)), hi = -1;
    }
    if (bytes.size() != 16 || hi >= 0) throw std::invalid_argument("uuid5: bad namespace");
    uint8_t d[20];
    detail::sha1(bytes + name, d);
    d[6] = (uint8_t)((d[6] & 0x0f) | 0x50);
    d[8] = (uint8_t)((d[8] & 0x3f) | 0x80);
    static const char *hex = "0123456789abcdef";
    std::string out;
    for (int i = 0; i < 16; ++i) {
        if (i == 4 || i == 6 || i == 8 || i == 10) out.push_back('-');
        out.push_back(hex[d[i] >> 4]);
        out.push_back(hex[d[i] & 15]);
    }
    return out;
}

inline const std::string &memex_namespace() {  // lib/libmemex/src/lib.rs:6
    static const std::string ns = "5fdfe40a-de2c-11ed-bfa7-00155deae876";
    return ns;
}
// db/document.rs:74: Uuid::new_v5(&NAMESPACE, task.id.to_string().as_bytes())
inline std::string document_uuid(int64_t task_id) { return uuid5(memex_namespace(), std::to_string(task_id)); }
// tasks.rs:36-40: Uuid::new_v5(&NAMESPACE, format!("{doc_uuid}-{idx}").as_bytes())
inline std::string segment_uuid(const std::string &doc_uuid, size_t idx) { return uuid5(memex_namespace(), doc_uuid + "-" + std::to_string(idx)); }

// tasks.rs:9-66 without the SQL: embedder.encode(content) -> one VectorData per window -> client.add_vectors.  Returns the
// VectorData the reference also writes to its `embeddings` table; a failing add_vectors is the caller's to log (it throws here).
inline std::vector<VectorData> process_embeddings(VectorStorage &client, SentenceEmbedder &embedder, int64_t task_id, const std::string &content) {
    const std::vector<EmbeddingResult> embeddings = embedder.encode(content);  // :19
    const std::string doc = document_uuid(task_id);                            // :28
    std::vector<VectorData> vectors;
    for (size_t idx = 0; idx < embeddings.size(); ++idx)                       // :34-56
        vectors.push_back(VectorData{segment_uuid(doc, idx), doc, embeddings[idx].content, embeddings[idx].vector, idx});
    client.add_vectors(vectors);                                               // :59
    return vectors;
}

// handlers.rs:72-85: embed the query, search; std::invalid_argument("Invalid query") where the handler rejects (:74-78)
inline std::vector<VectorSearchResult> search_docs(VectorStorage &client, SentenceEmbedder &embedder, const std::string &query, size_t limit = 10) {
    const std::optional<EmbeddingResult> res = embedder.encode_single(query);
    if (!res) throw std::invalid_argument("Invalid query");
    return client.search(res->vector, limit);
}

}  // namespace memex
