// memex_pretrained.hpp -- header-only C++17: build an encoder configuration + weight blob from a LOCAL
// sentence-transformers model directory, i.e. from the files `SentenceEmbeddingsBuilder::remote(..).create_model()`
// downloads in the reference (lib/libmemex/src/llm/embedding.rs:99-100; rust-bert 0.21.0):
//     modules.json, config.json, sentence_bert_config.json, 1_Pooling/config.json, model.safetensors, vocab.txt
// The C++ twin of memex_amd/pretrained.py (same rules, same refusals): what the HIP encoder does not implement -- a 2_Dense
// module, max / sqrt-length pooling, non-GELU activations, relative positions -- throws EmbeddingError(SetupError), never
// an approximation.  Weights: model.safetensors only (F32 / F16 / BF16 tensors); pytorch_model.bin and rust_model.ot are
// pickles, convert them first.  No dependency beyond the standard library.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "memex_hip.hpp"

namespace memex {

namespace pretrained_detail {

// ---- a JSON value tree, just enough for HF config files and the safetensors header ----
struct Json {
    enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json *get(const std::string &k) const {
        for (auto &kv : obj)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
    double number(const std::string &k, double dflt) const {
        const Json *v = get(k);
        return v && v->type == Num ? v->num : dflt;
    }
    std::string string(const std::string &k, const std::string &dflt) const {
        const Json *v = get(k);
        return v && v->type == Str ? v->str : dflt;
    }
    bool truthy(const std::string &k) const {
        const Json *v = get(k);
        return v && ((v->type == Bool && v->b) || (v->type == Num && v->num != 0.0));
    }
};

class JsonParser {
  public:
    explicit JsonParser(const std::string &s) : s_(s) {}
    Json parse() {
        Json v = value();
        ws();
        if (i_ != s_.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string &s_;
    size_t i_ = 0;
    int depth_ = 0;
    struct Nest {  // a hostile file must not turn nesting into stack depth
        JsonParser &p;
        explicit Nest(JsonParser &q) : p(q) {
            if (++p.depth_ > 64) p.fail("nested too deeply");
        }
        ~Nest() { --p.depth_; }
    };
    [[noreturn]] void fail(const char *m) const { throw EmbeddingError(EmbeddingError::SetupError, std::string("JSON: ") + m + " at byte " + std::to_string(i_)); }
    void ws() {
        while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) ++i_;
    }
    Json value() {
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        const char c = s_[i_];
        Json v;
        const Nest nest(*this);
        if (c == '{') {
            v.type = Json::Obj;
            ++i_;
            ws();
            if (i_ < s_.size() && s_[i_] == '}') return ++i_, v;
            for (;;) {
                ws();
                std::string k = string();
                ws();
                if (i_ >= s_.size() || s_[i_] != ':') fail("expected ':'");
                ++i_;
                v.obj.emplace_back(std::move(k), value());
                ws();
                if (i_ < s_.size() && s_[i_] == ',') {
                    ++i_;
                    continue;
                }
                if (i_ < s_.size() && s_[i_] == '}') return ++i_, v;
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.type = Json::Arr;
            ++i_;
            ws();
            if (i_ < s_.size() && s_[i_] == ']') return ++i_, v;
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (i_ < s_.size() && s_[i_] == ',') {
                    ++i_;
                    continue;
                }
                if (i_ < s_.size() && s_[i_] == ']') return ++i_, v;
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.type = Json::Str;
            v.str = string();
            return v;
        }
        if (s_.compare(i_, 4, "true") == 0) return i_ += 4, v.type = Json::Bool, v.b = true, v;
        if (s_.compare(i_, 5, "false") == 0) return i_ += 5, v.type = Json::Bool, v;
        if (s_.compare(i_, 4, "null") == 0) return i_ += 4, v;
        size_t used = 0;
        try {
            v.num = std::stod(s_.substr(i_, 64), &used);
        } catch (...) {
            fail("bad value");
        }
        v.type = Json::Num;
        i_ += used;
        return v;
    }
    std::string string() {
        if (i_ >= s_.size() || s_[i_] != '"') fail("expected a string");
        ++i_;
        std::string out;
        while (i_ < s_.size() && s_[i_] != '"') {
            if (s_[i_] == '\\' && i_ + 1 < s_.size()) {
                const char e = s_[i_ + 1];
                i_ += 2;
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {  // BMP code point -> UTF-8 (config files only use it for the odd special token)
                        if (i_ + 4 > s_.size()) fail("bad \\u escape");
                        unsigned cp = 0;
                        for (int h = 0; h < 4; ++h) {
                            const char x = s_[i_ + h];
                            const int d = x >= '0' && x <= '9' ? x - '0' : x >= 'a' && x <= 'f' ? x - 'a' + 10 : x >= 'A' && x <= 'F' ? x - 'A' + 10 : -1;
                            if (d < 0) fail("bad \\u escape");
                            cp = cp * 16 + (unsigned)d;
                        }
                        i_ += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) out += (char)(0xC0 | (cp >> 6)), out += (char)(0x80 | (cp & 0x3F));
                        else out += (char)(0xE0 | (cp >> 12)), out += (char)(0x80 | ((cp >> 6) & 0x3F)), out += (char)(0x80 | (cp & 0x3F));
                        break;
                    }
                    default: out += e;
                }
            } else {
                out += s_[i_++];
            }
        }
        if (i_ >= s_.size()) fail("unterminated string");
        ++i_;
        return out;
    }
};

inline bool file_exists(const std::string &p) {
    struct stat st {};
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
inline std::string slurp(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    if (!f) throw EmbeddingError(EmbeddingError::SetupError, "Unable to load model <" + p + ">: cannot open");
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
inline Json read_json(const std::string &p) {
    const std::string text = slurp(p);
    return JsonParser(text).parse();
}
inline float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {  // subnormal: renormalise
            int e = -1;
            uint32_t m = man;
            do {
                ++e;
                m <<= 1;
            } while (!(m & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3FFu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

// model.safetensors: u64 header length | JSON header {name: {dtype, shape, data_offsets}} | raw little-endian data
class SafeTensors {
  public:
    explicit SafeTensors(const std::string &path) : data_(slurp(path)) {
        if (data_.size() < 8) bad(path, "shorter than its header length");
        uint64_t hl = 0;
        std::memcpy(&hl, data_.data(), 8);
        if (hl > data_.size() - 8) bad(path, "header runs past the end of the file");
        const std::string hdr = data_.substr(8, (size_t)hl);
        header_ = JsonParser(hdr).parse();
        base_ = 8 + (size_t)hl;
        if (header_.type != Json::Obj) bad(path, "header is not an object");
    }
    // the tensor under any of `prefixes` + name as f32, with its shape checked; false = absent
    bool fetch(const std::string &name, const std::vector<int64_t> &shape, float *out) const {
        static const char *prefixes[] = {"", "bert.", "roberta.", "0.auto_model.", "model.", "auto_model."};
        for (const char *p : prefixes) {
            const Json *t = header_.get(std::string(p) + name);
            if (!t || t->type != Json::Obj) continue;
            const Json *sh = t->get("shape"), *off = t->get("data_offsets");
            if (!sh || !off || off->arr.size() != 2) throw EmbeddingError(EmbeddingError::SetupError, "safetensors: malformed entry " + name);
            size_t n = 1;
            bool same = sh->arr.size() == shape.size();
            for (size_t i = 0; i < sh->arr.size() && same; ++i) {
                const double dnum = sh->arr[i].num;
                same = sh->arr[i].type == Json::Num && dnum >= 0.0 && dnum < 1e12 && (int64_t)dnum == shape[i];
                if (same) n *= (size_t)dnum;
            }
            if (!same) throw EmbeddingError(EmbeddingError::SetupError, name + ": shape differs from what config.json implies");
            const std::string dt = t->string("dtype", "");
            const double o0 = off->arr[0].num, o1 = off->arr[1].num;
            if (off->arr[0].type != Json::Num || off->arr[1].type != Json::Num || !(o0 >= 0.0) || !(o1 >= o0) || !(o1 <= (double)data_.size()))
                throw EmbeddingError(EmbeddingError::SetupError, name + ": data_offsets outside the file");
            const size_t b0 = base_ + (size_t)o0, b1 = base_ + (size_t)o1;
            const size_t esz = dt == "F32" ? 4 : (dt == "F16" || dt == "BF16") ? 2 : 0;
            if (!esz) throw EmbeddingError(EmbeddingError::SetupError, name + ": dtype " + dt + " is not supported");
            if (b1 > data_.size() || b1 - b0 != n * esz) throw EmbeddingError(EmbeddingError::SetupError, name + ": data range does not match its shape");
            const char *src = data_.data() + b0;
            if (dt == "F32") std::memcpy(out, src, n * 4);
            else
                for (size_t i = 0; i < n; ++i) {
                    uint16_t h;
                    std::memcpy(&h, src + 2 * i, 2);
                    if (dt == "F16") out[i] = half_to_float(h);
                    else {
                        const uint32_t bits = (uint32_t)h << 16;
                        std::memcpy(&out[i], &bits, 4);
                    }
                }
            return true;
        }
        return false;
    }

  private:
    std::string data_;
    Json header_;
    size_t base_ = 0;
    [[noreturn]] static void bad(const std::string &p, const char *m) { throw EmbeddingError(EmbeddingError::SetupError, p + ": " + m); }
};

}  // namespace pretrained_detail

struct PretrainedModel {
    mx_encoder_cfg cfg{};
    std::vector<float> weights;   // the blob mx_encoder_create takes (include/memex_hip.h)
    std::string vocab_path;       // vocab.txt of a WordPiece model ("" otherwise): mx_tokenizer_create
    std::string vocab_json_path, merges_path;  // byte-level BPE files of a RoBERTa-family model ("" otherwise): mx_tokenizer_create_bpe
    std::string tokenizer_json_path;  // tokenizer.json ("" when absent): the fallback when neither of the above is there
    size_t max_seq_length = 0;    // sentence_bert_config.json
    bool do_lower_case = true;    // tokenizer_config.json (default: BERT uncased)
    std::vector<std::string> modules;  // module types of modules.json in order
};

inline PretrainedModel load_pretrained_dir_impl(const std::string &dir, int precision);

// -> configuration, f32 weight blob and tokenizer files of a sentence-transformers directory.  Throws EmbeddingError(SetupError)
// and nothing else: whatever a damaged file provokes underneath (a bad number, an allocation the header asked for) is reported
// as "Unable to load model", like the reference's create_model() failing (embedding.rs:99-100).
inline PretrainedModel load_pretrained_dir(const std::string &dir, int precision = -1 /* the loader's choice */) {
    try {
        return load_pretrained_dir_impl(dir, precision);
    } catch (const EmbeddingError &) {
        throw;
    } catch (const std::exception &e) {
        throw EmbeddingError(EmbeddingError::SetupError, "Unable to load model <" + dir + ">: " + e.what());
    }
}

inline PretrainedModel load_pretrained_dir_impl(const std::string &dir, int precision) {
    using namespace pretrained_detail;
    auto unsupported = [&](const std::string &what) -> EmbeddingError {
        return EmbeddingError(EmbeddingError::SetupError, "Unable to load model <" + dir + ">: " + what);
    };
    PretrainedModel pm;
    std::string pooling_dir = "1_Pooling", tdir = dir;
    bool normalize = false;
    if (file_exists(dir + "/modules.json")) {
        const Json mods = read_json(dir + "/modules.json");
        for (const Json &m : mods.arr) {
            const std::string full = m.string("type", ""), path = m.string("path", "");
            const std::string ty = full.substr(full.rfind('.') == std::string::npos ? 0 : full.rfind('.') + 1);
            pm.modules.push_back(ty);
            if (ty == "Transformer") tdir = path.empty() ? dir : dir + "/" + path;
            else if (ty == "Pooling") pooling_dir = path.empty() ? "1_Pooling" : path;
            else if (ty == "Normalize") normalize = true;
            else throw unsupported("module '" + (path.empty() ? ty : path) + "' (" + full + ") is not supported");
        }
    } else {
        pm.modules.push_back("Transformer");
    }
    const Json hc = read_json(tdir + "/config.json");
    const std::string mtype = hc.string("model_type", "bert");
    if (mtype != "bert" && mtype != "roberta" && mtype != "xlm-roberta" && mtype != "distilroberta") throw unsupported("model_type '" + mtype + "'");
    if (hc.string("hidden_act", "gelu") != "gelu") throw unsupported("hidden_act '" + hc.string("hidden_act", "") + "' (erf GELU only)");
    if (hc.string("position_embedding_type", "absolute") != "absolute") throw unsupported("position_embedding_type");
    mx_encoder_cfg &c = pm.cfg;
    c.layers = (int32_t)hc.number("num_hidden_layers", 0);
    c.hidden = (int32_t)hc.number("hidden_size", 0);
    c.heads = (int32_t)hc.number("num_attention_heads", 0);
    c.ffn = (int32_t)hc.number("intermediate_size", 0);
    c.vocab = (int32_t)hc.number("vocab_size", 0);
    c.max_pos = (int32_t)hc.number("max_position_embeddings", 0);
    c.type_vocab = (int32_t)hc.number("type_vocab_size", 2);
    c.ln_eps = (float)hc.number("layer_norm_eps", 1e-12);
    c.pos_offset = mtype == "bert" ? 0 : (int32_t)hc.number("pad_token_id", 1) + 1;
    c.normalize = normalize ? 1 : 0;
    c.precision = precision;
    if ((int32_t)hc.number("embedding_size", c.hidden) != c.hidden) throw unsupported("factorised embeddings");
    c.pooling = MX_POOL_MEAN;
    if (file_exists(dir + "/" + pooling_dir + "/config.json")) {
        const Json pc = read_json(dir + "/" + pooling_dir + "/config.json");
        const char *modes[] = {"pooling_mode_cls_token", "pooling_mode_mean_tokens", "pooling_mode_max_tokens",
                               "pooling_mode_mean_sqrt_len_tokens", "pooling_mode_weightedmean_tokens", "pooling_mode_lasttoken"};
        int on = 0, which = -1;
        for (int i = 0; i < 6; ++i)
            if (pc.truthy(modes[i])) ++on, which = i;
        if (on != 1 || which > 1) throw unsupported("pooling modes other than CLS / mean");
        c.pooling = which == 0 ? MX_POOL_CLS : MX_POOL_MEAN;
        if ((int32_t)pc.number("word_embedding_dimension", c.hidden) != c.hidden) throw unsupported("pooling dimension != hidden_size");
    }
    // precision < 0: the loader's choice -- the bf16 ingest mode, except for CLS-pooled hidden-768 models (bge-base-en), whose scores
    // move by 1e-2 ... 4e-2 on bf16 operands under checkpoint-like weights (north_star: 1e-3): MX_PREC_BF16X3 is the one mode that held
    // the bar on every draw of such weights tried (<= 6e-4; MX_PREC_MIXED reaches 1.1e-3 on two of eleven: profiles/r6_precision_modes_over_seeds.txt)
    if (precision < 0) c.precision = (c.pooling == MX_POOL_CLS && c.hidden == 768) ? MX_PREC_BF16X3 : MX_PREC_BF16;
    pm.max_seq_length = (size_t)std::min(512, c.max_pos - c.pos_offset);
    if (file_exists(dir + "/sentence_bert_config.json")) {
        const Json sb = read_json(dir + "/sentence_bert_config.json");
        const double m = sb.number("max_seq_length", 0);
        if (m > 0) pm.max_seq_length = (size_t)m;
        if (const Json *lc = sb.get("do_lower_case")) pm.do_lower_case = lc->type == Json::Bool ? lc->b : true;
    }
    if (file_exists(dir + "/tokenizer_config.json")) {
        const Json tc = read_json(dir + "/tokenizer_config.json");
        if (const Json *lc = tc.get("do_lower_case")) pm.do_lower_case = lc->type == Json::Bool ? lc->b : true;
    }
    for (const std::string &v : {dir + "/vocab.txt", tdir + "/vocab.txt"})
        if (file_exists(v)) {
            pm.vocab_path = v;
            break;
        }
    for (const std::string &d2 : {dir, tdir})
        if (file_exists(d2 + "/vocab.json") && file_exists(d2 + "/merges.txt")) {
            pm.vocab_json_path = d2 + "/vocab.json";
            pm.merges_path = d2 + "/merges.txt";
            break;
        }
    for (const std::string &d2 : {dir, tdir})
        if (file_exists(d2 + "/tokenizer.json")) {
            pm.tokenizer_json_path = d2 + "/tokenizer.json";
            break;
        }
    if (!file_exists(tdir + "/model.safetensors"))
        throw unsupported(file_exists(tdir + "/pytorch_model.bin") || file_exists(tdir + "/rust_model.ot")
                              ? "only a pickled checkpoint is present (pytorch_model.bin / rust_model.ot): convert it to model.safetensors"
                              : "no model.safetensors");
    const size_t nbytes = mx_encoder_weight_bytes(&c);
    if (nbytes == 0 || c.layers < 1) throw unsupported("config.json does not describe an encoder");
    pm.weights.resize(nbytes / sizeof(float));
    const SafeTensors st(tdir + "/model.safetensors");
    float *w = pm.weights.data();
    const int64_t H = c.hidden, F = c.ffn;
    auto take = [&](const std::string &name, std::vector<int64_t> shape) {
        size_t n = 1;
        for (int64_t d : shape) n *= (size_t)d;
        if ((size_t)(w - pm.weights.data()) + n > pm.weights.size()) throw unsupported("weight blob overrun at " + name);
        if (!st.fetch(name, shape, w)) throw unsupported("weight '" + name + "' missing");
        w += n;
    };
    take("embeddings.word_embeddings.weight", {c.vocab, H});
    take("embeddings.position_embeddings.weight", {c.max_pos, H});
    take("embeddings.token_type_embeddings.weight", {c.type_vocab, H});
    take("embeddings.LayerNorm.weight", {H});
    take("embeddings.LayerNorm.bias", {H});
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = "encoder.layer." + std::to_string(l) + ".";
        for (const char *n : {"query", "key", "value"}) {
            take(p + "attention.self." + n + ".weight", {H, H});
            take(p + "attention.self." + n + ".bias", {H});
        }
        take(p + "attention.output.dense.weight", {H, H});
        take(p + "attention.output.dense.bias", {H});
        take(p + "attention.output.LayerNorm.weight", {H});
        take(p + "attention.output.LayerNorm.bias", {H});
        take(p + "intermediate.dense.weight", {F, H});
        take(p + "intermediate.dense.bias", {F});
        take(p + "output.dense.weight", {H, F});
        take(p + "output.dense.bias", {H});
        take(p + "output.LayerNorm.weight", {H});
        take(p + "output.LayerNorm.bias", {H});
    }
    if ((size_t)(w - pm.weights.data()) != pm.weights.size()) throw unsupported("weight blob size mismatch");
    return pm;
}

// The model's own tokenizer: WordPiece over vocab.txt, byte-level BPE over vocab.json + merges.txt, or what tokenizer.json describes
inline std::shared_ptr<Tokenizer> pretrained_tokenizer(const PretrainedModel &pm) {
    if (!pm.vocab_path.empty()) return Tokenizer::wordpiece(pm.vocab_path, pm.do_lower_case);
    if (!pm.vocab_json_path.empty() && !pm.merges_path.empty()) return Tokenizer::bpe(pm.vocab_json_path, pm.merges_path);
    if (!pm.tokenizer_json_path.empty()) return Tokenizer::from_file(pm.tokenizer_json_path);
    throw EmbeddingError(EmbeddingError::SetupError, "Unable to load model: neither vocab.txt, vocab.json + merges.txt nor tokenizer.json");
}

// SentenceEmbedder::spawn(&ModelConfig) with create_model() reading a LOCAL sentence-transformers directory
// (embedding.rs:84-100): encoder + native tokenizer from the files the reference downloads
inline std::pair<std::thread, std::shared_ptr<SentenceEmbedder>> spawn_pretrained(const std::string &dir, const ModelConfig &mc = ModelConfig{},
                                                                                  int device = 0, int precision = -1 /* the loader's choice */) {
    PretrainedModel pm = load_pretrained_dir(dir, precision);
    std::shared_ptr<Tokenizer> tok = pretrained_tokenizer(pm);
    return SentenceEmbedder::spawn(mc, pm.cfg, std::move(pm.weights), pm.max_seq_length, device, tok);
}

}  // namespace memex
