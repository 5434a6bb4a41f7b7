// tokenizer.cpp -- native BERT WordPiece tokenizer + memex's sliding-window segmenter (host code).
//
// Replaces the `tokenizers 0.14` calls of segment_text (reference
// lib/libmemex/src/llm/embedding.rs:155-198: `Tokenizer::from_pretrained`, `with_truncation(256, 86)`,
// `encode(text, false)`, `get_overflowing()`, `decode(ids, true)`) and the tokenisation rust-bert
// performs inside `model.encode` (embedding.rs:109; [CLS] .. [SEP], truncation to max_seq_length,
// padding to the batch maximum).  The vocabulary is a BERT `vocab.txt` (one token per line).
//
// Pipeline = the HF "bert-base-uncased"-style stack the MiniLM / bge tokenizers use:
//   BertNormalizer (clean text, CJK spacing, NFD accent stripping, lower-casing)
//   -> BertPreTokenizer (whitespace split, every punctuation character its own token)
//   -> WordPiece ("##" continuation, [UNK], max 100 chars per word)
//   -> WordPiece decoder with clean-up (for segment_text's detokenised windows).
// Unicode coverage of the normaliser is deliberately bounded: ASCII, Latin-1 Supplement and
// Latin Extended-A letters are case-folded / accent-stripped by table; other scripts pass through
// unchanged (they still tokenise, but without case folding).  tests/test_tokenizer.py checks ids
// and windows against the `tokenizers` Python package on a synthetic vocabulary.
#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "mx_common.h"

using namespace mx;

struct mx_tokenizer {
    std::vector<std::string> vocab;
    std::unordered_map<std::string, int32_t> index;
    bool lowercase = true;
    int32_t pad = 0, unk = 100, cls = 101, sep = 102, mask = 103;
};

namespace {

// ---- UTF-8 ---------------------------------------------------------------------------------------
std::vector<uint32_t> decode_utf8(const char *s) {
    std::vector<uint32_t> out;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(s);
    while (*p) {
        uint32_t c = *p;
        int n = 0;
        if (c < 0x80) n = 0;
        else if ((c >> 5) == 0x6) { c &= 0x1f; n = 1; }
        else if ((c >> 4) == 0xe) { c &= 0x0f; n = 2; }
        else if ((c >> 3) == 0x1e) { c &= 0x07; n = 3; }
        else { out.push_back(0xfffd); ++p; continue; }
        ++p;
        bool ok = true;
        for (int i = 0; i < n; ++i) {
            if ((*p & 0xc0) != 0x80) { ok = false; break; }
            c = (c << 6) | (*p & 0x3f);
            ++p;
        }
        out.push_back(ok ? c : 0xfffd);
    }
    return out;
}

void append_utf8(std::string &o, uint32_t c) {
    if (c < 0x80) o += (char)c;
    else if (c < 0x800) { o += (char)(0xc0 | (c >> 6)); o += (char)(0x80 | (c & 0x3f)); }
    else if (c < 0x10000) { o += (char)(0xe0 | (c >> 12)); o += (char)(0x80 | ((c >> 6) & 0x3f)); o += (char)(0x80 | (c & 0x3f)); }
    else { o += (char)(0xf0 | (c >> 18)); o += (char)(0x80 | ((c >> 12) & 0x3f)); o += (char)(0x80 | ((c >> 6) & 0x3f)); o += (char)(0x80 | (c & 0x3f)); }
}

// ---- character classes (BertNormalizer / BertPreTokenizer) ----------------------------------------
bool is_whitespace(uint32_t c) {
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == 0xa0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200a) ||
           c == 0x2028 || c == 0x2029 || c == 0x202f || c == 0x205f || c == 0x3000;
}
bool is_control(uint32_t c) {
    if (c == '\t' || c == '\n' || c == '\r') return false;
    return c < 0x20 || (c >= 0x7f && c < 0xa0) || c == 0xad || (c >= 0x200b && c <= 0x200f) || (c >= 0x202a && c <= 0x202e) ||
           (c >= 0x2060 && c <= 0x2064) || c == 0xfeff;
}
bool is_cjk(uint32_t c) {
    return (c >= 0x4e00 && c <= 0x9fff) || (c >= 0x3400 && c <= 0x4dbf) || (c >= 0x20000 && c <= 0x2a6df) ||
           (c >= 0x2a700 && c <= 0x2b73f) || (c >= 0x2b740 && c <= 0x2b81f) || (c >= 0x2b820 && c <= 0x2ceaf) ||
           (c >= 0xf900 && c <= 0xfaff) || (c >= 0x2f800 && c <= 0x2fa1f);
}
bool is_punct(uint32_t c) {
    if ((c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126)) return true;
    if (c == 0xa1 || c == 0xa7 || c == 0xab || c == 0xb6 || c == 0xb7 || c == 0xbb || c == 0xbf) return true;  // Latin-1 P*
    return (c >= 0x2010 && c <= 0x2027) || (c >= 0x2030 && c <= 0x205e) || (c >= 0x3001 && c <= 0x3003) ||
           (c >= 0x3008 && c <= 0x3011) || (c >= 0xff01 && c <= 0xff0f) || (c >= 0xff1a && c <= 0xff20);
}

// Lower-case + NFD accent strip (drop Mn) for U+00C0..U+017F, generated with Python's unicodedata
// (NFD -> remove category Mn -> str.lower()); ASCII is folded arithmetically.
static const uint16_t kFoldLatin[0x180 - 0xC0] = {
    0x0061, 0x0061, 0x0061, 0x0061, 0x0061, 0x0061, 0x00e6, 0x0063, 0x0065, 0x0065, 0x0065, 0x0065,
    0x0069, 0x0069, 0x0069, 0x0069, 0x00f0, 0x006e, 0x006f, 0x006f, 0x006f, 0x006f, 0x006f, 0x00d7,
    0x00f8, 0x0075, 0x0075, 0x0075, 0x0075, 0x0079, 0x00fe, 0x00df, 0x0061, 0x0061, 0x0061, 0x0061,
    0x0061, 0x0061, 0x00e6, 0x0063, 0x0065, 0x0065, 0x0065, 0x0065, 0x0069, 0x0069, 0x0069, 0x0069,
    0x00f0, 0x006e, 0x006f, 0x006f, 0x006f, 0x006f, 0x006f, 0x00f7, 0x00f8, 0x0075, 0x0075, 0x0075,
    0x0075, 0x0079, 0x00fe, 0x0079, 0x0061, 0x0061, 0x0061, 0x0061, 0x0061, 0x0061, 0x0063, 0x0063,
    0x0063, 0x0063, 0x0063, 0x0063, 0x0063, 0x0063, 0x0064, 0x0064, 0x0111, 0x0111, 0x0065, 0x0065,
    0x0065, 0x0065, 0x0065, 0x0065, 0x0065, 0x0065, 0x0065, 0x0065, 0x0067, 0x0067, 0x0067, 0x0067,
    0x0067, 0x0067, 0x0067, 0x0067, 0x0068, 0x0068, 0x0127, 0x0127, 0x0069, 0x0069, 0x0069, 0x0069,
    0x0069, 0x0069, 0x0069, 0x0069, 0x0069, 0x0131, 0x0133, 0x0133, 0x006a, 0x006a, 0x006b, 0x006b,
    0x0138, 0x006c, 0x006c, 0x006c, 0x006c, 0x006c, 0x006c, 0x0140, 0x0140, 0x0142, 0x0142, 0x006e,
    0x006e, 0x006e, 0x006e, 0x006e, 0x006e, 0x0149, 0x014b, 0x014b, 0x006f, 0x006f, 0x006f, 0x006f,
    0x006f, 0x006f, 0x0153, 0x0153, 0x0072, 0x0072, 0x0072, 0x0072, 0x0072, 0x0072, 0x0073, 0x0073,
    0x0073, 0x0073, 0x0073, 0x0073, 0x0073, 0x0073, 0x0074, 0x0074, 0x0074, 0x0074, 0x0167, 0x0167,
    0x0075, 0x0075, 0x0075, 0x0075, 0x0075, 0x0075, 0x0075, 0x0075, 0x0075, 0x0075, 0x0075, 0x0075,
    0x0077, 0x0077, 0x0079, 0x0079, 0x0079, 0x007a, 0x007a, 0x007a, 0x007a, 0x007a, 0x007a, 0x017f};

uint32_t fold_latin(uint32_t c, bool lower) {
    if (!lower) return c;  // cased model: no lower-casing, and strip_accents follows lowercase (None)
    if (c < 0x80) return (c >= 'A' && c <= 'Z') ? c + 32 : c;
    if (c >= 0xC0 && c < 0x180) return kFoldLatin[c - 0xC0];
    return c;
}

std::vector<uint32_t> normalize(const mx_tokenizer *t, const char *text) {
    std::vector<uint32_t> out;
    for (uint32_t c : decode_utf8(text)) {
        if (c == 0 || c == 0xfffd || is_control(c)) continue;          // clean_text
        if (is_whitespace(c)) { out.push_back(' '); continue; }
        if (c >= 0x300 && c <= 0x36f && t->lowercase) continue;         // stray combining marks (Mn)
        if (is_cjk(c)) { out.push_back(' '); out.push_back(c); out.push_back(' '); continue; }
        out.push_back(fold_latin(c, t->lowercase));
    }
    return out;
}

// whitespace split + isolate punctuation
std::vector<std::string> pre_tokenize(const std::vector<uint32_t> &cp) {
    std::vector<std::string> words;
    std::string cur;
    for (uint32_t c : cp) {
        if (c == ' ') {
            if (!cur.empty()) words.push_back(cur), cur.clear();
        } else if (is_punct(c)) {
            if (!cur.empty()) words.push_back(cur), cur.clear();
            std::string p;
            append_utf8(p, c);
            words.push_back(p);
        } else {
            append_utf8(cur, c);
        }
    }
    if (!cur.empty()) words.push_back(cur);
    return words;
}

void wordpiece(const mx_tokenizer *t, const std::string &word, std::vector<int32_t> &ids) {
    // byte offsets of code-point boundaries
    std::vector<size_t> cp;
    for (size_t i = 0; i < word.size(); ++i)
        if ((word[i] & 0xc0) != 0x80) cp.push_back(i);
    if (cp.size() > 100) { ids.push_back(t->unk); return; }  // max_input_chars_per_word
    cp.push_back(word.size());
    std::vector<int32_t> pieces;
    size_t s = 0;  // index into cp
    const size_t n = cp.size() - 1;
    while (s < n) {
        size_t e = n;
        int32_t found = -1;
        for (; e > s; --e) {  // greedy longest match first
            std::string sub = word.substr(cp[s], cp[e] - cp[s]);
            if (s > 0) sub = "##" + sub;
            auto it = t->index.find(sub);
            if (it != t->index.end()) { found = it->second; break; }
        }
        if (found < 0) { ids.push_back(t->unk); return; }  // any failing piece -> whole word is [UNK]
        pieces.push_back(found);
        s = e;
    }
    ids.insert(ids.end(), pieces.begin(), pieces.end());
}

std::vector<int32_t> encode_plain(const mx_tokenizer *t, const char *text) {
    std::vector<int32_t> ids;
    for (const std::string &w : pre_tokenize(normalize(t, text))) wordpiece(t, w, ids);
    return ids;
}

bool is_special(const mx_tokenizer *t, int32_t id) {
    return id == t->pad || id == t->unk || id == t->cls || id == t->sep || id == t->mask;
}

void replace_all(std::string &s, const std::string &a, const std::string &b) {
    size_t p = 0;
    while ((p = s.find(a, p)) != std::string::npos) {
        s.replace(p, a.size(), b);
        p += b.size();
    }
}

// WordPiece decoder (prefix "##", cleanup = true), as tokenizers::decoders::wordpiece
std::string decode_ids(const mx_tokenizer *t, const int32_t *ids, int n, bool skip_special) {
    std::string out;
    bool first = true;
    for (int i = 0; i < n; ++i) {
        const int32_t id = ids[i];
        if (id < 0 || id >= (int32_t)t->vocab.size()) continue;
        if (skip_special && is_special(t, id)) continue;
        std::string tok = t->vocab[id];
        if (!first) {
            if (tok.rfind("##", 0) == 0) tok = tok.substr(2);
            else tok = " " + tok;
        }
        // clean-up runs per token in tokenizers 0.14 (decoders::wordpiece::cleanup)
        replace_all(tok, " .", "."); replace_all(tok, " ?", "?"); replace_all(tok, " !", "!"); replace_all(tok, " ,", ",");
        replace_all(tok, " ' ", "'"); replace_all(tok, " n't", "n't"); replace_all(tok, " 'm", "'m");
        replace_all(tok, " do not", " don't"); replace_all(tok, " 's", "'s"); replace_all(tok, " 've", "'ve");
        replace_all(tok, " 're", "'re");
        out += tok;
        first = false;
    }
    return out;
}

int finish_vocab(mx_tokenizer *t) {
    if (t->vocab.empty()) return fail(MX_EINVAL, "empty vocabulary");
    for (size_t i = 0; i < t->vocab.size(); ++i) t->index.emplace(t->vocab[i], (int32_t)i);
    auto need = [&](const char *tok, int32_t &dst) {
        auto it = t->index.find(tok);
        if (it == t->index.end()) return false;
        dst = it->second;
        return true;
    };
    if (!need("[PAD]", t->pad) || !need("[UNK]", t->unk) || !need("[CLS]", t->cls) || !need("[SEP]", t->sep))
        return fail(MX_EINVAL, "vocabulary lacks [PAD]/[UNK]/[CLS]/[SEP]");
    if (!need("[MASK]", t->mask)) t->mask = -1;
    return MX_OK;
}

int parse_vocab(mx_tokenizer *t, std::istream &in) {
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        t->vocab.push_back(line);
    }
    while (!t->vocab.empty() && t->vocab.back().empty()) t->vocab.pop_back();
    return finish_vocab(t);
}

}  // namespace

extern "C" {

int mx_tokenizer_create(const char *vocab_path, int lowercase, mx_tokenizer **out) {
    if (!vocab_path || !out) return fail(MX_EINVAL, "null argument");
    *out = nullptr;
    std::ifstream f(vocab_path);
    if (!f) return fail(MX_EIO, "Unable to load model <%s>", vocab_path);  // embedding.rs:166-169 wording
    mx_tokenizer *t = new mx_tokenizer();
    t->lowercase = lowercase != 0;
    int rc = parse_vocab(t, f);
    if (rc != MX_OK) { delete t; return rc; }
    *out = t;
    return MX_OK;
}

int mx_tokenizer_create_from_memory(const char *vocab, size_t nbytes, int lowercase, mx_tokenizer **out) {
    if (!vocab || !out) return fail(MX_EINVAL, "null argument");
    *out = nullptr;
    std::istringstream in(std::string(vocab, nbytes));
    mx_tokenizer *t = new mx_tokenizer();
    t->lowercase = lowercase != 0;
    int rc = parse_vocab(t, in);
    if (rc != MX_OK) { delete t; return rc; }
    *out = t;
    return MX_OK;
}

void mx_tokenizer_destroy(mx_tokenizer *t) { delete t; }

int mx_tokenizer_vocab_size(mx_tokenizer *t, int *n) {
    if (!t || !n) return fail(MX_EINVAL, "null argument");
    *n = (int)t->vocab.size();
    return MX_OK;
}

int mx_tokenizer_encode(mx_tokenizer *t, const char *text, int add_special_tokens, int32_t *ids, int cap, int *n) {
    if (!t || !text || !n || (cap > 0 && !ids)) return fail(MX_EINVAL, "null argument");
    std::vector<int32_t> v = encode_plain(t, text);
    if (add_special_tokens) {
        v.insert(v.begin(), t->cls);
        v.push_back(t->sep);
    }
    *n = (int)v.size();  // always the full length: call again with a larger buffer if n > cap
    for (int i = 0; i < (int)v.size() && i < cap; ++i) ids[i] = v[i];
    return MX_OK;
}

int mx_tokenizer_decode(mx_tokenizer *t, const int32_t *ids, int n, int skip_special_tokens, char *out, size_t cap,
                        size_t *nbytes) {
    if (!t || (n > 0 && !ids) || !nbytes) return fail(MX_EINVAL, "null argument");
    const std::string s = decode_ids(t, ids, n, skip_special_tokens != 0);
    *nbytes = s.size() + 1;
    if (out && cap >= s.size() + 1) memcpy(out, s.c_str(), s.size() + 1);
    return MX_OK;
}

int mx_tokenizer_segment(mx_tokenizer *t, const char *text, int max_length, int stride, char *out, size_t cap,
                         size_t *nbytes, int *n_segments) {
    if (!t || !text || !nbytes || !n_segments) return fail(MX_EINVAL, "null argument");
    if (max_length < 1 || stride < 0 || stride >= max_length) return fail(MX_EINVAL, "need 0 <= stride < max_length");
    const std::vector<int32_t> ids = encode_plain(t, text);  // no special tokens (embedding.rs:181)
    std::string buf;
    int nseg = 0;
    const size_t len = ids.size(), offset = (size_t)(max_length - stride);
    if (len == 0) {
        buf.push_back('\0');
        nseg = 1;
    }
    bool end = false;  // tokenizers' Encoding::truncate: windows start every max_length - stride tokens
    for (size_t start = 0; start < len && !end; start += offset) {
        const size_t stop = std::min(start + (size_t)max_length, len);
        end = stop == len;
        std::string seg = decode_ids(t, ids.data() + start, (int)(stop - start), true);
        if (nseg == 0) replace_all(seg, " ' ", "'");  // only the first window (embedding.rs:183 vs :189-194)
        buf += seg;
        buf.push_back('\0');
        ++nseg;
    }
    *nbytes = buf.size();
    *n_segments = nseg;
    if (out && cap >= buf.size()) memcpy(out, buf.data(), buf.size());
    return MX_OK;
}

int mx_tokenizer_encode_batch(mx_tokenizer *t, const char *const *texts, int B, int max_seq_length, int32_t *ids,
                              int s_cap, int32_t *lens, int *S) {
    if (!t || (B > 0 && (!texts || !ids || !lens)) || !S) return fail(MX_EINVAL, "null argument");
    if (max_seq_length < 2) return fail(MX_EINVAL, "max_seq_length must be >= 2");
    for (int b = 0; b < B; ++b)
        if (!texts[b]) return fail(MX_EINVAL, "texts[%d] is null", b);
    std::vector<std::vector<int32_t>> rows((size_t)B);
    // rows are independent and the tokenizer state is read-only: split the batch over host threads
    // (the GPU consumes ~20M tokens/s; one thread produces 1-2M)
    auto work = [&](int b0, int b1) {
        for (int b = b0; b < b1; ++b) {
            std::vector<int32_t> v = encode_plain(t, texts[b]);
            if ((int)v.size() > max_seq_length - 2) v.resize((size_t)max_seq_length - 2);  // truncate, keep room for specials
            v.insert(v.begin(), t->cls);
            v.push_back(t->sep);
            rows[(size_t)b] = std::move(v);
        }
    };
    int nthr = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    nthr = std::min(nthr, B / 16);  // below ~16 texts per thread the spawn cost dominates
    if (nthr <= 1) {
        work(0, B);
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < nthr; ++i) pool.emplace_back(work, (int)((long)B * i / nthr), (int)((long)B * (i + 1) / nthr));
        for (auto &th : pool) th.join();
    }
    int smax = 0;
    for (int b = 0; b < B; ++b) smax = std::max(smax, (int)rows[(size_t)b].size());
    *S = smax;
    if (smax > s_cap) return fail(MX_EINVAL, "row capacity %d < batch maximum %d", s_cap, smax);
    for (int b = 0; b < B; ++b) {
        lens[b] = (int32_t)rows[b].size();
        for (int i = 0; i < s_cap; ++i) ids[(size_t)b * s_cap + i] = i < (int)rows[b].size() ? rows[b][i] : t->pad;
    }
    return MX_OK;
}

}  // extern "C"
