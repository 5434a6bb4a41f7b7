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
// The normaliser and the pre-tokenizer cover all of Unicode through generated tables
// (unicode_tables.inc, scripts/gen_unicode_tables.py: the `tokenizers` package probed per code
// point).  tests/test_tokenizer.py checks ids, decoded text and windows against the `tokenizers`
// Python package on a synthetic vocabulary, including a fuzz over random Unicode strings.
//
// Second kind (round 5): BYTE-LEVEL BPE, the tokenizer of all-distilroberta-v1 -- the third model `segment_text` accepts
// (embedding.rs:159).  Its stack (tokenizer.json of the RoBERTa family): no normaliser; ByteLevel pre-tokenizer (the GPT-2
// regex, add_prefix_space = false; code-point classes from unicode_bpe_tables.inc, probed from the `tokenizers` package by
// scripts/gen_bpe_tables.py); bytes -> 256 printable code points; BPE over `vocab.json` + `merges.txt`; <s> .. </s>; ByteLevel
// decoder with Rust's lossy UTF-8 (a window of segment_text may cut a character in two).  tests/test_tokenizer_bpe.py checks
// ids, decoded text and windows against `tokenizers.ByteLevelBPETokenizer` on a vocabulary trained offline, fuzz included.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "mx_common.h"

using namespace mx;

// Vocabulary lookup of the WordPiece loop: open addressing over polynomial hashes, so that the hash of every candidate
// substring word[s, e) is two multiplications away from the word's prefix hashes (the greedy longest-match loop tries them
// from the longest down) and no candidate is copied.  Keys live in one arena; the first id wins for duplicate strings,
// as with the map this replaces.
struct PieceTable {
    struct Slot { uint64_t h; uint32_t off; uint32_t len; int32_t id; };
    std::vector<Slot> slots;
    std::string arena;
    uint64_t mask = 0;
    size_t max_len = 0;  // longest key in bytes: longer candidates cannot match
    static constexpr uint64_t kBase = 0x100000001b3ull * 31 + 2;  // odd
    static uint64_t hash_bytes(const char *p, size_t n) {
        uint64_t h = 0;
        for (size_t i = 0; i < n; ++i) h = h * kBase + (uint64_t)(unsigned char)p[i] + 1;
        return h;
    }
    static size_t spread(uint64_t h) { return (size_t)((h ^ (h >> 29)) * 0x9e3779b97f4a7c15ull >> 20); }
    void build(const std::vector<std::pair<std::string, int32_t>> &keys) {
        size_t cap = 64;
        while (cap < keys.size() * 2 + 2) cap <<= 1;
        slots.assign(cap, Slot{0, 0, 0, -1});
        mask = cap - 1;
        for (const auto &kv : keys) {
            const uint64_t h = hash_bytes(kv.first.data(), kv.first.size());
            size_t i = spread(h) & mask;
            bool dup = false;
            while (slots[i].id >= 0) {
                if (slots[i].h == h && slots[i].len == kv.first.size() && memcmp(arena.data() + slots[i].off, kv.first.data(), kv.first.size()) == 0) { dup = true; break; }
                i = (i + 1) & mask;
            }
            if (dup) continue;
            slots[i] = Slot{h, (uint32_t)arena.size(), (uint32_t)kv.first.size(), kv.second};
            arena += kv.first;
            max_len = std::max(max_len, kv.first.size());
        }
    }
    int32_t find(uint64_t h, const char *p, size_t n) const {
        for (size_t i = spread(h) & mask;; i = (i + 1) & mask) {
            const Slot &sl = slots[i];
            if (sl.id < 0) return -1;
            if (sl.h == h && sl.len == n && memcmp(arena.data() + sl.off, p, n) == 0) return sl.id;
        }
    }
};

struct mx_tokenizer {
    std::vector<std::string> vocab;
    std::unordered_map<std::string, int32_t> index;
    bool lowercase = true;
    int32_t pad = 0, unk = 100, cls = 101, sep = 102, mask = 103;
    // WordPiece fast path: every token by its full string (a word's first piece) / the "##x" tokens by x (continuations);
    // the decoder's per-token strings, cleaned up once: [2 id] = as the first token of a text, [2 id + 1] = as a later one
    PieceTable full, cont;
    std::vector<uint32_t> dec_off;
    std::string dec_arena;
    // byte-level BPE (kind 1): merge ranks by "left right", the byte <-> code point tables of the ByteLevel stage
    int kind = 0;
    std::unordered_map<std::string, int32_t> merges;
    // ... the same merges over token ids (a << 32 | b -> rank, id of the merged token) when every merge's operands and product
    // are vocabulary entries -- the case for every trained byte-level BPE; the string form stays for files that break that rule
    std::unordered_map<uint64_t, std::pair<int32_t, int32_t>> pair_rank;
    int32_t byte_id[256];
    bool bpe_ids = false;
    std::string byte_chr[256];                      // byte -> its printable code point, UTF-8 encoded
    std::unordered_map<uint32_t, uint8_t> chr_byte;  // code point -> byte
    // added tokens: strings the `tokenizers` crate cuts out of the RAW text before anything else (literal "[SEP]" / "<s>" /
    // "<mask>" in a document become that token's id, and decode(skip_special_tokens) then drops them): tokenizer.json's
    // `added_tokens`, or the model family's five specials (finish_vocab / finish_bpe).  lstrip / rstrip: the match swallows the
    // white space on that side (RoBERTa's <mask> has lstrip)
    struct Added {
        std::string s;
        int32_t id;
        bool lstrip, rstrip, special;
    };
    std::vector<Added> added;
    bool added_from_json = false;
};

namespace {

// ---- UTF-8 ---------------------------------------------------------------------------------------
// one code point from a NUL-terminated string (0xfffd for a malformed sequence: an invalid lead byte is consumed alone, a
// sequence cut short ends in front of the byte that does not continue it)
inline uint32_t next_cp(const unsigned char *&p) {
    uint32_t c = *p;
    int n = 0;
    if (c < 0x80) n = 0;
    else if ((c >> 5) == 0x6) { c &= 0x1f; n = 1; }
    else if ((c >> 4) == 0xe) { c &= 0x0f; n = 2; }
    else if ((c >> 3) == 0x1e) { c &= 0x07; n = 3; }
    else { ++p; return 0xfffd; }
    ++p;
    for (int i = 0; i < n; ++i) {
        if ((*p & 0xc0) != 0x80) return 0xfffd;
        c = (c << 6) | (*p & 0x3f);
        ++p;
    }
    return c;
}

std::vector<uint32_t> decode_utf8(const char *s) {
    std::vector<uint32_t> out;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(s);
    while (*p) out.push_back(next_cp(p));
    return out;
}

void append_utf8(std::string &o, uint32_t c) {
    if (c < 0x80) o += (char)c;
    else if (c < 0x800) { o += (char)(0xc0 | (c >> 6)); o += (char)(0x80 | (c & 0x3f)); }
    else if (c < 0x10000) { o += (char)(0xe0 | (c >> 12)); o += (char)(0x80 | ((c >> 6) & 0x3f)); o += (char)(0x80 | (c & 0x3f)); }
    else { o += (char)(0xf0 | (c >> 18)); o += (char)(0x80 | ((c >> 12) & 0x3f)); o += (char)(0x80 | ((c >> 6) & 0x3f)); o += (char)(0x80 | (c & 0x3f)); }
}

// ---- character classes and folding (BertNormalizer / BertPreTokenizer) ----------------------------
// unicode_tables.inc is generated by scripts/gen_unicode_tables.py, which probes the `tokenizers`
// package (the crate the reference calls) for every code point: what clean_text drops or turns into a
// space, what NFD + strip-Mn + lower-casing maps each character to, and which characters the
// pre-tokenizer treats as whitespace / isolates as punctuation.
#include "unicode_tables.inc"

template <size_t N>
bool in_ranges(const uint32_t (&r)[N][2], uint32_t c) {
    size_t lo = 0, hi = N;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (c > r[mid][1]) lo = mid + 1;
        else hi = mid;
    }
    return lo < N && c >= r[lo][0];
}

bool is_whitespace(uint32_t c) { return c < 0x80 ? (c == ' ' || c == '\t' || c == '\n' || c == '\r') : in_ranges(kSpaceRanges, c); }
bool is_dropped(uint32_t c) { return c < 0x80 ? (c < 0x20 && c != '\t' && c != '\n' && c != '\r') || c == 0x7f : in_ranges(kDropRanges, c); }
bool is_punct(uint32_t c) {
    if (c < 0x80) return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
    return in_ranges(kPunctRanges, c);
}
bool is_cjk(uint32_t c) {
    return (c >= 0x4e00 && c <= 0x9fff) || (c >= 0x3400 && c <= 0x4dbf) || (c >= 0x20000 && c <= 0x2a6df) ||
           (c >= 0x2a700 && c <= 0x2b73f) || (c >= 0x2b740 && c <= 0x2b81f) || (c >= 0x2b820 && c <= 0x2ceaf) ||
           (c >= 0xf900 && c <= 0xfaff) || (c >= 0x2f800 && c <= 0x2fa1f);
}

// NFD -> drop Mn -> lower-case of one code point, appended to out (uncased models only)
void fold_append(std::vector<uint32_t> &out, uint32_t c) {
    if (c < 0x80) {
        out.push_back((c >= 'A' && c <= 'Z') ? c + 32 : c);
        return;
    }
    if (c >= 0xac00 && c <= 0xd7a3) {  // Hangul syllable: arithmetic canonical decomposition into jamo
        const uint32_t si = c - 0xac00;
        out.push_back(0x1100 + si / 588);
        out.push_back(0x1161 + (si % 588) / 28);
        if (si % 28) out.push_back(0x11a7 + si % 28);
        return;
    }
    constexpr size_t n = sizeof(kFoldIndex) / sizeof(kFoldIndex[0]);
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (kFoldIndex[mid][0] < c) lo = mid + 1;
        else hi = mid;
    }
    if (lo < n && kFoldIndex[lo][0] == c) {
        for (uint32_t i = 0; i < kFoldIndex[lo][2]; ++i) out.push_back(kFoldData[kFoldIndex[lo][1] + i]);
        return;
    }
    out.push_back(c);
}

// BertNormalizer::normalize: clean_text -> handle_chinese_chars -> strip_accents -> lowercase
// (strip_accents = None follows lowercase, so a cased model only gets the first two steps)
std::vector<uint32_t> normalize(const mx_tokenizer *t, const char *text) {
    std::vector<uint32_t> out;
    for (uint32_t c : decode_utf8(text)) {
        if (c == 0 || c == 0xfffd || is_dropped(c)) continue;          // clean_text
        if (is_whitespace(c)) { out.push_back(' '); continue; }
        if (is_cjk(c)) { out.push_back(' '); out.push_back(c); out.push_back(' '); continue; }
        if (t->lowercase) fold_append(out, c);
        else out.push_back(c);
    }
    return out;
}

// whitespace split + isolate punctuation
std::vector<std::string> pre_tokenize(const std::vector<uint32_t> &cp) {
    std::vector<std::string> words;
    std::string cur;
    for (uint32_t c : cp) {
        if (c == ' ' || is_whitespace(c)) {  // folding can yield characters the pre-tokenizer splits on
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

std::vector<int32_t> bpe_encode_plain(const mx_tokenizer *t, const char *text);
std::string bpe_decode_ids(const mx_tokenizer *t, const int32_t *ids, int n, bool skip_special);

// ---- the WordPiece path as it runs: one pass over the bytes, no per-word or per-candidate allocation -------------------------
// (normalize / pre_tokenize / wordpiece above are the same steps spelled out stage by stage: mx_tokenizer_encode_staged;
// tests/test_tokenizer.py holds the two against each other and this one against the `tokenizers` package)
struct PowTable {
    uint64_t p[401];
    PowTable() {
        p[0] = 1;
        for (int i = 1; i <= 400; ++i) p[i] = p[i - 1] * PieceTable::kBase;
    }
};
const PowTable kPow;

// one pre-tokenised word (<= 100 code points, else [UNK]) -> its pieces, greedy longest match first
void wordpiece_fast(const mx_tokenizer *t, const char *w, size_t wn, std::vector<int32_t> &ids) {
    uint16_t cpb[102];  // byte offsets of the code-point boundaries
    size_t n = 0;
    for (size_t i = 0; i < wn; ++i)
        if ((w[i] & 0xc0) != 0x80) {
            if (n == 100) { ids.push_back(t->unk); return; }  // max_input_chars_per_word
            cpb[n++] = (uint16_t)i;
        }
    cpb[n] = (uint16_t)wn;
    uint64_t H[401];  // prefix hashes: hash(w[a, b)) = H[b] - H[a] * base^(b - a)
    H[0] = 0;
    for (size_t i = 0; i < wn; ++i) H[i + 1] = H[i] * PieceTable::kBase + (uint64_t)(unsigned char)w[i] + 1;
    const size_t start = ids.size();
    size_t s = 0;
    while (s < n) {
        const PieceTable &tb = s ? t->cont : t->full;
        int32_t found = -1;
        size_t e = n;
        for (; e > s; --e) {
            const size_t len = (size_t)cpb[e] - cpb[s];
            if (len > tb.max_len) continue;
            found = tb.find(H[cpb[e]] - H[cpb[s]] * kPow.p[len], w + cpb[s], len);
            if (found >= 0) break;
        }
        if (found < 0) {  // any failing piece -> the whole word is [UNK]
            ids.resize(start);
            ids.push_back(t->unk);
            return;
        }
        ids.push_back(found);
        s = e;
    }
}

void encode_wordpiece(const mx_tokenizer *t, const char *text, std::vector<int32_t> &ids) {
    std::string cur;
    cur.reserve(64);
    std::vector<uint32_t> folded;
    auto flush = [&] {
        if (!cur.empty()) {
            wordpiece_fast(t, cur.data(), cur.size(), ids);
            cur.clear();
        }
    };
    auto emit = [&](uint32_t c) {  // a normalised code point through the pre-tokenizer
        if (c == ' ' || is_whitespace(c)) {
            flush();
        } else if (is_punct(c)) {
            flush();
            append_utf8(cur, c);
            flush();
        } else {
            append_utf8(cur, c);
        }
    };
    const bool lower = t->lowercase;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(text);
    while (*p) {
        if (*p < 0x80) {  // ASCII: clean_text, lower-casing and the pre-tokenizer's classes inline
            const unsigned char c = *p++;
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { flush(); continue; }
            if (c < 0x20 || c == 0x7f) continue;
            if ((c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126)) {
                flush();
                const char b = (char)c;
                wordpiece_fast(t, &b, 1, ids);
                continue;
            }
            cur.push_back((char)((lower && c >= 'A' && c <= 'Z') ? c + 32 : c));
            continue;
        }
        const uint32_t c = next_cp(p);
        if (c == 0xfffd || is_dropped(c)) continue;
        if (is_whitespace(c)) { flush(); continue; }
        if (is_cjk(c)) {
            flush();
            emit(c);
            flush();
            continue;
        }
        if (lower) {
            folded.clear();
            fold_append(folded, c);
            for (uint32_t x : folded) emit(x);
        } else {
            emit(c);
        }
    }
    flush();
}

std::vector<int32_t> encode_piece(const mx_tokenizer *t, const char *text) {
    if (t->kind == 1) return bpe_encode_plain(t, text);
    std::vector<int32_t> ids;
    encode_wordpiece(t, text, ids);
    return ids;
}

inline bool ascii_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// the raw text cut at its added tokens (leftmost match, the longest token at a position -- the crate's Aho-Corasick
// "leftmost-longest" over the un-normalised text), the pieces between them through the model's own pipeline
std::vector<int32_t> encode_plain(const mx_tokenizer *t, const char *text) {
    if (t->added.empty()) return encode_piece(t, text);
    // (every added token of the supported families starts with '[' or '<': a text without either has none)
    const size_t n = strlen(text);
    std::vector<int32_t> ids;
    size_t piece0 = 0, i = 0;
    std::string piece;
    auto flush = [&](size_t end) {
        if (end > piece0) {
            piece.assign(text + piece0, end - piece0);
            const std::vector<int32_t> v = encode_piece(t, piece.c_str());
            ids.insert(ids.end(), v.begin(), v.end());
        }
    };
    while (i < n) {
        const mx_tokenizer::Added *hit = nullptr;
        for (const auto &a : t->added)
            if (a.s.size() <= n - i && text[i] == a.s[0] && memcmp(text + i, a.s.data(), a.s.size()) == 0 && (!hit || a.s.size() > hit->s.size())) hit = &a;
        if (!hit) {
            ++i;
            continue;
        }
        size_t b = i, e = i + hit->s.size();
        if (hit->lstrip)
            while (b > piece0 && ascii_space(text[b - 1])) --b;
        if (hit->rstrip)
            while (e < n && ascii_space(text[e])) ++e;
        flush(b);
        ids.push_back(hit->id);
        piece0 = i = e;
    }
    flush(n);
    return ids;
}

// the stage-by-stage form (tests only: mx_tokenizer_encode_staged)
std::vector<int32_t> encode_staged(const mx_tokenizer *t, const char *text) {
    std::vector<int32_t> ids;
    for (const std::string &w : pre_tokenize(normalize(t, text))) wordpiece(t, w, ids);
    return ids;
}

bool is_special(const mx_tokenizer *t, int32_t id) {
    if (id == t->pad || id == t->unk || id == t->cls || id == t->sep || id == t->mask) return true;
    if (t->added_from_json)
        for (const auto &a : t->added)
            if (a.special && a.id == id) return true;
    return false;
}

void replace_all(std::string &s, const std::string &a, const std::string &b) {
    size_t p = 0;
    while ((p = s.find(a, p)) != std::string::npos) {
        s.replace(p, a.size(), b);
        p += b.size();
    }
}

// WordPiece decoder (prefix "##", cleanup = true), as tokenizers::decoders::wordpiece
// what token `id` contributes to a decoded text as its first token / as a later one
std::string decoded_token(const mx_tokenizer *t, int32_t id, bool first) {
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
    return tok;
}

void decode_append(const mx_tokenizer *t, const int32_t *ids, int n, bool skip_special, std::string &out) {
    bool first = true;
    for (int i = 0; i < n; ++i) {
        const int32_t id = ids[i];
        if (id < 0 || id >= (int32_t)t->vocab.size()) continue;
        if (skip_special && is_special(t, id)) continue;
        const size_t e = 2 * (size_t)id + (first ? 0 : 1);  // the clean-up is per token: both forms were prepared once (finish_vocab)
        out.append(t->dec_arena.data() + t->dec_off[e], t->dec_off[e + 1] - t->dec_off[e]);
        first = false;
    }
}

std::string decode_ids(const mx_tokenizer *t, const int32_t *ids, int n, bool skip_special) {
    if (t->kind == 1) return bpe_decode_ids(t, ids, n, skip_special);
    std::string out;
    decode_append(t, ids, n, skip_special, out);
    return out;
}

// ---- byte-level BPE (RoBERTa / GPT-2 family) -----------------------------------------------------------------------------
#include "unicode_bpe_tables.inc"

bool bpe_letter(uint32_t c) { return c < 0x80 ? ((c | 0x20) >= 'a' && (c | 0x20) <= 'z') : in_ranges(kBpeLetterRanges, c); }
bool bpe_number(uint32_t c) { return c < 0x80 ? (c >= '0' && c <= '9') : in_ranges(kBpeNumberRanges, c); }
bool bpe_space(uint32_t c) { return c < 0x80 ? (c == ' ' || (c >= 9 && c <= 13)) : in_ranges(kBpeSpaceRanges, c); }
bool bpe_other(uint32_t c) { return !bpe_space(c) && !bpe_letter(c) && !bpe_number(c); }

// strict UTF-8 -> code points; an ill-formed sequence yields U+FFFD per maximal subpart (Rust's from_utf8_lossy, which the
// crate's ByteLevel decoder applies and whose output re-enters the tokenizer when a decoded window is embedded)
std::vector<uint32_t> decode_utf8_lossy(const unsigned char *p, size_t n) {
    std::vector<uint32_t> out;
    size_t i = 0;
    while (i < n) {
        const unsigned char b = p[i];
        if (b < 0x80) { out.push_back(b); ++i; continue; }
        int need = 0;
        unsigned char lo = 0x80, hi = 0xbf;
        uint32_t c = 0;
        if (b >= 0xc2 && b <= 0xdf) { need = 1; c = b & 0x1f; }
        else if (b >= 0xe0 && b <= 0xef) { need = 2; c = b & 0x0f; if (b == 0xe0) lo = 0xa0; if (b == 0xed) hi = 0x9f; }
        else if (b >= 0xf0 && b <= 0xf4) { need = 3; c = b & 0x07; if (b == 0xf0) lo = 0x90; if (b == 0xf4) hi = 0x8f; }
        else { out.push_back(0xfffd); ++i; continue; }
        size_t j = i + 1;
        bool ok = true;
        for (int k = 0; k < need; ++k, ++j) {
            if (j >= n || p[j] < lo || p[j] > hi) { ok = false; break; }
            c = (c << 6) | (p[j] & 0x3f);
            lo = 0x80; hi = 0xbf;
        }
        out.push_back(ok ? c : 0xfffd);
        i = j;  // (on failure: the maximal well-formed prefix is replaced as a whole, the offending byte is read again)
    }
    return out;
}

// the GPT-2 pre-tokenizer regex as a hand-written leftmost-first matcher over code points:
//   's|'t|'re|'ve|'m|'ll|'d| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+
std::vector<std::pair<size_t, size_t>> bpe_pre_tokenize(const std::vector<uint32_t> &cp) {
    std::vector<std::pair<size_t, size_t>> out;  // [begin, end) in code points
    const size_t n = cp.size();
    size_t i = 0;
    auto run = [&](size_t from, bool (*cls)(uint32_t)) {
        size_t e = from;
        while (e < n && cls(cp[e])) ++e;
        return e;
    };
    while (i < n) {
        size_t e = i;
        if (cp[i] == '\'' && i + 1 < n) {
            const uint32_t a = cp[i + 1], b = i + 2 < n ? cp[i + 2] : 0;
            if (a == 's' || a == 't' || a == 'm' || a == 'd') e = i + 2;
            if (e == i && ((a == 'r' && b == 'e') || (a == 'v' && b == 'e') || (a == 'l' && b == 'l'))) e = i + 3;
            // (alternation order 's 't 're 've 'm 'll 'd: the one-letter forms never shadow a two-letter one)
        }
        if (e == i) {
            const size_t j = cp[i] == ' ' ? i + 1 : i;  // the optional literal space
            for (auto cls : {&bpe_letter, &bpe_number, &bpe_other}) {
                if (j < n && cls(cp[j])) { e = run(j, cls); break; }
                // without the optional space the class must match at i itself: a space is in none of the three
            }
        }
        if (e == i) {  // whitespace: \s+(?!\S) -- all of a run that ends the text, else all but its last character -- then \s+
            const size_t r = run(i, &bpe_space);
            if (r == n || r - i == 1) e = r;  // (a single whitespace before a non-space: \s+(?!\S) fails, \s+ takes it)
            else e = r - 1;
        }
        if (e == i) e = i + 1;  // unreachable: every code point is in one of the four classes
        out.emplace_back(i, e);
        i = e;
    }
    return out;
}

void bpe_word(const mx_tokenizer *t, const std::string &bytes, std::vector<int32_t> &ids) {
    std::vector<std::string> sym;
    sym.reserve(bytes.size());
    for (unsigned char b : bytes) sym.push_back(t->byte_chr[b]);
    while (sym.size() > 1) {
        int32_t best = INT32_MAX;
        size_t at = 0;
        for (size_t k = 0; k + 1 < sym.size(); ++k) {
            auto it = t->merges.find(sym[k] + " " + sym[k + 1]);
            if (it != t->merges.end() && it->second < best) { best = it->second; at = k; }
        }
        if (best == INT32_MAX) break;
        // merge every occurrence of the best pair, left to right (what the crate's rank-ordered queue amounts to)
        const std::string a = sym[at], b = sym[at + 1];
        std::vector<std::string> next;
        next.reserve(sym.size());
        for (size_t k = 0; k < sym.size();) {
            if (k + 1 < sym.size() && sym[k] == a && sym[k + 1] == b) { next.push_back(a + b); k += 2; }
            else { next.push_back(sym[k]); ++k; }
        }
        sym.swap(next);
    }
    for (const std::string &s2 : sym) {
        auto it = t->index.find(s2);
        if (it != t->index.end()) ids.push_back(it->second);
        else if (t->unk >= 0) ids.push_back(t->unk);  // (a byte-level vocabulary holds all 256 base symbols: not reached)
    }
}

// bpe_word over token ids: the lowest-ranked adjacent pair, every occurrence of it merged left to right, until none is left
void bpe_word_ids(const mx_tokenizer *t, const std::string &bytes, std::vector<int32_t> &ids) {
    std::vector<int32_t> sym(bytes.size()), next;
    for (size_t k = 0; k < bytes.size(); ++k) sym[k] = t->byte_id[(unsigned char)bytes[k]];
    while (sym.size() > 1) {
        int32_t best = INT32_MAX, merged = -1, a = 0, b = 0;
        for (size_t k = 0; k + 1 < sym.size(); ++k) {
            auto it = t->pair_rank.find((uint64_t)(uint32_t)sym[k] << 32 | (uint32_t)sym[k + 1]);
            if (it != t->pair_rank.end() && it->second.first < best) {
                best = it->second.first;
                merged = it->second.second;
                a = sym[k];
                b = sym[k + 1];
            }
        }
        if (best == INT32_MAX) break;
        next.clear();
        for (size_t k = 0; k < sym.size();) {
            if (k + 1 < sym.size() && sym[k] == a && sym[k + 1] == b) { next.push_back(merged); k += 2; }
            else { next.push_back(sym[k]); ++k; }
        }
        sym.swap(next);
    }
    ids.insert(ids.end(), sym.begin(), sym.end());
}

std::vector<int32_t> bpe_encode_plain(const mx_tokenizer *t, const char *text) {
    const size_t nb = strlen(text);
    const std::vector<uint32_t> cp = decode_utf8_lossy(reinterpret_cast<const unsigned char *>(text), nb);
    std::vector<int32_t> ids;
    // words repeat (Zipf): each distinct pre-token of this text is merged once (the cache lives for the call: no shared state
    // between the host threads of a batch)
    std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> seen;  // word -> [begin, end) in `pieces`
    std::vector<int32_t> pieces;
    std::string bytes;
    for (const auto &pc : bpe_pre_tokenize(cp)) {
        bytes.clear();
        for (size_t k = pc.first; k < pc.second; ++k) append_utf8(bytes, cp[k]);
        auto it = seen.find(bytes);
        if (it == seen.end()) {
            const uint32_t b0 = (uint32_t)pieces.size();
            if (t->bpe_ids) bpe_word_ids(t, bytes, pieces);
            else bpe_word(t, bytes, pieces);
            it = seen.emplace(bytes, std::make_pair(b0, (uint32_t)pieces.size())).first;
        }
        ids.insert(ids.end(), pieces.begin() + it->second.first, pieces.begin() + it->second.second);
    }
    return ids;
}

// ByteLevel decoder: tokens -> bytes (a token with a character outside the byte alphabet contributes its own UTF-8) -> lossy UTF-8
std::string bpe_decode_ids(const mx_tokenizer *t, const int32_t *ids, int n, bool skip_special) {
    std::string bytes;
    for (int i = 0; i < n; ++i) {
        const int32_t id = ids[i];
        if (id < 0 || id >= (int32_t)t->vocab.size()) continue;
        if (skip_special && is_special(t, id)) continue;
        bytes.append(t->dec_arena.data() + t->dec_off[(size_t)id], t->dec_off[(size_t)id + 1] - t->dec_off[(size_t)id]);
    }
    std::string out;
    for (uint32_t c : decode_utf8_lossy(reinterpret_cast<const unsigned char *>(bytes.data()), bytes.size())) append_utf8(out, c);
    return out;
}

// {"token": id, ...} (vocab.json): strings with JSON escapes, integer values
int parse_vocab_json(mx_tokenizer *t, const std::string &js) {
    size_t i = 0;
    const size_t n = js.size();
    auto ws = [&] { while (i < n && (js[i] == ' ' || js[i] == '\n' || js[i] == '\r' || js[i] == '\t')) ++i; };
    ws();
    if (i >= n || js[i] != '{') return fail(MX_EINVAL, "vocab.json: expected an object");
    ++i;
    std::vector<std::pair<std::string, int64_t>> items;
    int64_t max_id = -1;
    for (;;) {
        ws();
        if (i < n && js[i] == '}') break;
        if (i >= n || js[i] != '"') return fail(MX_EINVAL, "vocab.json: expected a string at byte %zu", i);
        ++i;
        std::string key;
        while (i < n && js[i] != '"') {
            if (js[i] == '\\' && i + 1 < n) {
                const char e = js[i + 1];
                i += 2;
                if (e == 'u') {
                    if (i + 4 > n) return fail(MX_EINVAL, "vocab.json: bad \\u escape");
                    uint32_t c = (uint32_t)strtoul(js.substr(i, 4).c_str(), nullptr, 16);
                    i += 4;
                    if (c >= 0xd800 && c <= 0xdbff && i + 6 <= n && js[i] == '\\' && js[i + 1] == 'u') {  // surrogate pair
                        const uint32_t lo = (uint32_t)strtoul(js.substr(i + 2, 4).c_str(), nullptr, 16);
                        if (lo >= 0xdc00 && lo <= 0xdfff) { c = 0x10000 + ((c - 0xd800) << 10) + (lo - 0xdc00); i += 6; }
                    }
                    append_utf8(key, c);
                } else {
                    key += e == 'n' ? '\n' : e == 't' ? '\t' : e == 'r' ? '\r' : e == 'b' ? '\b' : e == 'f' ? '\f' : e;
                }
            } else {
                key += js[i++];
            }
        }
        if (i >= n) return fail(MX_EINVAL, "vocab.json: unterminated string");
        ++i;
        ws();
        if (i >= n || js[i] != ':') return fail(MX_EINVAL, "vocab.json: expected ':'");
        ++i;
        ws();
        char *endp = nullptr;
        const long long v = strtoll(js.c_str() + i, &endp, 10);
        if (endp == js.c_str() + i || v < 0 || v > 10000000) return fail(MX_EINVAL, "vocab.json: bad id for '%s'", key.c_str());
        i = (size_t)(endp - js.c_str());
        items.emplace_back(std::move(key), v);
        max_id = std::max<int64_t>(max_id, v);
        ws();
        if (i < n && js[i] == ',') { ++i; continue; }
        ws();
        if (i < n && js[i] == '}') break;
        return fail(MX_EINVAL, "vocab.json: expected ',' or '}' at byte %zu", i);
    }
    if (items.empty()) return fail(MX_EINVAL, "empty vocabulary");
    t->vocab.assign((size_t)max_id + 1, std::string());
    for (auto &kv : items) {
        t->vocab[(size_t)kv.second] = kv.first;
        t->index.emplace(kv.first, (int32_t)kv.second);
    }
    return MX_OK;
}

int finish_bpe(mx_tokenizer *t, std::istream &merges) {
    t->kind = 1;
    t->lowercase = false;
    // bytes_to_unicode of GPT-2: printable Latin-1 bytes map to themselves, the other 68 to U+0100 ..
    int extra = 0;
    for (int b = 0; b < 256; ++b) {
        const bool keep = (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255);
        const uint32_t c = keep ? (uint32_t)b : 256u + (uint32_t)extra++;
        append_utf8(t->byte_chr[b], c);
        t->chr_byte.emplace(c, (uint8_t)b);
    }
    std::string line;
    int32_t rank = 0;
    while (std::getline(merges, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line.rfind("#version", 0) == 0) continue;
        if (line.find(' ') == std::string::npos) return fail(MX_EINVAL, "merges.txt: line without a pair: '%s'", line.c_str());
        t->merges.emplace(line, rank++);
    }
    auto need = [&](const char *tok, int32_t &dst) {
        auto it = t->index.find(tok);
        dst = it == t->index.end() ? -1 : it->second;
        return dst >= 0;
    };
    if (!need("<s>", t->cls) || !need("</s>", t->sep) || !need("<pad>", t->pad))
        return fail(MX_EINVAL, "vocabulary lacks <s> / </s> / <pad>");
    need("<unk>", t->unk);
    need("<mask>", t->mask);
    // the ByteLevel decoder's bytes of every token, once (a token with a character outside the byte alphabet stands for its own UTF-8)
    t->dec_off.assign(1, 0u);
    for (const std::string &tok : t->vocab) {
        std::string mapped;
        bool ok = true;
        for (uint32_t c : decode_utf8_lossy(reinterpret_cast<const unsigned char *>(tok.data()), tok.size())) {
            auto it = t->chr_byte.find(c);
            if (it == t->chr_byte.end()) { ok = false; break; }
            mapped.push_back((char)it->second);
        }
        t->dec_arena += ok ? mapped : tok;
        t->dec_off.push_back((uint32_t)t->dec_arena.size());
    }
    t->bpe_ids = true;
    for (int b = 0; b < 256 && t->bpe_ids; ++b) {
        auto it = t->index.find(t->byte_chr[b]);
        if (it == t->index.end()) t->bpe_ids = false;
        else t->byte_id[b] = it->second;
    }
    for (const auto &m : t->merges) {
        if (!t->bpe_ids) break;
        const size_t sp = m.first.find(' ');
        auto a = t->index.find(m.first.substr(0, sp)), b = t->index.find(m.first.substr(sp + 1));
        auto ab = t->index.find(m.first.substr(0, sp) + m.first.substr(sp + 1));
        if (a == t->index.end() || b == t->index.end() || ab == t->index.end() || m.first.find(' ', sp + 1) != std::string::npos) {
            t->bpe_ids = false;
            break;
        }
        t->pair_rank.emplace((uint64_t)(uint32_t)a->second << 32 | (uint32_t)b->second, std::make_pair(m.second, ab->second));
    }
    // two different strings must not share an id pair's product by accident: the id form is only used when ids and strings are
    // in bijection for everything a merge can produce (vocab.json maps distinct strings to distinct ids by construction)
    if (!t->bpe_ids) t->pair_rank.clear();
    if (!t->added_from_json) {  // what tokenizer.json of the RoBERTa family lists as added_tokens (<mask> swallows the space before it)
        t->added.clear();
        for (const auto &sp : {std::make_pair("<s>", t->cls), std::make_pair("<pad>", t->pad), std::make_pair("</s>", t->sep), std::make_pair("<unk>", t->unk)})
            if (sp.second >= 0 && t->index.count(sp.first)) t->added.push_back({sp.first, sp.second, false, false, true});
        if (t->index.count("<mask>")) t->added.push_back({"<mask>", t->mask, true, false, true});
    }
    return MX_OK;
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
    if (!t->added_from_json) {  // what tokenizer.json of the BERT family lists as added_tokens
        t->added.clear();
        for (const auto &sp : {std::make_pair("[PAD]", t->pad), std::make_pair("[UNK]", t->unk), std::make_pair("[CLS]", t->cls),
                               std::make_pair("[SEP]", t->sep), std::make_pair("[MASK]", t->mask)})
            if (sp.second >= 0) t->added.push_back({sp.first, sp.second, false, false, true});
    }
    std::vector<std::pair<std::string, int32_t>> all, cont;
    for (size_t i = 0; i < t->vocab.size(); ++i) {
        all.emplace_back(t->vocab[i], (int32_t)i);
        if (t->vocab[i].size() > 2 && t->vocab[i].compare(0, 2, "##") == 0) cont.emplace_back(t->vocab[i].substr(2), (int32_t)i);
    }
    t->full.build(all);
    t->cont.build(cont);
    t->dec_off.assign(1, 0u);
    for (size_t i = 0; i < t->vocab.size(); ++i)
        for (int later = 0; later < 2; ++later) {
            t->dec_arena += decoded_token(t, (int32_t)i, later == 0);
            t->dec_off.push_back((uint32_t)t->dec_arena.size());
        }
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

// ---- tokenizer.json (what Tokenizer::from_pretrained fetches, embedding.rs:163): a span scanner, no tree --------------------
// A value is the byte range [b, e) of the source text.  Only what the loader reads is interpreted; everything else is skipped
// with bracket / string matching (depth-limited like the model-directory loaders).
struct JSpan { size_t b = 0, e = 0; bool ok() const { return e > b; } };

struct JScan {
    const std::string &s;
    explicit JScan(const std::string &src) : s(src) {}
    void ws(size_t &i) const { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\r' || s[i] == '\t')) ++i; }
    // i at the opening quote -> one past the closing quote (false: unterminated)
    bool skip_string(size_t &i) const {
        for (++i; i < s.size(); ++i) {
            if (s[i] == '\\') { ++i; continue; }
            if (s[i] == '"') { ++i; return true; }
        }
        return false;
    }
    bool skip_value(size_t &i, int depth = 0) const {
        ws(i);
        if (i >= s.size() || depth > 64) return false;
        if (s[i] == '"') return skip_string(i);
        if (s[i] == '{' || s[i] == '[') {
            const char close = s[i] == '{' ? '}' : ']';
            const bool obj = s[i] == '{';
            ++i;
            ws(i);
            if (i < s.size() && s[i] == close) { ++i; return true; }
            for (;;) {
                if (obj) {
                    ws(i);
                    if (i >= s.size() || s[i] != '"' || !skip_string(i)) return false;
                    ws(i);
                    if (i >= s.size() || s[i] != ':') return false;
                    ++i;
                }
                if (!skip_value(i, depth + 1)) return false;
                ws(i);
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == close) { ++i; return true; }
                return false;
            }
        }
        const size_t b = i;  // number / true / false / null
        while (i < s.size() && s[i] != ',' && s[i] != '}' && s[i] != ']' && s[i] != ' ' && s[i] != '\n' && s[i] != '\r' && s[i] != '\t') ++i;
        return i > b;
    }
    // the value of `key` in the object `o` (span of "{...}"); empty span when absent or o is not an object
    JSpan find(JSpan o, const char *key) const {
        size_t i = o.b;
        ws(i);
        if (i >= o.e || s[i] != '{') return {};
        ++i;
        const size_t klen = strlen(key);
        for (;;) {
            ws(i);
            if (i >= o.e || s[i] != '"') return {};
            const size_t kb = i + 1;
            if (!skip_string(i)) return {};
            const bool hit = i - 1 - kb == klen && s.compare(kb, klen, key) == 0;
            ws(i);
            if (i >= o.e || s[i] != ':') return {};
            ++i;
            ws(i);
            JSpan v;
            v.b = i;
            if (!skip_value(i)) return {};
            v.e = i;
            if (hit) return v;
            ws(i);
            if (i < o.e && s[i] == ',') { ++i; continue; }
            return {};
        }
    }
    bool is_null(JSpan v) const { return !v.ok() || s.compare(v.b, v.e - v.b, "null") == 0; }
    bool is_true(JSpan v) const { return v.ok() && s.compare(v.b, v.e - v.b, "true") == 0; }
    bool is_false(JSpan v) const { return v.ok() && s.compare(v.b, v.e - v.b, "false") == 0; }
    // a JSON string value, unescaped ("" for anything else)
    std::string str(JSpan v) const {
        std::string out;
        if (!v.ok() || s[v.b] != '"') return out;
        for (size_t i = v.b + 1; i + 1 < v.e; ++i) {
            if (s[i] != '\\') { out += s[i]; continue; }
            const char e = s[++i];
            if (e == 'u' && i + 4 < v.e) {
                uint32_t c = (uint32_t)strtoul(s.substr(i + 1, 4).c_str(), nullptr, 16);
                i += 4;
                if (c >= 0xd800 && c <= 0xdbff && i + 6 < v.e && s[i + 1] == '\\' && s[i + 2] == 'u') {
                    const uint32_t lo = (uint32_t)strtoul(s.substr(i + 3, 4).c_str(), nullptr, 16);
                    if (lo >= 0xdc00 && lo <= 0xdfff) { c = 0x10000 + ((c - 0xd800) << 10) + (lo - 0xdc00); i += 6; }
                }
                append_utf8(out, c);
            } else {
                out += e == 'n' ? '\n' : e == 't' ? '\t' : e == 'r' ? '\r' : e == 'b' ? '\b' : e == 'f' ? '\f' : e;
            }
        }
        return out;
    }
    // the elements of an array value
    bool elements(JSpan a, std::vector<JSpan> &out) const {
        size_t i = a.b;
        ws(i);
        if (i >= a.e || s[i] != '[') return false;
        ++i;
        ws(i);
        if (i < a.e && s[i] == ']') return true;
        for (;;) {
            ws(i);
            JSpan v;
            v.b = i;
            if (!skip_value(i)) return false;
            v.e = i;
            out.push_back(v);
            ws(i);
            if (i < a.e && s[i] == ',') { ++i; continue; }
            return i < a.e && s[i] == ']';
        }
    }
};

// tokenizer.json -> a WordPiece or byte-level BPE handle.  The stacks the two native tokenizers implement are the ones the
// reference's three models ship (BertNormalizer + BertPreTokenizer + WordPiece("##") + WordPiece decoder; ByteLevel + BPE +
// ByteLevel decoder); any other component is refused with MX_EUNSUPPORTED, never approximated.  The file's own truncation /
// padding blocks are ignored, as segment_text overrides the first (with_truncation, embedding.rs:172-176) and its decode calls
// skip the pad tokens of the second (skip_special_tokens = true, :182,189).
int build_from_tokenizer_json(mx_tokenizer *t, const std::string &js) {
    JScan J(js);
    JSpan root;
    root.b = 0;
    {
        size_t i = 0;
        if (!J.skip_value(i)) return fail(MX_EINVAL, "tokenizer.json: not a JSON document");
        root.e = i;
    }
    const JSpan model = J.find(root, "model");
    if (!model.ok()) return fail(MX_EINVAL, "tokenizer.json: no \"model\" object");
    const JSpan vocab = J.find(model, "vocab"), merges = J.find(model, "merges");
    if (!vocab.ok()) return fail(MX_EINVAL, "tokenizer.json: model has no vocab");
    std::string type = J.str(J.find(model, "type"));
    if (type.empty()) type = merges.ok() ? "BPE" : "WordPiece";  // files written before the tag existed
    auto type_of = [&](const char *key) { const JSpan v = J.find(root, key); return J.is_null(v) ? std::string() : J.str(J.find(v, "type")); };
    const std::string norm = type_of("normalizer"), pre = type_of("pre_tokenizer"), dec = type_of("decoder");
    {   // added_tokens: [{"id": 0, "content": "<s>", "single_word": false, "lstrip": false, "rstrip": false, "normalized": false, "special": true}, ..]
        const JSpan at = J.find(root, "added_tokens");
        std::vector<JSpan> items;
        if (at.ok() && !J.is_null(at) && J.elements(at, items)) {
            for (const JSpan &it : items) {
                const JSpan id = J.find(it, "id"), content = J.find(it, "content");
                if (!id.ok() || !content.ok()) return fail(MX_EINVAL, "tokenizer.json: an added token without id / content");
                if (J.is_true(J.find(it, "single_word")) || J.is_true(J.find(it, "normalized")))
                    return fail(MX_EUNSUPPORTED, "tokenizer.json: added token options single_word / normalized");
                const std::string s = J.str(content);
                if (s.empty()) continue;
                t->added.push_back({s, (int32_t)strtol(js.c_str() + id.b, nullptr, 10), J.is_true(J.find(it, "lstrip")), J.is_true(J.find(it, "rstrip")),
                                    !J.is_false(J.find(it, "special"))});
            }
            t->added_from_json = true;
        }
    }
    if (type == "WordPiece") {
        if (norm != "BertNormalizer") return fail(MX_EUNSUPPORTED, "tokenizer.json: normalizer '%s' (BertNormalizer only)", norm.c_str());
        if (pre != "BertPreTokenizer") return fail(MX_EUNSUPPORTED, "tokenizer.json: pre_tokenizer '%s' (BertPreTokenizer only)", pre.c_str());
        if (!dec.empty() && dec != "WordPiece") return fail(MX_EUNSUPPORTED, "tokenizer.json: decoder '%s'", dec.c_str());
        const JSpan nz = J.find(root, "normalizer");
        const bool lower = !J.is_false(J.find(nz, "lowercase"));
        // the native normaliser is BertNormalizer with its defaults: clean_text, handle_chinese_chars, accents stripped iff lowercase
        const JSpan sa = J.find(nz, "strip_accents");
        if (J.is_false(J.find(nz, "clean_text")) || J.is_false(J.find(nz, "handle_chinese_chars")) ||
            (!J.is_null(sa) && J.is_true(sa) != lower))
            return fail(MX_EUNSUPPORTED, "tokenizer.json: BertNormalizer options other than the defaults");
        const JSpan pfx = J.find(model, "continuing_subword_prefix"), unk = J.find(model, "unk_token");
        if (!J.is_null(pfx) && J.str(pfx) != "##") return fail(MX_EUNSUPPORTED, "tokenizer.json: continuing_subword_prefix '%s'", J.str(pfx).c_str());
        if (!J.is_null(unk) && J.str(unk) != "[UNK]") return fail(MX_EUNSUPPORTED, "tokenizer.json: unk_token '%s'", J.str(unk).c_str());
        const JSpan mc = J.find(model, "max_input_chars_per_word");
        if (mc.ok() && !J.is_null(mc) && strtol(js.c_str() + mc.b, nullptr, 10) != 100)
            return fail(MX_EUNSUPPORTED, "tokenizer.json: max_input_chars_per_word != 100");
        t->lowercase = lower;
        int rc = parse_vocab_json(t, js.substr(vocab.b, vocab.e - vocab.b));
        if (rc != MX_OK) return rc;
        return finish_vocab(t);
    }
    if (type == "BPE") {
        if (!norm.empty()) return fail(MX_EUNSUPPORTED, "tokenizer.json: normalizer '%s' in front of a byte-level BPE", norm.c_str());
        if (pre != "ByteLevel") return fail(MX_EUNSUPPORTED, "tokenizer.json: pre_tokenizer '%s' (ByteLevel only)", pre.c_str());
        if (!dec.empty() && dec != "ByteLevel") return fail(MX_EUNSUPPORTED, "tokenizer.json: decoder '%s'", dec.c_str());
        const JSpan pt = J.find(root, "pre_tokenizer");
        if (J.is_true(J.find(pt, "add_prefix_space")) || J.is_false(J.find(pt, "use_regex")))
            return fail(MX_EUNSUPPORTED, "tokenizer.json: ByteLevel options other than add_prefix_space = false, use_regex = true");
        for (const char *k : {"continuing_subword_prefix", "end_of_word_suffix"}) {
            const JSpan v = J.find(model, k);
            if (!J.is_null(v) && !J.str(v).empty()) return fail(MX_EUNSUPPORTED, "tokenizer.json: BPE option %s", k);
        }
        {
            const JSpan v = J.find(model, "dropout");
            if (!J.is_null(v) && strtod(js.c_str() + v.b, nullptr) != 0.0) return fail(MX_EUNSUPPORTED, "tokenizer.json: BPE dropout");
        }
        if (J.is_true(J.find(model, "byte_fallback")) || J.is_true(J.find(model, "ignore_merges")))
            return fail(MX_EUNSUPPORTED, "tokenizer.json: BPE byte_fallback / ignore_merges");
        if (!merges.ok()) return fail(MX_EINVAL, "tokenizer.json: BPE model has no merges");
        std::vector<JSpan> ms;
        if (!J.elements(merges, ms)) return fail(MX_EINVAL, "tokenizer.json: merges is not an array");
        std::string lines;  // the merges.txt the same tokenizer would ship
        for (const JSpan &m : ms) {
            if (js[m.b] == '"') {  // "a b"
                lines += J.str(m);
            } else {  // ["a", "b"] (tokenizers >= 0.20)
                std::vector<JSpan> ab;
                if (!J.elements(m, ab) || ab.size() != 2) return fail(MX_EINVAL, "tokenizer.json: a merge that is not a pair");
                lines += J.str(ab[0]) + " " + J.str(ab[1]);
            }
            lines += '\n';
        }
        int rc = parse_vocab_json(t, js.substr(vocab.b, vocab.e - vocab.b));
        if (rc != MX_OK) return rc;
        std::istringstream in(lines);
        return finish_bpe(t, in);
    }
    return fail(MX_EUNSUPPORTED, "tokenizer.json: model type '%s' (WordPiece / byte-level BPE only)", type.c_str());
}

}  // namespace

extern "C" {

// tokenizer.json, the file Tokenizer::from_pretrained reads (embedding.rs:163)
int mx_tokenizer_create_from_json_memory(const char *json, size_t nbytes, mx_tokenizer **out) try {
    if (!json || !out) return fail(MX_EINVAL, "null argument");
    *out = nullptr;
    std::unique_ptr<mx_tokenizer> t(new mx_tokenizer());  // (the builders may throw: nothing leaks)
    int rc = build_from_tokenizer_json(t.get(), std::string(json, nbytes));
    if (rc != MX_OK) return rc;
    *out = t.release();
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_tokenizer_create_from_json(const char *tokenizer_json_path, mx_tokenizer **out) try {
    if (!tokenizer_json_path || !out) return fail(MX_EINVAL, "null argument");
    *out = nullptr;
    std::ifstream f(tokenizer_json_path, std::ios::binary);
    if (!f) return fail(MX_EIO, "Unable to load model <%s>", tokenizer_json_path);  // embedding.rs:166-169 wording
    std::ostringstream ss;
    ss << f.rdbuf();
    const std::string js = ss.str();
    return mx_tokenizer_create_from_json_memory(js.data(), js.size(), out);
} catch (...) {
    return guard_exception();
}

int mx_tokenizer_create(const char *vocab_path, int lowercase, mx_tokenizer **out) try {
    if (!vocab_path || !out) return fail(MX_EINVAL, "null argument");
    *out = nullptr;
    std::ifstream f(vocab_path);
    if (!f) return fail(MX_EIO, "Unable to load model <%s>", vocab_path);  // embedding.rs:166-169 wording
    std::unique_ptr<mx_tokenizer> t(new mx_tokenizer());
    t->lowercase = lowercase != 0;
    int rc = parse_vocab(t.get(), f);
    if (rc != MX_OK) return rc;
    *out = t.release();
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_tokenizer_create_from_memory(const char *vocab, size_t nbytes, int lowercase, mx_tokenizer **out) try {
    if (!vocab || !out) return fail(MX_EINVAL, "null argument");
    *out = nullptr;
    std::istringstream in(std::string(vocab, nbytes));
    std::unique_ptr<mx_tokenizer> t(new mx_tokenizer());
    t->lowercase = lowercase != 0;
    int rc = parse_vocab(t.get(), in);
    if (rc != MX_OK) return rc;
    *out = t.release();
    return MX_OK;
} catch (...) {
    return guard_exception();
}

// Byte-level BPE (RoBERTa family: all-distilroberta-v1, embedding.rs:29,159): vocab.json + merges.txt as the checkpoints ship them
int mx_tokenizer_create_bpe_from_memory(const char *vocab_json, size_t n_vocab, const char *merges, size_t n_merges, mx_tokenizer **out) try {
    if (!vocab_json || !merges || !out) return fail(MX_EINVAL, "null argument");
    *out = nullptr;
    std::unique_ptr<mx_tokenizer> t(new mx_tokenizer());
    int rc = parse_vocab_json(t.get(), std::string(vocab_json, n_vocab));
    if (rc == MX_OK) {
        std::istringstream m(std::string(merges, n_merges));
        rc = finish_bpe(t.get(), m);
    }
    if (rc != MX_OK) return rc;
    *out = t.release();
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_tokenizer_create_bpe(const char *vocab_json_path, const char *merges_path, mx_tokenizer **out) try {
    if (!vocab_json_path || !merges_path || !out) return fail(MX_EINVAL, "null argument");
    *out = nullptr;
    std::ifstream fv(vocab_json_path, std::ios::binary), fm(merges_path, std::ios::binary);
    if (!fv) return fail(MX_EIO, "Unable to load model <%s>", vocab_json_path);
    if (!fm) return fail(MX_EIO, "Unable to load model <%s>", merges_path);
    std::ostringstream sv, sm;
    sv << fv.rdbuf();
    sm << fm.rdbuf();
    const std::string v = sv.str(), m = sm.str();
    return mx_tokenizer_create_bpe_from_memory(v.data(), v.size(), m.data(), m.size(), out);
} catch (...) {
    return guard_exception();
}

void mx_tokenizer_destroy(mx_tokenizer *t) { delete t; }

int mx_tokenizer_vocab_size(mx_tokenizer *t, int *n) try {
    if (!t || !n) return fail(MX_EINVAL, "null argument");
    *n = (int)t->vocab.size();
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_tokenizer_encode(mx_tokenizer *t, const char *text, int add_special_tokens, int32_t *ids, int cap, int *n) try {
    if (!t || !text || !n || (cap > 0 && !ids)) return fail(MX_EINVAL, "null argument");
    std::vector<int32_t> v = encode_plain(t, text);
    if (add_special_tokens) {
        v.insert(v.begin(), t->cls);
        v.push_back(t->sep);
    }
    *n = (int)v.size();  // always the full length: call again with a larger buffer if n > cap
    for (int i = 0; i < (int)v.size() && i < cap; ++i) ids[i] = v[i];
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_tokenizer_decode(mx_tokenizer *t, const int32_t *ids, int n, int skip_special_tokens, char *out, size_t cap,
                        size_t *nbytes) try {
    if (!t || (n > 0 && !ids) || !nbytes) return fail(MX_EINVAL, "null argument");
    const std::string s = decode_ids(t, ids, n, skip_special_tokens != 0);
    *nbytes = s.size() + 1;
    if (out && cap >= s.size() + 1) memcpy(out, s.c_str(), s.size() + 1);
    return MX_OK;
} catch (...) {
    return guard_exception();
}

// segment_text's windows of one text, each NUL-terminated, appended to buf; -> number of windows
static int segment_into(const mx_tokenizer *t, const char *text, int max_length, int stride, std::string &buf) {
    const std::vector<int32_t> ids = encode_plain(t, text);  // no special tokens (embedding.rs:181)
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
        if (nseg == 0) {  // only the first window gets the replace (embedding.rs:183 vs :189-194)
            std::string seg = decode_ids(t, ids.data() + start, (int)(stop - start), true);
            replace_all(seg, " ' ", "'");
            buf += seg;
        } else if (t->kind == 1) {
            buf += decode_ids(t, ids.data() + start, (int)(stop - start), true);
        } else {
            decode_append(t, ids.data() + start, (int)(stop - start), true, buf);
        }
        buf.push_back('\0');
        ++nseg;
    }
    return nseg;
}

int mx_tokenizer_segment(mx_tokenizer *t, const char *text, int max_length, int stride, char *out, size_t cap,
                         size_t *nbytes, int *n_segments) try {
    if (!t || !text || !nbytes || !n_segments) return fail(MX_EINVAL, "null argument");
    if (max_length < 1 || stride < 0 || stride >= max_length) return fail(MX_EINVAL, "need 0 <= stride < max_length");
    std::string buf;
    *n_segments = segment_into(t, text, max_length, stride, buf);
    *nbytes = buf.size();
    if (out && cap >= buf.size()) memcpy(out, buf.data(), buf.size());
    return MX_OK;
} catch (...) {
    return guard_exception();
}

// segment_text for a batch of documents (the ingest worker drains its queue: tasks.rs:17-19 runs one document per task, up to
// five tasks at a time, worker/lib.rs:36): documents are independent and the tokenizer is read-only, so they are dealt to
// host threads.  out: the windows of text 0, then of text 1, ... each NUL-terminated; n_segments[i] = windows of text i.
// nbytes is always the size needed; out is filled only when cap covers it (call again with a larger buffer otherwise).
int mx_tokenizer_segment_batch(mx_tokenizer *t, const char *const *texts, int n_texts, int max_length, int stride, char *out,
                               size_t cap, size_t *nbytes, int32_t *n_segments) try {
    if (!t || (n_texts > 0 && (!texts || !n_segments)) || !nbytes || n_texts < 0) return fail(MX_EINVAL, "null argument");
    if (max_length < 1 || stride < 0 || stride >= max_length) return fail(MX_EINVAL, "need 0 <= stride < max_length");
    for (int i = 0; i < n_texts; ++i)
        if (!texts[i]) return fail(MX_EINVAL, "texts[%d] is null", i);
    std::vector<std::string> bufs((size_t)n_texts);
    std::atomic<int> next{0};
    std::atomic<bool> failed{false};  // an exception must not leave a helper thread (std::terminate): reported as MX_ENOMEM below
    auto work = [&] {
        try {
            for (int i; (i = next.fetch_add(1)) < n_texts;) n_segments[i] = segment_into(t, texts[i], max_length, stride, bufs[(size_t)i]);
        } catch (...) {
            failed = true;
        }
    };
    const int nthr = std::min<int>((int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u), n_texts);
    if (nthr <= 1) {
        work();
    } else {
        // (a std::system_error from thread creation must not unwind past joinable threads: std::terminate.  The threads
        // that did start share the queue; this thread takes the rest)
        std::vector<std::thread> pool;
        pool.reserve((size_t)nthr);
        try {
            for (int i = 0; i < nthr; ++i) pool.emplace_back(work);
        } catch (...) {
            work();
        }
        for (auto &th : pool) th.join();
    }
    if (failed) return fail(MX_ENOMEM, "segmenting a batch of %d texts failed (out of host memory)", n_texts);
    size_t total = 0;
    for (const std::string &b : bufs) total += b.size();
    *nbytes = total;
    if (out && cap >= total) {
        size_t o = 0;
        for (const std::string &b : bufs) {
            memcpy(out + o, b.data(), b.size());
            o += b.size();
        }
    }
    return MX_OK;
} catch (...) {
    return guard_exception();
}

// the stage-by-stage WordPiece encoder (normalize -> pre_tokenize -> wordpiece), for tests that hold the one-pass form against it
int mx_tokenizer_encode_staged(mx_tokenizer *t, const char *text, int32_t *ids, int cap, int *n) try {
    if (!t || !text || !n || (cap > 0 && !ids)) return fail(MX_EINVAL, "null argument");
    if (t->kind != 0) return fail(MX_EUNSUPPORTED, "WordPiece handles only");
#ifdef MEMEX_TESTING  // libmemex_hip_testing.so only (tests/test_abi.py: an exception inside an entry point comes back as an error code)
    if (strcmp(text, "\x01\x02throw:bad_alloc") == 0) throw std::bad_alloc();
    if (strcmp(text, "\x01\x02throw:logic_error") == 0) throw std::logic_error("thrown on request");
#endif
    const std::vector<int32_t> v = encode_staged(t, text);
    *n = (int)v.size();
    for (int i = 0; i < (int)v.size() && i < cap; ++i) ids[i] = v[i];
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_tokenizer_encode_batch(mx_tokenizer *t, const char *const *texts, int B, int max_seq_length, int32_t *ids,
                              int s_cap, int32_t *lens, int *S) try {
    if (!t || (B > 0 && (!texts || !ids || !lens)) || !S) return fail(MX_EINVAL, "null argument");
    if (max_seq_length < 2) return fail(MX_EINVAL, "max_seq_length must be >= 2");
    for (int b = 0; b < B; ++b)
        if (!texts[b]) return fail(MX_EINVAL, "texts[%d] is null", b);
    std::vector<std::vector<int32_t>> rows((size_t)B);
    // rows are independent and the tokenizer state is read-only: split the batch over host threads
    // (the GPU consumes ~20M tokens/s; one thread produces 1-2M)
    std::atomic<bool> failed{false};  // (as in mx_tokenizer_segment_batch)
    auto work = [&](int b0, int b1) {
        try {
            for (int b = b0; b < b1; ++b) {
                std::vector<int32_t> v = encode_plain(t, texts[b]);
                if ((int)v.size() > max_seq_length - 2) v.resize((size_t)max_seq_length - 2);  // truncate, keep room for specials
                v.insert(v.begin(), t->cls);
                v.push_back(t->sep);
                rows[(size_t)b] = std::move(v);
            }
        } catch (...) {
            failed = true;
        }
    };
    int nthr = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    nthr = std::min(nthr, B / 16);  // below ~16 texts per thread the spawn cost dominates
    if (nthr <= 1) {
        work(0, B);
    } else {
        std::vector<std::thread> pool;
        pool.reserve((size_t)nthr);
        int started = 0;  // (thread creation may fail: the slices of the threads that did not start run here)
        try {
            for (; started < nthr; ++started) pool.emplace_back(work, (int)((long)B * started / nthr), (int)((long)B * (started + 1) / nthr));
        } catch (...) {
            work((int)((long)B * started / nthr), B);
        }
        for (auto &th : pool) th.join();
    }
    if (failed) return fail(MX_ENOMEM, "tokenising a batch of %d texts failed (out of host memory)", B);
    int smax = 0;
    for (int b = 0; b < B; ++b) smax = std::max(smax, (int)rows[(size_t)b].size());
    *S = smax;
    if (smax > s_cap) return fail(MX_EINVAL, "row capacity %d < batch maximum %d", s_cap, smax);
    for (int b = 0; b < B; ++b) {
        lens[b] = (int32_t)rows[b].size();
        for (int i = 0; i < s_cap; ++i) ids[(size_t)b * s_cap + i] = i < (int)rows[b].size() ? rows[b][i] : t->pad;
    }
    return MX_OK;
} catch (...) {
    return guard_exception();
}

}  // extern "C"
