// encoder.hip -- host logic of the sentence encoder behind the C ABI (include/memex_hip.h).
//
// Stands in for the rust-bert model that memex's embedder thread owns
// (reference lib/libmemex/src/llm/embedding.rs:94-135): `create_model()` (:99-100) becomes
// mx_encoder_create, `model.encode(&segments)` (:109) becomes mx_encoder_encode after host-side
// tokenisation.  One handle = one device + one HIP stream; use it from one thread at a time
// (the reference confines the model to a dedicated thread too, embedding.rs:98).
//
// Per call: sequences are packed back to back (no padding FLOPs), then
//   embed+LN -> L x { QKV GEMM -> attention -> out-proj GEMM(+res+LN) -> FFN1 GEMM(+GELU)
//                     -> FFN2 GEMM(+res+LN) } -> pool (+L2 normalise)
// with bf16 activations/weights, f32 accumulation and f32 LayerNorm/softmax/GELU.  Three kernel sets serve that layer:
// large passes (pgemm_kernel / the fused tail_kernel), passes of <= 2048 rows of the hidden-384 models (encoder_small.hip:
// query-time embedding, embedding.rs:146-151), and MX_PREC_BF16X3 (encoder_precise.hip: split bf16 operands, f32 hidden
// state and attention) -- see include/memex_hip.h for what a caller can observe of the difference.
#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "encoder_kernels.h"
#include "mx_common.h"
#include "mx_debug.h"

using namespace mx;

namespace {

inline uint16_t f32_to_bf16(float f) {  // round to nearest even
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

struct Layer {
    bf16_t *wqkv = nullptr;  // [3H, H]
    float *bqkv = nullptr;   // [3H]
    bf16_t *wo = nullptr;    // [H, H]
    float *bo = nullptr, *ln1g = nullptr, *ln1b = nullptr;
    bf16_t *wi = nullptr;    // [F, H]
    float *bi = nullptr;
    bf16_t *wo2 = nullptr;   // [H, F]
    bf16_t *wf = nullptr;    // wo, wi and wo2 once more, as tail_kernel's per-wave fragment streams (fused tail only)
    float *bo2 = nullptr, *ln2g = nullptr, *ln2b = nullptr;
    // MX_PREC_BF16X3: the same four matrices with k tripled, [hi | hi | lo] (upload_weight3); the bf16 ones above stay null
    bf16_t *wqkv3 = nullptr, *wo3 = nullptr, *wi3 = nullptr, *wo23 = nullptr;
    // MX_PREC_MIXED: the attention block as in MX_PREC_BF16X3 (wqkv3, wo3), the MLP's two matrices as fp16 with k DOUBLED,
    // [w | w] against activations split as fp16 [hi | lo] (upload_weight2h); wi3 / wo23 stay null
    bf16_t *wi2h = nullptr, *wo22h = nullptr;
};

// kernel attributes (dynamic LDS sizes) are per device: set up once on every device an encoder is created on.  The grid
// sizes the kernels derive from the CU count are taken from the first device (the GPUs of a node are identical).
std::mutex g_enc_setup_mu;
bool g_enc_setup_done[64] = {};
std::mutex g_enc_reg_mu;
std::map<std::string, mx_encoder *> g_enc_registry;

}  // namespace

struct mx_encoder {
    mx_encoder_cfg cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::vector<void *> allocs;  // every weight buffer (freed on destroy)
    float *word = nullptr, *pos = nullptr, *type0 = nullptr, *eg = nullptr, *eb = nullptr;
    std::vector<Layer> layers;
    // workspace (grown on demand)
    int ws_rows = 0, ws_seqs = 0, ws_ids = 0;
    bf16_t *x = nullptr, *x1 = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *ctx = nullptr, *hbuf = nullptr;
    // MX_PREC_BF16X3 workspace: f32 hidden state / QKV / GEMM result, split (3x wide) GEMM operands
    float *xf = nullptr, *qkvf = nullptr, *af = nullptr;
    bf16_t *xs = nullptr, *ctxs = nullptr, *hs = nullptr;
    bool precise = false;
    bool mixed = false;       // MX_PREC_MIXED / MX_PREC_MIXED1: precise, with the MLP on fp16 products
    bool mlp1 = false;        // MX_PREC_MIXED1: ... ONE fp16 product per product in the MLP (weights, input and GELU output one fp16 value each),
                              // and P as one bf16 value in P.V
    // small passes (<= kSmallRows packed rows, fused-tail models): x1 of the layer in flight, the MLP's partial products
    bf16_t *sp_x1 = nullptr;
    float *sp_part = nullptr;
    // small passes of the GEMM-by-GEMM models (hidden 768): the two Add & LayerNorm GEMMs run split over k with f32 partials and a
    // reduce + LayerNorm kernel behind them (one 64 x 768 workgroup looping over k = 3072 is an 80 us latency chain)
    bool split_small = true;  // MEMEX_HIP_DEBUG splitk=0: keep the fused Add & LayerNorm GEMMs at every pass size (tests, A/B)
    float *sk_part = nullptr; // [kSplitMax][kSplitRows][H] f32
    float *sk_zero = nullptr; // [3H] zeros: the partial GEMMs' bias (the reduce kernel adds the real one)
    bool small_pass = true;   // MEMEX_HIP_DEBUG small=0: small passes take the large-pass kernels (tests, A/B)
    bool attn_f32 = false;    // MEMEX_HIP_DEBUG attn_f32=1: the bf16x3 mode's attention on the f32 MFMA instead of split bf16 products (tests)
    int small_rows = kSmallRows;  // passes of at most this many packed rows take the small-pass layer (MEMEX_HIP_DEBUG small_rows=N)
    char *h_io = nullptr;     // pinned, device-mapped page of a query-sized host call: ids | lens | embeddings (mx_encoder_encode)
    int32_t *cu = nullptr, *tok_seq = nullptr, *tok_pos = nullptr, *lens_dev = nullptr, *ids_dev = nullptr;
    void *attn_plan = nullptr;  // attention work list of the pass in flight (kAttnPlanBytesPerSeq per sequence)
    float *out_dev = nullptr;
    bool profiling = false;
    bool fused_tail = false;  // layer tail as one kernel (hidden 384); MEMEX_HIP_DEBUG unfused_tail=1 keeps the three GEMMs
    bool pgemm = true;        // large passes run their GEMMs on pgemm_kernel (encoder_pgemm.hip); MEMEX_HIP_DEBUG pgemm=0: gemm_kernel
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_wait = nullptr, ev_done = nullptr;
    mx_encoder_stats stats{};
    std::string key;  // registry key (mx_encoder_open); empty = private
    int refs = 1;
};

namespace {

int upload_f32(mx_encoder *e, const float *src, size_t n, float **dst) {
    MX_HIP(hipMalloc(dst, n * sizeof(float)));
    e->allocs.push_back(*dst);
    MX_HIP(hipMemcpy(*dst, src, n * sizeof(float), hipMemcpyHostToDevice));
    return MX_OK;
}

// Linear weight [rows, k] (nn.Linear layout) -> bf16, K-blocked [k/32][rows][32]: every 16-row x 64-byte
// staging piece of the GEMM is then one contiguous KiB
int upload_weight(mx_encoder *e, const float *src, size_t rows, size_t k, bf16_t **dst) {
    std::vector<uint16_t> tmp(rows * k);
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < k; ++c) tmp[((c >> 5) * rows + r) * 32 + (c & 31)] = f32_to_bf16(src[r * k + c]);
    MX_HIP(hipMalloc(dst, tmp.size() * sizeof(uint16_t)));
    e->allocs.push_back(*dst);
    MX_HIP(hipMemcpy(*dst, tmp.data(), tmp.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return MX_OK;
}

// the bf16x3 form of a Linear weight: k tripled, [hi | hi | lo] against activations split as [hi | lo | hi]
// (encoder_precise.hip), K-blocked like upload_weight's
int upload_weight3(mx_encoder *e, const float *src, size_t rows, size_t k, bf16_t **dst) {
    std::vector<uint16_t> tmp(rows * 3 * k);
    auto at = [&](size_t r, size_t c) -> uint16_t & { return tmp[((c >> 5) * rows + r) * 32 + (c & 31)]; };
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < k; ++c) {
            const float w = src[r * k + c];
            const uint16_t hi = f32_to_bf16(w);
            uint32_t hb = (uint32_t)hi << 16;
            float hf;
            memcpy(&hf, &hb, 4);
            const uint16_t lo = f32_to_bf16(w - hf);
            at(r, c) = hi;
            at(r, k + c) = hi;
            at(r, 2 * k + c) = lo;
        }
    MX_HIP(hipMalloc(dst, tmp.size() * sizeof(uint16_t)));
    e->allocs.push_back(*dst);
    MX_HIP(hipMemcpy(*dst, tmp.data(), tmp.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return MX_OK;
}

// the mixed modes' form of a Linear weight: ONE fp16 value per element (saturating).  copies = 2 (MX_PREC_MIXED): k doubled, [w | w]
// against activations split as fp16 [hi | lo] -- two fp16 MFMA products per product, 11 + 22 significant bits; copies = 1
// (MX_PREC_MIXED1): [w] against ONE fp16 value per activation
int upload_weight2h(mx_encoder *e, const float *src, size_t rows, size_t k, bf16_t **dst, int copies = 2) {
    std::vector<uint16_t> tmp(rows * (size_t)copies * k);
    auto at = [&](size_t r, size_t c) -> uint16_t & { return tmp[((c >> 5) * rows + r) * 32 + (c & 31)]; };
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < k; ++c) {
            const float w = std::min(std::max(src[r * k + c], -65504.0f), 65504.0f);
            const _Float16 hv = (_Float16)w;  // round to nearest even, subnormals kept
            uint16_t bits;
            memcpy(&bits, &hv, 2);
            at(r, c) = bits;
            if (copies == 2) at(r, k + c) = bits;
        }
    MX_HIP(hipMalloc(dst, tmp.size() * sizeof(uint16_t)));
    e->allocs.push_back(*dst);
    MX_HIP(hipMemcpy(*dst, tmp.data(), tmp.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return MX_OK;
}

// wo [H, H], wi [F, H] and wo2 [H, F] -> the fused tail kernel's per-wave fragment streams (encoder_tail.hip)
int upload_tail_stream(mx_encoder *e, const float *wo, const float *wi, const float *wo2, size_t F, bf16_t **dst) {
    std::vector<uint16_t> st(tail_stream_elems((int)F));
    tail_stream_layout(wo, wi, wo2, (int)F, st.data(), &f32_to_bf16);
    MX_HIP(hipMalloc(dst, st.size() * sizeof(uint16_t)));
    e->allocs.push_back(*dst);
    MX_HIP(hipMemcpy(*dst, st.data(), st.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return MX_OK;
}

void free_ws(mx_encoder *e) {
    if (e->stream) (void)hipStreamSynchronize(e->stream);  // nothing queued may still use the buffers
    void *ptrs[] = {e->x, e->x1, e->q, e->k, e->vt, e->ctx, e->hbuf, e->tok_seq, e->tok_pos, e->xf, e->qkvf, e->af, e->xs, e->ctxs, e->hs};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    e->x = e->x1 = e->q = e->k = e->vt = e->ctx = e->hbuf = nullptr;
    e->xf = e->qkvf = e->af = nullptr;
    e->xs = e->ctxs = e->hs = nullptr;
    e->tok_seq = e->tok_pos = nullptr;
    e->ws_rows = 0;
}

int ensure_ws(mx_encoder *e, int rows, int seqs, int n_ids) {
    const int H = e->cfg.hidden, F = e->cfg.ffn;
    if (rows > e->ws_rows) {
        free_ws(e);
        const size_t r = (size_t)rows;
        if (e->precise) {
            auto zalloc = [&](void **b, size_t bytes) -> hipError_t {
                hipError_t he = hipMalloc(b, bytes);
                return he != hipSuccess ? he : hipMemsetAsync(*b, 0, bytes, e->stream);
            };
            MX_HIP(zalloc((void **)&e->xf, r * H * sizeof(float)));
            MX_HIP(zalloc((void **)&e->qkvf, r * 3 * H * sizeof(float)));
            MX_HIP(zalloc((void **)&e->af, r * H * sizeof(float)));
            MX_HIP(zalloc((void **)&e->xs, r * 3 * H * sizeof(uint16_t)));
            MX_HIP(zalloc((void **)&e->ctxs, r * 3 * H * sizeof(uint16_t)));
            MX_HIP(zalloc((void **)&e->hs, r * 3 * F * sizeof(uint16_t)));
        }
        bf16_t **bufs[] = {&e->x, &e->q, &e->k, &e->vt, &e->ctx};
        for (bf16_t **b : bufs) {
            if (e->precise) break;
            MX_HIP(hipMalloc(b, r * H * sizeof(uint16_t)));
            MX_HIP(hipMemsetAsync(*b, 0, r * H * sizeof(uint16_t), e->stream));
        }
        if (!e->fused_tail && !e->precise) {  // x1 [rows, H] and the [rows, ffn] MLP intermediate only exist GEMM by GEMM:
            // the fused tail kernel keeps both on chip (and is the faster path at every pass size, a
            // single short query included: 0.49 vs 0.58 ms)
            MX_HIP(hipMalloc(&e->x1, r * H * sizeof(uint16_t)));
            MX_HIP(hipMemsetAsync(e->x1, 0, r * H * sizeof(uint16_t), e->stream));
            MX_HIP(hipMalloc(&e->hbuf, r * F * sizeof(uint16_t)));
            MX_HIP(hipMemsetAsync(e->hbuf, 0, r * F * sizeof(uint16_t), e->stream));
        }
        MX_HIP(hipMalloc(&e->tok_seq, r * sizeof(int32_t)));
        MX_HIP(hipMalloc(&e->tok_pos, r * sizeof(int32_t)));
        e->ws_rows = rows;
    }
    if (seqs > e->ws_seqs) {
        if (e->cu) (void)hipFree(e->cu);
        if (e->lens_dev) (void)hipFree(e->lens_dev);
        if (e->out_dev) (void)hipFree(e->out_dev);
        if (e->attn_plan) (void)hipFree(e->attn_plan);
        e->cu = nullptr; e->lens_dev = nullptr; e->out_dev = nullptr; e->attn_plan = nullptr;
        e->ws_seqs = 0;
        MX_HIP(hipMalloc(&e->cu, ((size_t)seqs + 1) * sizeof(int32_t)));
        MX_HIP(hipMalloc(&e->lens_dev, (size_t)seqs * sizeof(int32_t)));
        MX_HIP(hipMalloc(&e->out_dev, (size_t)seqs * H * sizeof(float)));
        MX_HIP(hipMalloc(&e->attn_plan, (size_t)seqs * kAttnPlanBytesPerSeq));
        e->ws_seqs = seqs;
    }
    if (n_ids > e->ws_ids) {
        if (e->ids_dev) (void)hipFree(e->ids_dev);
        e->ids_dev = nullptr;
        e->ws_ids = 0;
        MX_HIP(hipMalloc(&e->ids_dev, (size_t)n_ids * sizeof(int32_t)));
        e->ws_ids = n_ids;
    }
    return MX_OK;
}

// query-sized host calls (mx_encoder_encode): up to kIoIds token ids in kIoSeqs sequences travel through one mapped page
constexpr size_t kIoIds = 4096, kIoSeqs = 64;
inline size_t kIoBytes(int hidden) { return (kIoIds + kIoSeqs) * sizeof(int32_t) + kIoSeqs * (size_t)hidden * sizeof(float); }
constexpr int kMaxSeqsPerPass = 1024;
// packed rows per pass (131072): bounds the workspace; measured: larger passes are faster (fixed
// per-pass costs), smaller ones do not help (the GEMMs are not bandwidth-bound)
constexpr int kMaxRowsPerPass = 1 << 17;
// passes of at least this many packed rows use pgemm_kernel: 128 row tiles of 256 = 16 per XCD, every CU busy for
// the 768-wide outputs (3 column tiles); below, gemm_kernel's small tiles fill the chip better
constexpr int kPgemmRows = 32768;
constexpr int kSplitRows = 4096;  // passes up to this many packed rows split the Add & LayerNorm GEMMs of a hidden-768 layer over k
constexpr int kSplitMax = 8;

// one pass: sequences [0, B) with device ids [B,S] (row pitch S) and HOST lens; output d_out [B,H]
int encode_pass(mx_encoder *e, const int32_t *d_ids, const int32_t *h_lens, const int32_t *d_lens, int B, int S,
                float *d_out) {
    const mx_encoder_cfg &c = e->cfg;
    const int H = c.hidden, F = c.ffn, heads = c.heads, dh = H / heads;
    long rows = 0;
    int max_len = 1;
    double attn_flops = 0.0;
    uint64_t tokens = 0;
    for (int b = 0; b < B; ++b) {
        const int l = std::min(std::max(h_lens[b], 1), S);
        rows += (l + kSeqAlign - 1) / kSeqAlign * kSeqAlign;
        max_len = std::max(max_len, l);
        tokens += (uint64_t)l;
        attn_flops += 4.0 * (double)l * (double)l * H;
    }
    const int t_pad = (int)round_up((uint64_t)rows + 32, kRowPad);
    // t_pad rows are ALLOCATED (leading dimensions, the token map, the 32 rows a key block may read past the last sequence);
    // m_c rows are COMPUTED by the GEMMs, the layer tail and the LayerNorm passes.  They differ by one 256-row tile exactly
    // when a pass is full (131072 packed rows: 256 chunks of 512 tokens), and that 513th tile cost a whole extra round of
    // workgroups in every kernel whose grid the other 512 fill exactly (pgemm_kernel: 7 tiles on three CUs of XCD 0 where
    // every other CU has 6; tail_kernel / gemm_kernel: 2052 workgroups on 512 slots) -- 8-17 % of those launches.  Rows
    // past m_c keep whatever the buffers held (zeros, or finite activations of an earlier pass): nothing reads them
    // unmasked.
    const int m_c = (int)round_up((uint64_t)rows, kRowPad);  // (A/B against the old extent: profiles/r5_whole_tile_passes_ab.txt)
    int rc = ensure_ws(e, t_pad, B, 0);
    if (rc != MX_OK) return rc;
    hipStream_t st = e->stream;
    if ((size_t)attention_groups(heads, dh, max_len, B) * 16 > kAttnPlanBytesPerSeq) return fail(MX_EINVAL, "more than 16 head groups per sequence");
    MX_HIP(launch_token_map(st, d_lens, B, S, e->cu, e->tok_seq, e->tok_pos, t_pad, heads, dh, max_len, e->attn_plan));
    if (e->precise) {
        // MX_PREC_BF16X3 (encoder_precise.hip): split operands through the unchanged GEMM loops with k tripled (pgemm_kernel for
        // the shapes it takes in large passes, gemm_kernel otherwise), f32 everywhere else
        const bool pbig = e->pgemm && t_pad >= kPgemmRows;
        auto pgemm_or_gemm = [&](int epi, const GemmParams &gp) -> hipError_t {
            if (pbig && pgemm_supported(epi, gp)) return launch_pgemm(st, epi, gp);
            return launch_gemm(st, epi, gp);
        };
        MX_HIP(launch_embed_ln_precise(st, d_ids, S, e->tok_seq, e->tok_pos, t_pad, H, e->word, e->pos, e->type0, e->eg, e->eb,
                                       c.ln_eps, c.vocab, e->xf, e->xs));
        for (const Layer &L : e->layers) {
            GemmParams g{};
            g.a = e->xs; g.lda = 3 * H; g.w = L.wqkv3; g.w_rows = 3 * H; g.bias = L.bqkv; g.m = m_c; g.n = 3 * H; g.k = 3 * H;
            g.out_f32 = e->qkvf; g.ldo = 3 * H;
            MX_HIP(pgemm_or_gemm(EPI_F32, g));
            MX_HIP(launch_attention_f32(st, e->qkvf, e->cu, d_lens, B, max_len, heads, dh, H, e->ctxs, e->attn_f32, e->mlp1));
            GemmParams o{};
            o.a = e->ctxs; o.lda = 3 * H; o.w = L.wo3; o.w_rows = H; o.bias = L.bo; o.m = m_c; o.n = H; o.k = 3 * H;
            o.out_f32 = e->af; o.ldo = H;
            MX_HIP(pgemm_or_gemm(EPI_F32, o));
            // the MLP: three bf16 products per product (MX_PREC_BF16X3), or two fp16 ones (MX_PREC_MIXED: x1 and gelu(..) as fp16
            // hi + lo in the same buffers, the weights as one fp16 value)
            const int sp = e->mlp1 ? 1 : e->mixed ? 2 : 3;
            MX_HIP(launch_add_ln_split(st, e->af, e->xf, e->xs, m_c, H, L.ln1g, L.ln1b, c.ln_eps, e->mlp1 ? 2 : e->mixed ? 1 : 0));
            GemmParams f1{};
            f1.a = e->xs; f1.lda = sp * H; f1.w = e->mixed ? L.wi2h : L.wi3; f1.w_rows = F; f1.bias = L.bi; f1.m = m_c; f1.n = F; f1.k = sp * H;
            f1.out = e->hs; f1.ldo = sp * F; f1.single = e->mlp1 ? 1 : 0;
            MX_HIP(pgemm_or_gemm(e->mixed ? EPI_GELU_SPLIT_H : EPI_GELU_SPLIT, f1));
            GemmParams f2{};
            f2.a = e->hs; f2.lda = sp * F; f2.w = e->mixed ? L.wo22h : L.wo23; f2.w_rows = H; f2.bias = L.bo2; f2.m = m_c; f2.n = H; f2.k = sp * F;
            f2.out_f32 = e->af; f2.ldo = H;
            MX_HIP(pgemm_or_gemm(e->mixed ? EPI_F32_H : EPI_F32, f2));
            MX_HIP(launch_add_ln_split(st, e->af, e->xf, e->xs, m_c, H, L.ln2g, L.ln2b, c.ln_eps));
        }
        MX_HIP(launch_pool(st, nullptr, e->xf, e->cu, d_lens, B, H, c.pooling == MX_POOL_CLS, c.normalize, d_out));
        e->stats.sequences += (uint64_t)B;
        e->stats.tokens += tokens;
        e->stats.flops += (double)c.layers * ((double)tokens * (8.0 * H * H + 4.0 * (double)H * F) + attn_flops);
        return MX_OK;
    }
    MX_HIP(launch_embed_ln(st, d_ids, S, e->tok_seq, e->tok_pos, t_pad, H, e->word, e->pos, e->type0, e->eg, e->eb,
                           c.ln_eps, c.vocab, e->x));
    const float qscale = (float)(1.4426950408889634 / std::sqrt((double)dh));
    // Large passes (>= kPgemmRows packed rows) run every GEMM whose shape it takes on pgemm_kernel; the Add & LayerNorm
    // GEMMs then leave y = product + bias + residual and ln_rows_kernel normalises it in place.
    const bool big = e->pgemm && t_pad >= kPgemmRows;
    auto gemm = [&](int epi, const GemmParams &gp) -> hipError_t {
        if (big && pgemm_supported(epi, gp)) return launch_pgemm(st, epi, gp);
        return launch_gemm(st, epi, gp);
    };
    auto gemm_res_ln = [&](GemmParams gp) -> hipError_t {  // out = LayerNorm(a W^T + bias + res)
        if (big && pgemm_supported(EPI_BIAS_RES, gp)) {
            hipError_t he = launch_pgemm(st, EPI_BIAS_RES, gp);
            if (he != hipSuccess) return he;
            return launch_ln_rows(st, gp.out, gp.ldo, gp.m, gp.n, gp.gamma, gp.beta, gp.eps);
        }
        return launch_gemm(st, EPI_BIAS_RES_LN, gp);
    };
    const bool small = e->small_pass && t_pad <= e->small_rows;
    for (const Layer &L : e->layers) {
        if (small) {
            // query-time passes (encoder_small.hip): projections by one wave per 32 features, the MLP split over the ffn chunks
            MX_HIP(launch_sp_qkv(st, e->x, L.wqkv, L.bqkv, t_pad, (int)rows, qscale, e->q, e->k, e->vt, t_pad));
            MX_HIP(launch_attention(st, e->q, e->k, e->vt, t_pad, e->attn_plan, B, heads, dh, H, max_len, e->ctx));
            MX_HIP(launch_sp_out_ln(st, e->ctx, e->x, L.wo, L.bo, L.ln1g, L.ln1b, c.ln_eps, t_pad, (int)rows, e->sp_x1));
            MX_HIP(launch_sp_ffn(st, e->sp_x1, L.wf, L.bi, F, t_pad, (int)rows, e->sp_part));
            MX_HIP(launch_sp_reduce_ln(st, e->sp_part, F, t_pad, (int)rows, L.bo2, e->sp_x1, L.ln2g, L.ln2b, c.ln_eps, e->x));
            continue;
        }
        // small passes of the hidden-768 models (no fused tail, no small-pass layer of their own): k is split, see below
        const bool split = e->split_small && e->sk_part && t_pad <= kSplitRows && H == 768 && F % 384 == 0;
        if (split && t_pad <= kQueryRows) {  // (beyond a few row tiles the partials and the transposed V stores cost more than the launch saved)
            // Q, K and V in ONE product over the concatenated weights, two k-chunks of f32 partials, then reduce_qkv_kernel
            // (bias, q scale, the V third transposed): two 24-k-tile latency chains become one of 12 on 3 x the workgroups
            GemmParams g3{};
            g3.a = e->x; g3.lda = H; g3.w = L.wqkv; g3.w_rows = 3 * H; g3.w_row0 = 0; g3.bias = e->sk_zero; g3.m = m_c; g3.n = 3 * H;
            g3.k = H / 2; g3.ksplit = 2; g3.out_f32 = e->sk_part; g3.ldo = 3 * H;
            MX_HIP(launch_gemm(st, EPI_F32, g3));
            MX_HIP(launch_reduce_qkv(st, e->sk_part, 2, m_c, H, L.bqkv, qscale, e->q, e->k, e->vt, t_pad));
        } else {
            GemmParams g{};
            g.a = e->x; g.lda = H; g.w = L.wqkv; g.w_rows = 3 * H; g.w_row0 = 0; g.bias = L.bqkv; g.m = m_c; g.n = 2 * H; g.k = H;
            g.out = e->q; g.out_k = e->k; g.ldo = H; g.hidden = H; g.qscale = qscale;
            MX_HIP(gemm(EPI_QKV, g));
            GemmParams gv{};  // V third of the concatenated projection, written feature-major
            gv.a = e->x; gv.lda = H; gv.w = L.wqkv; gv.w_rows = 3 * H; gv.w_row0 = 2 * H; gv.bias = L.bqkv + 2 * H; gv.m = m_c; gv.n = H;
            gv.k = H; gv.out_vt = e->vt; gv.ldvt = t_pad; gv.hidden = H;
            MX_HIP(gemm(EPI_VT, gv));
        }
        MX_HIP(launch_attention(st, e->q, e->k, e->vt, t_pad, e->attn_plan, B, heads, dh, H, max_len, e->ctx));
        if (e->fused_tail) {
            // out-projection + Add&Norm + MLP + Add&Norm in one kernel, in place on e->x
            TailParams tp{};
            tp.ctx = e->ctx; tp.ldc = H; tp.x = e->x; tp.ldx = H; tp.wf = L.wf; tp.bo = L.bo; tp.ln1g = L.ln1g; tp.ln1b = L.ln1b;
            tp.b1 = L.bi; tp.b2 = L.bo2; tp.f = F; tp.m = m_c; tp.out = e->x; tp.ldo = H; tp.gamma = L.ln2g; tp.beta = L.ln2b;
            tp.eps = c.ln_eps;
            MX_HIP(launch_tail(st, tp));
            continue;
        }
        GemmParams o{};
        o.a = e->ctx; o.lda = H; o.w = L.wo; o.w_rows = H; o.w_row0 = 0; o.bias = L.bo; o.m = m_c; o.n = H; o.k = H;
        o.out = e->x1; o.ldo = H; o.res = e->x; o.ldres = H; o.gamma = L.ln1g; o.beta = L.ln1b; o.eps = c.ln_eps;
        // small passes: product split over k into f32 partials (gemm_kernel<EPI_F32>, chunks of >= 384 columns), then
        // reduce + bias + residual + LayerNorm -- the rounding points of the fused epilogue, another f32 summation order
        auto split_res_ln = [&](GemmParams gp, int nsplit) -> hipError_t {
            const float *bias = gp.bias;
            bf16_t *out = gp.out;
            gp.k /= nsplit; gp.ksplit = nsplit; gp.bias = e->sk_zero; gp.out = nullptr; gp.out_f32 = e->sk_part; gp.ldo = gp.n;
            hipError_t he = launch_gemm(st, EPI_F32, gp);
            if (he != hipSuccess) return he;
            return launch_reduce_res_ln(st, e->sk_part, nsplit, m_c, gp.n, bias, gp.res, gp.ldres, gp.gamma, gp.beta, gp.eps, out, gp.n);
        };
        const int ns_o = H / 384, ns_2 = std::min(kSplitMax, F / 384);
        auto res_ln = [&](const GemmParams &gp, int nsplit) -> hipError_t {  // k must divide into chunks of whole k-tiles
            const bool ok = split && nsplit >= 2 && gp.k % nsplit == 0 && (gp.k / nsplit) % 32 == 0;
            return ok ? split_res_ln(gp, nsplit) : gemm_res_ln(gp);
        };
        MX_HIP(res_ln(o, ns_o));
        GemmParams f1{};
        f1.a = e->x1; f1.lda = H; f1.w = L.wi; f1.w_rows = F; f1.w_row0 = 0; f1.bias = L.bi; f1.m = m_c; f1.n = F; f1.k = H;
        f1.out = e->hbuf; f1.ldo = F;
        MX_HIP(gemm(EPI_BIAS_GELU, f1));
        GemmParams f2{};
        f2.a = e->hbuf; f2.lda = F; f2.w = L.wo2; f2.w_rows = H; f2.w_row0 = 0; f2.bias = L.bo2; f2.m = m_c; f2.n = H; f2.k = F;
        f2.out = e->x; f2.ldo = H; f2.res = e->x1; f2.ldres = H; f2.gamma = L.ln2g; f2.beta = L.ln2b; f2.eps = c.ln_eps;
        MX_HIP(res_ln(f2, ns_2));
    }
    MX_HIP(launch_pool(st, e->x, nullptr, e->cu, d_lens, B, H, c.pooling == MX_POOL_CLS, c.normalize, d_out));
    // no synchronisation here: the passes of one call queue up on the stream (same workspace, stream order)
    e->stats.sequences += (uint64_t)B;
    e->stats.tokens += tokens;
    e->stats.flops += (double)c.layers * ((double)tokens * (8.0 * H * H + 4.0 * (double)H * F) + attn_flops);
    return MX_OK;
}

// splits a batch into passes bounded by kMaxSeqsPerPass / kMaxRowsPerPass
int encode_all(mx_encoder *e, const int32_t *d_ids, const int32_t *h_lens, const int32_t *d_lens, int B, int S,
               float *d_out) {
    for (int b = 0; b < B; ++b)
        if (h_lens[b] < 1 || h_lens[b] > S) return fail(MX_EINVAL, "lens[%d] = %d outside [1, %d]", b, h_lens[b], S);
    // the pass split first: the workspace is sized ONCE for the largest pass before anything is launched (the
    // passes queue on the stream without synchronisation, so a later pass must never reallocate buffers an
    // earlier one is still using)
    std::vector<std::pair<int, int>> passes;  // (first sequence, count)
    const long pass_rows = kMaxRowsPerPass;  // (2^18 / 2^19 rows per pass: +0.3 ... +1.2 %, profiles/r5_pass_rows_ab.txt)
    long max_rows = 0;
    int max_nb = 0;
    for (int b0 = 0; b0 < B;) {
        int nb = 0;
        long rows = 0;
        while (b0 + nb < B && nb < kMaxSeqsPerPass) {
            const int l = std::min(std::max(h_lens[b0 + nb], 1), S);
            const long r = (l + kSeqAlign - 1) / kSeqAlign * kSeqAlign;
            if (nb > 0 && rows + r > pass_rows) break;
            rows += r;
            ++nb;
        }
        passes.push_back({b0, nb});
        max_rows = std::max(max_rows, rows);
        max_nb = std::max(max_nb, nb);
        b0 += nb;
    }
    int rc = ensure_ws(e, (int)round_up((uint64_t)max_rows + 32, kRowPad), max_nb, 0);
    if (rc != MX_OK) return rc;
    if (e->profiling) MX_HIP(hipEventRecord(e->ev0, e->stream));
    for (const auto &ps : passes) {
        rc = encode_pass(e, d_ids + (size_t)ps.first * S, h_lens + ps.first, d_lens + ps.first, ps.second, S,
                         d_out + (size_t)ps.first * e->cfg.hidden);
        if (rc != MX_OK) return rc;
    }
    if (e->profiling) MX_HIP(hipEventRecord(e->ev1, e->stream));
    // results complete in d_out when the call returns.  The thread naps meanwhile -- except through a query-sized call (one
    // small pass: 0.2-0.4 ms), where a 50 us nap granularity would be a fifth of the latency: those are polled through
    const bool query_sized = passes.size() == 1 && max_rows + 32 <= kQueryRows;
    MX_HIP(napping_sync(e->stream, e->ev_done, query_sized ? 1000 : 100));
    if (e->profiling) {
        float ms = 0.f;
        MX_HIP(hipEventElapsedTime(&ms, e->ev0, e->ev1));
        e->stats.gpu_ms += ms;
    }
    e->stats.calls += 1;
    return MX_OK;
}

int check_cfg(const mx_encoder_cfg *c) {
    if (!c) return fail(MX_EINVAL, "cfg is null");
    if (c->layers < 1 || c->layers > 48) return fail(MX_EUNSUPPORTED, "layers %d", c->layers);
    if (c->hidden != 384 && c->hidden != 768)
        return fail(MX_EUNSUPPORTED, "hidden %d: the fused LayerNorm GEMM is built for 384 and 768", c->hidden);
    if (c->heads < 1 || c->hidden % c->heads) return fail(MX_EINVAL, "heads %d does not divide hidden", c->heads);
    const int dh = c->hidden / c->heads;
    if (dh != 32 && dh != 64) return fail(MX_EUNSUPPORTED, "head dim %d (need 32 or 64)", dh);
    if (c->ffn < 384 || c->ffn % 384) return fail(MX_EUNSUPPORTED, "ffn %d must be a multiple of 384", c->ffn);
    if (c->vocab < 1 || c->max_pos < 1 || c->max_pos > 8192 || c->type_vocab < 1)
        return fail(MX_EINVAL, "vocab/max_pos/type_vocab out of range");
    if (c->pos_offset < 0 || c->pos_offset >= c->max_pos) return fail(MX_EINVAL, "pos_offset %d outside [0, max_pos)", c->pos_offset);
    if (c->pooling != MX_POOL_MEAN && c->pooling != MX_POOL_CLS) return fail(MX_EINVAL, "pooling %d", c->pooling);
    if (!(c->ln_eps >= 0.0f)) return fail(MX_EINVAL, "ln_eps");
    if (c->precision != MX_PREC_BF16 && c->precision != MX_PREC_BF16X3 && c->precision != MX_PREC_MIXED && c->precision != MX_PREC_MIXED1)
        return fail(MX_EINVAL, "precision %d", c->precision);
    if (c->precision != MX_PREC_BF16 && c->ffn % 192) return fail(MX_EUNSUPPORTED, "ffn %d: the split-operand modes need a multiple of 192", c->ffn);
    return MX_OK;
}

}  // namespace

static void destroy_impl(mx_encoder *e);

extern "C" {

size_t mx_encoder_cfg_size(void) { return sizeof(mx_encoder_cfg); }

size_t mx_encoder_weight_bytes(const mx_encoder_cfg *c) {
    if (!c) return 0;
    const size_t H = (size_t)c->hidden, F = (size_t)c->ffn;
    size_t n = ((size_t)c->vocab + c->max_pos + c->type_vocab) * H + 2 * H;
    n += (size_t)c->layers * (3 * (H * H + H) + (H * H + H) + 2 * H + (F * H + F) + (H * F + H) + 2 * H);
    return n * sizeof(float);
}

int mx_encoder_create(const mx_encoder_cfg *cfg, const void *weights, size_t nbytes, int device, mx_encoder **out) try {
    if (!out) return fail(MX_EINVAL, "out is null");
    *out = nullptr;
    int rc = check_cfg(cfg);
    if (rc != MX_OK) return rc;
    if (!weights) return fail(MX_EINVAL, "weights is null");
    if (nbytes != mx_encoder_weight_bytes(cfg))
        return fail(MX_EINVAL, "weight blob is %zu bytes, config needs %zu", nbytes, mx_encoder_weight_bytes(cfg));
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0)
        return fail(MX_EDEVICE, "no HIP device available (%s)", he == hipSuccess ? "count 0" : hipGetErrorString(he));
    if (device < 0 || device >= ndev) return fail(MX_EDEVICE, "device %d out of range (have %d)", device, ndev);
    DeviceGuard g(device);
    if (!g.ok) return fail(MX_EDEVICE, "hipSetDevice(%d) failed", device);
    {
        std::lock_guard<std::mutex> lk(g_enc_setup_mu);
        if (device >= 64 || !g_enc_setup_done[device]) {
            hipError_t se = encoder_kernels_setup();
            if (se == hipSuccess) se = pgemm_setup();
            if (se == hipSuccess) se = tail_setup();
            if (se == hipSuccess) se = precise_setup();
            if (se == hipSuccess) se = small_setup();
            if (se != hipSuccess) return fail(MX_EDEVICE, "encoder kernel setup failed: %s", hipGetErrorString(se));
            if (device < 64) g_enc_setup_done[device] = true;
        }
    }

    mx_encoder *e = new mx_encoder();
    e->cfg = *cfg;
    e->device = device;
    {
        // kernel-variant keys of MEMEX_HIP_DEBUG (mx_debug.h), fixed per encoder handle
        e->precise = cfg->precision != MX_PREC_BF16;
        e->mixed = cfg->precision == MX_PREC_MIXED || cfg->precision == MX_PREC_MIXED1;
        e->mlp1 = cfg->precision == MX_PREC_MIXED1;
        e->fused_tail = !e->precise && tail_supported(cfg->hidden, cfg->ffn) && debug_flag("unfused_tail", 0) != 1;
        e->pgemm = debug_flag("pgemm", 1) != 0;
        e->small_pass = e->fused_tail && debug_flag("small", 1) != 0;
        e->attn_f32 = debug_flag("attn_f32", 0) == 1;
    }
    auto bail = [&](int code) {
        destroy_impl(e);
        return code;
    };
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_wait, hipEventDisableTiming) != hipSuccess)
        return bail(fail(MX_EDEVICE, "stream/event creation failed"));

    const size_t H = (size_t)cfg->hidden, F = (size_t)cfg->ffn;
    const float *p = static_cast<const float *>(weights);
    auto take = [&](size_t n) {
        const float *r = p;
        p += n;
        return r;
    };
#define MX_TRY(x)                      \
    do {                               \
        int _rc = (x);                 \
        if (_rc != MX_OK) return bail(_rc); \
    } while (0)
    MX_TRY(upload_f32(e, take((size_t)cfg->vocab * H), (size_t)cfg->vocab * H, &e->word));
    MX_TRY(upload_f32(e, take((size_t)cfg->max_pos * H), (size_t)cfg->max_pos * H, &e->pos));
    e->pos += (size_t)cfg->pos_offset * H;  // RoBERTa-style tables: token t of a sequence uses row pos_offset + t
    {
        const float *ty = take((size_t)cfg->type_vocab * H);
        MX_TRY(upload_f32(e, ty, H, &e->type0));  // token_type 0 only (single-segment inputs)
    }
    MX_TRY(upload_f32(e, take(H), H, &e->eg));
    MX_TRY(upload_f32(e, take(H), H, &e->eb));
    e->layers.resize(cfg->layers);
    for (Layer &L : e->layers) {
        std::vector<float> wqkv(3 * H * H), bqkv(3 * H);
        for (int part = 0; part < 3; ++part) {
            memcpy(wqkv.data() + part * H * H, take(H * H), H * H * sizeof(float));
            memcpy(bqkv.data() + part * H, take(H), H * sizeof(float));
        }
        if (e->precise) MX_TRY(upload_weight3(e, wqkv.data(), 3 * H, H, &L.wqkv3));
        else MX_TRY(upload_weight(e, wqkv.data(), 3 * H, H, &L.wqkv));
        MX_TRY(upload_f32(e, bqkv.data(), 3 * H, &L.bqkv));
        const float *wo_src = take(H * H);
        if (e->precise) MX_TRY(upload_weight3(e, wo_src, H, H, &L.wo3));
        else MX_TRY(upload_weight(e, wo_src, H, H, &L.wo));
        const float *bo_src = take(H), *g1_src = take(H), *be1_src = take(H);
        MX_TRY(upload_f32(e, bo_src, H, &L.bo));
        MX_TRY(upload_f32(e, g1_src, H, &L.ln1g));
        MX_TRY(upload_f32(e, be1_src, H, &L.ln1b));
        const float *wi_src = take(F * H);
        if (e->precise && !e->mixed) MX_TRY(upload_weight3(e, wi_src, F, H, &L.wi3));
        if (e->mixed) MX_TRY(upload_weight2h(e, wi_src, F, H, &L.wi2h, e->mlp1 ? 1 : 2));
        else MX_TRY(upload_weight(e, wi_src, F, H, &L.wi));
        const float *b1_src = take(F);
        MX_TRY(upload_f32(e, b1_src, F, &L.bi));
        const float *wo2_src = take(H * F);
        if (e->precise && !e->mixed) MX_TRY(upload_weight3(e, wo2_src, H, F, &L.wo23));
        if (e->mixed) MX_TRY(upload_weight2h(e, wo2_src, H, F, &L.wo22h, e->mlp1 ? 1 : 2));
        else MX_TRY(upload_weight(e, wo2_src, H, F, &L.wo2));
        if (e->fused_tail) MX_TRY(upload_tail_stream(e, wo_src, wi_src, wo2_src, F, &L.wf));
        const float *b2_src = take(H), *g2_src = take(H), *be2_src = take(H);
        MX_TRY(upload_f32(e, b2_src, H, &L.bo2));
        MX_TRY(upload_f32(e, g2_src, H, &L.ln2g));
        MX_TRY(upload_f32(e, be2_src, H, &L.ln2b));
    }
    if (!e->fused_tail && !e->precise && H == 768) {
        e->split_small = debug_flag("splitk", 1) != 0 && debug_flag("small", 1) != 0;  // (small=0: one kernel set at every pass size)
        if (e->split_small) {
            void *pp = nullptr, *pz = nullptr;
            if (hipMalloc(&pp, (size_t)kSplitMax * kSplitRows * H * sizeof(float)) != hipSuccess || hipMalloc(&pz, (size_t)3 * H * sizeof(float)) != hipSuccess ||
                hipMemset(pz, 0, (size_t)3 * H * sizeof(float)) != hipSuccess) {
                if (pp) (void)hipFree(pp);
                if (pz) (void)hipFree(pz);
                return bail(fail(MX_ENOMEM, "hipMalloc(split-k workspace) failed"));
            }
            e->allocs.push_back(pp);
            e->allocs.push_back(pz);
            e->sk_part = static_cast<float *>(pp);
            e->sk_zero = static_cast<float *>(pz);
        }
    }
    if (e->small_pass) {
        void *px = nullptr, *pp = nullptr;
        {   // where the small-pass layer hands over to the bulk kernels (a multiple of 64)
            const int v = debug_flag("small_rows", 0);
            if (v >= 64 && v <= 16384) e->small_rows = v / 64 * 64;
        }
        if (hipMalloc(&px, (size_t)e->small_rows * H * sizeof(uint16_t)) != hipSuccess ||
            hipMalloc(&pp, (size_t)(F / 128) * e->small_rows * H * sizeof(float)) != hipSuccess) {
            if (px) (void)hipFree(px);
            return bail(fail(MX_ENOMEM, "hipMalloc(small-pass workspace) failed"));
        }
        e->allocs.push_back(px);
        e->allocs.push_back(pp);
        e->sp_x1 = static_cast<bf16_t *>(px);
        e->sp_part = static_cast<float *>(pp);
    }
#undef MX_TRY
    *out = e;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

static void destroy_impl(mx_encoder *e) {
    if (!e) return;
    DeviceGuard g(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (void *p : e->allocs) (void)hipFree(p);
    free_ws(e);
    void *ptrs[] = {e->cu, e->lens_dev, e->ids_dev, e->out_dev, e->attn_plan};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->ev_done) (void)hipEventDestroy(e->ev_done);
    if (e->ev_wait) (void)hipEventDestroy(e->ev_wait);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->h_io) (void)hipHostFree(e->h_io);
    delete e;
}

void mx_encoder_destroy(mx_encoder *e) {
    if (!e) return;
    {
        std::lock_guard<std::mutex> lk(g_enc_reg_mu);
        if (--e->refs > 0) return;
        if (!e->key.empty()) g_enc_registry.erase(e->key);
    }
    {   // nobody may still be inside a call on this handle
        std::lock_guard<std::mutex> lk(e->mu);
    }
    destroy_impl(e);
}

// Replaces the per-request / per-task `create_model()` (the reference spawns an embedder, i.e. loads the
// checkpoint, for every HTTP search and every ingest task: handlers.rs:61-63, tasks.rs:17): the first
// open of `key` uploads the weights, later opens return the SAME resident encoder (ref-counted; calls
// on it are serialised inside).  weights may be NULL when the key is expected to be resident.
int mx_encoder_open(const char *key, const mx_encoder_cfg *cfg, const void *weights, size_t nbytes, int device,
                    mx_encoder **out) try {
    if (!out) return fail(MX_EINVAL, "out is null");
    *out = nullptr;
    const std::string k = key ? key : "";
    if (k.empty()) return mx_encoder_create(cfg, weights, nbytes, device, out);
    std::lock_guard<std::mutex> lk(g_enc_reg_mu);
    auto it = g_enc_registry.find(k);
    if (it != g_enc_registry.end()) {
        mx_encoder *e = it->second;
        if (cfg && memcmp(cfg, &e->cfg, sizeof(*cfg)) != 0) return fail(MX_EINVAL, "encoder '%s' is resident with another configuration", k.c_str());
        if (e->device != device) return fail(MX_EINVAL, "encoder '%s' lives on device %d, not %d", k.c_str(), e->device, device);
        e->refs += 1;
        *out = e;
        return MX_OK;
    }
    if (!weights) return fail(MX_EINVAL, "encoder '%s' is not resident and no weights were given", k.c_str());
    mx_encoder *e = nullptr;
    int rc = mx_encoder_create(cfg, weights, nbytes, device, &e);
    if (rc != MX_OK) return rc;
    e->key = k;
    g_enc_registry[k] = e;
    *out = e;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_encoder_wait_stream(mx_encoder *e, void *stream) try {
    if (!e) return fail(MX_EINVAL, "null encoder");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    MX_HIP(hipEventRecord(e->ev_wait, static_cast<hipStream_t>(stream)));
    MX_HIP(hipStreamWaitEvent(e->stream, e->ev_wait, 0));
    return MX_OK;
} catch (...) {
    return guard_exception();
}

static int check_call(mx_encoder *e, const void *ids, const void *lens, int B, int S, const void *out) {
    if (!e) return fail(MX_ESEARCH, "null encoder");
    if (B < 0 || S < 1) return fail(MX_EINVAL, "bad batch shape B=%d S=%d", B, S);
    if (S > 512) return fail(MX_EUNSUPPORTED, "S=%d: sequences are limited to 512 tokens", S);
    if (S + e->cfg.pos_offset > e->cfg.max_pos)
        return fail(MX_EINVAL, "S=%d (+ pos_offset %d) exceeds max_pos=%d", S, e->cfg.pos_offset, e->cfg.max_pos);
    if (B > 0 && (!ids || !lens || !out)) return fail(MX_EINVAL, "null argument");
    return MX_OK;
}

int mx_encoder_encode_device(mx_encoder *e, const int32_t *d_ids, const int32_t *d_lens, int B, int S, float *d_out) try {
    int rc = check_call(e, d_ids, d_lens, B, S, d_out);
    if (rc != MX_OK || B == 0) return rc;
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    // the lengths decide the pass split and the workspace size on the host: fetch them ON the encoder's stream,
    // i.e. behind whatever mx_encoder_wait_stream ordered in front of this call (a blocking copy on the null
    // stream does not wait for a caller's non-blocking stream and could read the lengths before they exist)
    std::vector<int32_t> h_lens((size_t)B);
    MX_HIP(hipMemcpyAsync(h_lens.data(), d_lens, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    MX_HIP(hipStreamSynchronize(e->stream));
    return encode_all(e, d_ids, h_lens.data(), d_lens, B, S, d_out);
} catch (...) {
    return guard_exception();
}

int mx_encoder_encode(mx_encoder *e, const int32_t *ids, const int32_t *lens, int B, int S, float *out) try {
    int rc = check_call(e, ids, lens, B, S, out);
    if (rc != MX_OK || B == 0) return rc;
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    // Query-sized calls skip the three copy commands: ids and lengths are placed in a pinned, device-mapped page that the
    // kernels read in place (68 bytes for a 16-token query), and pool_kernel writes the embeddings into that page, visible to
    // the host when the completion event is -- two H2D copies from pageable memory in front of the first kernel and a blocking
    // D2H copy behind the last one were ~40 us of a 0.27 ms call.
    if ((size_t)B * S <= kIoIds && B <= kIoSeqs) {
        if (!e->h_io) {
            MX_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->h_io), kIoBytes(e->cfg.hidden), hipHostMallocMapped | hipHostMallocCoherent));
        }
        rc = ensure_ws(e, 0, B, 0);
        if (rc != MX_OK) return rc;
        int32_t *io_ids = reinterpret_cast<int32_t *>(e->h_io), *io_lens = io_ids + kIoIds;
        float *io_out = reinterpret_cast<float *>(io_lens + kIoSeqs);
        memcpy(io_ids, ids, (size_t)B * S * sizeof(int32_t));
        memcpy(io_lens, lens, (size_t)B * sizeof(int32_t));
        rc = encode_all(e, io_ids, lens, io_lens, B, S, io_out);
        if (rc != MX_OK) return rc;
        memcpy(out, io_out, (size_t)B * e->cfg.hidden * sizeof(float));
        return MX_OK;
    }
    rc = ensure_ws(e, 0, B, B * S);
    if (rc != MX_OK) return rc;
    MX_HIP(hipMemcpyAsync(e->ids_dev, ids, (size_t)B * S * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    MX_HIP(hipMemcpyAsync(e->lens_dev, lens, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    rc = encode_all(e, e->ids_dev, lens, e->lens_dev, B, S, e->out_dev);
    if (rc != MX_OK) return rc;
    MX_HIP(hipMemcpy(out, e->out_dev, (size_t)B * e->cfg.hidden * sizeof(float), hipMemcpyDeviceToHost));
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_encoder_set_profiling(mx_encoder *e, int on) try {
    if (!e) return fail(MX_EINVAL, "null encoder");
    std::lock_guard<std::mutex> lk(e->mu);
    e->profiling = on != 0;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_encoder_get_stats(mx_encoder *e, mx_encoder_stats *out) try {
    if (!e || !out) return fail(MX_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    *out = e->stats;
    return MX_OK;
} catch (...) {
    return guard_exception();
}

int mx_encoder_reset_stats(mx_encoder *e) try {
    if (!e) return fail(MX_EINVAL, "null encoder");
    std::lock_guard<std::mutex> lk(e->mu);
    e->stats = mx_encoder_stats{};
    return MX_OK;
} catch (...) {
    return guard_exception();
}

}  // extern "C"
