/*
 * memex_hip.h -- C ABI of libmemex_hip.so: the MI355X (gfx950) embedding + vector-search path
 * that drops in under memex's `VectorStore` trait and `SentenceEmbedder` actor.
 *
 * The reference (spyglass-search/memex, Rust) has no FFI layer; the seam is the Rust surface of
 * lib/libmemex.  Every entry point below names the reference item it replaces (paths relative to
 * the reference root).  INTEGRATION.md shows the ~100-line Rust shim (`extern "C"` block +
 * `impl VectorStore for HipFlatStore`) a maintainer would add.
 *
 * Conventions
 *   - every function returns MX_OK (0) or a negative mx_status; mx_last_error() is thread-local.
 *   - nothing aborts or throws across this boundary (contrast: the reference panics at
 *     storage/local.rs:83 and :31): every entry point is a function-try-block, a C++ exception inside
 *     the library (std::bad_alloc -> MX_ENOMEM, anything else -> MX_EDEVICE "internal error: ...")
 *     comes back as a status code; helper threads and the leader of a combined search hand their
 *     exceptions to the callers they serve the same way.
 *   - the caller owns every input buffer (copied/consumed before return) and every output buffer.
 *   - handles are safe to use from several threads; calls on one handle are serialised inside.
 *   - "_device" variants take pointers into the HBM of the handle's device (for callers that keep
 *     data resident: the encoder output feeding mx_index_add_device, benchmarks, multi-GPU merge).
 *
 * Environment (everything the library reads; there is no fault-injection switch -- the two hooks the
 * tests need exist only in libmemex_hip_testing.so, built with -DMEMEX_TESTING):
 *   MEMEX_HIP_SPIN=1            host waits poll the completion word from the start (a benchmark owns its
 *                               core); default: sleep for most of the expected batch time, then poll
 *   MEMEX_HIP_EXCHANGE=rccl|p2p sharded index: insist on the RCCL all-gather / force peer copies
 *                               (default: RCCL when every shard has its own device)
 *   MEMEX_HIP_RCCL_TIMEOUT=s    deadline of ncclCommInitAll and of the communicator self-test (30)
 *   MEMEX_HIP_SHARD_THREADS=0|1 per-shard helper threads never / also for logical shards on one device
 *   MEMEX_HIP_FILTER=i8|bf16    pins the filter copy of every index opened afterwards (mx_index_set_filter_copy)
 *   MEMEX_HIP_DEBUG=key=v,...   which of two equivalent kernel forms runs (parity tests; keys in
 *                               memex_amd/csrc/mx_debug.h).  Results do not depend on it beyond the bounds
 *                               stated there (bit-identical forms, or different f32 summation orders).
 */
#ifndef MEMEX_HIP_H
#define MEMEX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: map 1:1 onto VectorStoreError (lib/libmemex/src/storage/mod.rs:31-48) and
 *      EmbeddingError (lib/libmemex/src/llm/embedding.rs:11-16) in the Rust shim ------------- */
typedef enum mx_status {
    MX_OK = 0,
    MX_EINVAL = -1,       /* bad argument                    -> InsertionError / SearchError      */
    MX_EDEVICE = -2,      /* HIP runtime / no device         -> ConnectionError / SetupError      */
    MX_EINSERT = -3,      /* insert failed                   -> InsertionError                    */
    MX_ESEARCH = -4,      /* search failed                   -> SearchError / EncodingFailure     */
    MX_EIO = -5,          /* save / load / delete files      -> FileIOError / SaveError           */
    MX_EUNSUPPORTED = -6, /* unsupported config              -> Unsupported / SetupError          */
    MX_ENOMEM = -7        /* host or device allocation       -> InsertionError / SetupError       */
} mx_status;

const char *mx_last_error(void); /* thread-local message of the last failing call on this thread */
const char *mx_version(void);
int mx_device_count(int *n_devices);

/* =====================================================================================
 * Flat cosine index  (replaces HnswStore, lib/libmemex/src/storage/local.rs:21-166)
 * ===================================================================================== */
typedef struct mx_index mx_index;

/*
 * Open (or attach to) the GPU-resident index registered under `key`.
 * Replaces HnswStore::new / the per-request construction in get_vector_storage
 * (storage/mod.rs:95-121, storage/local.rs:95-108): callers there build a store per request, so
 * a second open of the same key returns the SAME resident index (ref-counted, O(1)).
 * key == NULL or "" creates a private, unregistered index.  dim >= 1 (any; rows are padded
 * internally).  device = HIP device ordinal.
 */
int mx_index_open(const char *key, int dim, int device, mx_index **out);
void mx_index_close(mx_index *idx); /* drops one reference; frees HBM when the last one goes */

/*
 * Multi-GPU index inside ONE process (SURVEY.md section 8b `n_dev`, 8e): the Rust host is a single
 * process (bin/memex/src/main.rs spawns api + worker as tasks), so the 8-way shard of BASELINE
 * configs[3] has to live behind this handle.  Rows are dealt to the n_dev shards in blocks of
 * block_rows consecutive rows (0 = 65536; rounded up to 32): global row r lives on shard
 * (r / block_rows) % n_dev, so an append-only collection stays balanced without knowing its final
 * size.  devices[g] = HIP ordinal of shard g (NULL = 0 .. n_dev-1); ordinals may repeat (logical
 * shards on one GPU: how the sharded path is tested on a 1-GPU box).
 * A search runs the local scan of every shard concurrently (one host thread + one stream per
 * device), exchanges the per-shard top-k blocks ([ids | dists], B*k*12 bytes each) with ONE RCCL
 * all-gather over xGMI (librccl is dlopen'ed on first use; peer copies into a slot per shard when the
 * shards share a device or MEMEX_HIP_EXCHANGE=p2p) and merges by (dist, id) on devices[0].
 * Results are bit-identical to the unsharded index for every n_dev.
 * Every mx_index_* function accepts the returned handle; *_device pointers refer to devices[0];
 * vectors.mxflat written by mx_index_save does not depend on n_dev.
 */
int mx_index_open_sharded(const char *key, int dim, int n_dev, const int *devices, uint64_t block_rows,
                          mx_index **out);
int mx_index_n_shards(mx_index *idx, int *n_shards); /* 1 for a plain index */
/* How the shards exchange their per-shard top-k blocks: 0 = plain index (nothing to exchange), 1 = copies into
 * a slot per shard on devices[0] (peer-to-peer between devices), 2 = one RCCL all-gather. */
int mx_index_exchange(mx_index *idx, int *kind);

/*
 * Stream contract of the *_device entry points.  The index works on its own HIP stream and every
 * call returns only after its results are complete in HBM (host-synchronous), so the caller may read
 * outputs from any stream afterwards.  INPUTS (and zero-fills of output buffers) enqueued on a
 * caller stream are NOT ordered against the index's stream by themselves: either synchronise that
 * stream first, or call mx_index_wait_stream(idx, stream) -- the next operation on the index then
 * starts after everything enqueued on `stream` (a hipStream_t; NULL = the default stream) so far.
 */
int mx_index_wait_stream(mx_index *idx, void *hip_stream);

int mx_index_dim(mx_index *idx, int *dim);
int mx_index_size(mx_index *idx, uint64_t *n_rows); /* = _id_map.len(), storage/local.rs:63 */
int mx_index_reserve(mx_index *idx, uint64_t n_rows); /* pre-size HBM (optional)             */

/*
 * Row-sharding support: ids returned by search are
 * `id_offset + local_row + 1`.  Default 0 = the reference's dense 1-based ids.
 */
int mx_index_set_id_offset(mx_index *idx, uint64_t id_offset);

/*
 * Append n rows ([n, dim] row-major f32).  Replaces HnswStore::insert / bulk_insert
 * (storage/local.rs:55-69): ids are dense, 1-based, in insertion order; *first_id receives the id
 * of rows[0] (later rows follow consecutively).  Non-finite values are rejected (MX_EINVAL) and
 * nothing is inserted.  (Finite values whose f32 products overflow -- elements beyond ~1.8e19 in a row or
 * a query -- take DistCosine into inf / inf = NaN, where hnsw_rs asserts, i.e. the reference panics;
 * here such a call returns, rows at a NaN distance are left out, everything that has a defined distance
 * is ranked by it: tests/test_centred_gpu.py.)  Unlike the reference there is no save-per-insert (local.rs:67); call
 * mx_index_save.
 */
int mx_index_add(mx_index *idx, const float *rows, uint64_t n, uint64_t *first_id);
int mx_index_add_device(mx_index *idx, const float *d_rows, uint64_t n, uint64_t *first_id);

/* Replaces HnswStore::delete_all (storage/local.rs:34-53): drops every row, ids restart at 1.
 * (Removing the persisted files is mx_index_remove_files; the shim calls both.) */
int mx_index_clear(mx_index *idx);

/*
 * Top-k cosine search.  Replaces HnswStore::search (storage/local.rs:71-91) + hnsw_rs DistCosine:
 *   dist  = max(0, 1 - sum_f64(fl32(q_i*c_i)) / sqrt(sum_f64(fl32(q_i^2)) * sum_f64(fl32(c_i^2)))) as f32
 *           (0 when either norm is 0)
 *   score = 1.0f - (1.0f / (1.0f / dist))                                  (local.rs:86)
 * Results per query are ordered by (dist ascending, id ascending) -- exact brute force, so recall
 * is 1 by construction.  Outputs are row-major [B, k]; n_found[b] = min(k, size); unused slots
 * hold id 0 / score 0 / dist +inf.  `dists` may be NULL.  B > 1 is an extension (the trait is
 * single-query); B is processed in batches of up to 512.  Thread-safe, and concurrent calls are COMBINED:
 * requests that arrive while a pass is on the GPU are served together (up to 512 queries with the
 * same k per pass, FIFO), which is what turns the reference's one-query-per-HTTP-request pattern
 * (api/handlers.rs:55-109) into full batches without touching the trait.  k <= 4096 (MX_EUNSUPPORTED above; the
 * reference's callers pass limit = 10, api/handlers.rs search default).
 */
int mx_index_search(mx_index *idx, const float *queries, int B, int k, uint64_t *ids, float *scores,
                    float *dists, int32_t *n_found);
int mx_index_search_device(mx_index *idx, const float *d_queries, int B, int k, uint64_t *d_ids,
                           float *d_scores, float *d_dists, int32_t *d_n_found);

/* Search strategy (testing / diagnostics).  AUTO = low-precision MFMA streaming scan (int8 or bf16
 * filter copy, or the f32 rows) that certifies a candidate superset, f32 then exact f64 rescoring of the
 * candidates, per-query fallback to EXACT when a candidate buffer overflows twice.  EXACT = f64 arithmetic on every row (slow, always available). */
enum { MX_SEARCH_AUTO = 0, MX_SEARCH_EXACT = 1 };
int mx_index_set_search_mode(mx_index *idx, int mode);

/* Filter copy.  By default the index keeps, next to the f32 rows, a low-precision copy of them laid out
 * for the MFMA scan; the AUTO scan streams that copy instead of the f32 rows and candidates are rescored
 * from the f32 rows, so results are bit-identical with every kind and without one.
 *   on = 0  no copy (the scan reads the f32 rows: 4 bytes per element streamed)
 *   on = 1  a copy whose kind the library chooses (the default): int8 up to 1024 dims, bf16 above; an int8
 *           copy is rebuilt as bf16 -- once, from the f32 rows -- when the corpus proves too dense for its
 *           certificate (more than 1/16 of a batch overflows the int8 pass, or the retry pass has become
 *           habitual) and built again as int8 once the collection has doubled since (or when this call is
 *           repeated with on = 1); mx_index_stats.filter_kind / filter_demotions / filter_promotions tell
 *   on = 2  int8 copy (+25 % HBM: rows*dim_pad bytes + 16 bytes per 64 rows): one quantisation step and one
 *           measured residual bound per 32 rows, exact integer sums; the certificate is 4-5x wider than
 *           bf16's, so finish_kernel sifts a few hundred candidates per query instead of a few dozen
 *   on = 3  bf16 copy (+50 % HBM: rows*dim_pad*2 bytes)
 * (Re)builds from the resident rows when the kind changes.  If HBM for the copy cannot be allocated while
 * the index grows, it is dropped silently and the index continues on the f32 scan.  Environment:
 * MEMEX_HIP_FILTER=i8|bf16 pins the kind for every index opened afterwards. */
int mx_index_set_filter_copy(mx_index *idx, int on);

/*
 * Corpus mode (compressed corpus).  MX_CORPUS_F32 (default): the f32 rows are
 * kept and results are bit-identical to the reference's arithmetic on them.  MX_CORPUS_BF16: the index
 * keeps ONLY bf16(c / |c|) -- 2 bytes per element instead of 4 (+2 for the filter copy), i.e. a third
 * of the default footprint: 7.7 GB for 10M x 384, ~110M x 384-d rows per 288 GB GPU.  Searches are then
 * EXACT with respect to the stored (rounded) rows: same pipeline, the rescoring stages read the
 * stored rows, and the answer is bit-identical to the reference's arithmetic applied to
 * mx_index_get_rows() -- each cosine is within ~2e-3 of the f32 one, so recall@10 against the f32
 * corpus stays ~0.99 while the scan is unchanged.  Choose the mode while the index is empty.
 * dim <= 1536.  mx_index_save marks such a store in the file header; loading it into an index of
 * either mode reproduces the stored rows exactly.
 */
enum { MX_CORPUS_F32 = 0, MX_CORPUS_BF16 = 1 };
int mx_index_set_corpus_mode(mx_index *idx, int mode);
/* Rows [first_row, first_row + n) (0-based, insertion order) as stored, into out [n, dim] f32: the
 * inserted values (F32 mode) or the stored near-unit bf16 values widened to f32 (BF16 mode; zero-norm
 * rows come back as zeros).  Also how a collection is re-exported / rebuilt elsewhere. */
int mx_index_get_rows(mx_index *idx, uint64_t first_row, uint64_t n, float *out);

/*
 * Persistence.  Replaces HnswStore::save / load / has_store (storage/local.rs:110-165).  Files in
 * `dir`: `vectors.mxflat` (header + raw f32 rows).  The string-id map `vectors.meta.json`
 * (local.rs:19,156-163) stays on the Rust side unchanged.
 */
int mx_index_save(mx_index *idx, const char *dir);
int mx_index_load(mx_index *idx, const char *dir); /* replaces current contents */
int mx_index_has_store(const char *dir, int *exists);
int mx_index_store_info(const char *dir, int *dim, uint64_t *n_rows); /* header of vectors.mxflat */
int mx_index_remove_files(const char *dir);        /* the file half of delete_all, local.rs:36-46 */

/* Counters for the bench / roofline report (cumulative since open or last reset). */
typedef struct mx_index_stats {
    uint64_t searches;          /* query batches served                                   */
    uint64_t queries;           /* queries served                                          */
    uint64_t fallback_queries;  /* queries answered by the EXACT path (rescan overflowed too) */
    uint64_t scan_launches;     /* launches of the main streaming-scan kernel              */
    uint64_t scan_bytes;        /* bytes those launches streamed: rows*dim_pad*(1 int8 copy, 2 bf16 copy, 4 f32 rows) */
    double scan_ms;             /* HIP-event time of those launches (profiling on)         */
    uint64_t candidates;        /* candidates that passed the filter and were rescored in f32 */
    double max_abs_err;         /* profiling only: max |approx - exact| cosine on candidates */
    uint64_t filter_copy_bytes; /* HBM held by the filter copy (0 = scanning the f32 rows) */
    uint64_t retry_queries;     /* queries rescanned once with a tightened threshold (lane buffer overflow) */
    double approx_err_bound;    /* largest per-query bound e1 on |filter score - cosine| of the last batch */
    uint64_t filter_kind;       /* what the scan streams now: 0 = the f32 rows, 2 = int8 filter copy, 3 = bf16 filter copy */
    uint64_t filter_demotions;  /* times an automatically chosen int8 copy was rebuilt as bf16 (dense corpus)   */
    uint64_t filter_promotions; /* times a demoted index got its int8 copy back (the collection had doubled since)  */
    uint64_t listed_rows;       /* rows on the side list finish_kernel adds to every query: zero norm, or a norm outside [1e-15, 1e15] */
    uint64_t exchange_fallbacks;/* sharded index: times the RCCL exchange failed at run time and peer copies took over  */
    uint64_t filter_centred;    /* 1 = the bf16 copy holds the rows minus their component along the corpus mean direction
                                   (a rebuilt copy of a corpus that sits in a cone: a several times tighter certificate)   */
    double exchange_ms;         /* sharded index: host time from "every shard has answered" to "merged result complete" (the
                                   all-gather or peer copies' tail, merge_kernel, n_found, one synchronise): the serial tail of
                                   a step that the shards' own work does not hide                                           */
} mx_index_stats;
/* sizeof(mx_index_stats) of the library that is loaded: a shim compares it with its own at start-up (the struct grows
 * at the end from version to version; mx_version() names the release). */
size_t mx_index_stats_size(void);
int mx_index_set_profiling(mx_index *idx, int on); /* record HIP events around the scan kernel */
int mx_index_get_stats(mx_index *idx, mx_index_stats *out);
int mx_index_reset_stats(mx_index *idx);

/*
 * Multi-GPU merge: after an RCCL all-gather of per-shard results
 * ([G, B, k] ids + dists, each shard list ordered), produce the global top-k ordered by
 * (dist, id) and the scores.  Pointers are device pointers on `device`.
 */
int mx_topk_merge_device(int device, const uint64_t *d_ids, const float *d_dists, int G, int B, int k,
                         uint64_t *d_out_ids, float *d_out_dists, float *d_out_scores);
/* Same merge for ONE all-gather: every shard writes its results into one block
 * [ids: B*k u64][dists: B*k f32] (pass `block` and `block + B*k*8` to mx_index_search_device), the
 * collective gathers the G blocks back to back into d_packed. */
int mx_topk_merge_packed_device(int device, const void *d_packed, int G, int B, int k, uint64_t *d_out_ids,
                                float *d_out_dists, float *d_out_scores);
/* The same merge ENQUEUED on the caller's stream (a hipStream_t; NULL = default stream), returning
 * at once: put it on the stream the all-gather was issued on and no host round trip separates the
 * collective from the merge. */
int mx_topk_merge_packed_async(int device, void *hip_stream, const void *d_packed, int G, int B, int k,
                               uint64_t *d_out_ids, float *d_out_dists, float *d_out_scores);

/* =====================================================================================
 * Sentence encoder  (replaces the rust-bert model owned by SentenceEmbedder::runner,
 * lib/libmemex/src/llm/embedding.rs:94-135; `model.encode(&segments)` at :109)
 * ===================================================================================== */
typedef struct mx_encoder mx_encoder;

enum { MX_POOL_MEAN = 0, MX_POOL_CLS = 1 };

typedef struct mx_encoder_cfg {
    int32_t layers;     /* 6  (all-MiniLM-L6-v2) / 12 (all-MiniLM-L12-v2, bge-base-en)      */
    int32_t hidden;     /* 384 / 768; multiple of 64                                         */
    int32_t heads;      /* 12; hidden/heads must be 32 or 64                                 */
    int32_t ffn;        /* 1536 / 3072; multiple of 64                                       */
    int32_t vocab;      /* 30522                                                              */
    int32_t max_pos;    /* rows of the position table: 512 (BERT) / 514 (RoBERTa); sequences <= 512 tokens */
    int32_t type_vocab; /* 2                                                                  */
    float ln_eps;       /* 1e-12                                                              */
    int32_t pooling;    /* MX_POOL_MEAN (MiniLM) | MX_POOL_CLS (bge)                          */
    int32_t normalize;  /* 1 = L2-normalise the pooled vector                                 */
    int32_t pos_offset; /* row of the position table used by a sequence's first token: 0 for BERT
                           (MiniLM, bge); 2 for RoBERTa-style checkpoints such as all-distilroberta-v1
                           (embedding.rs:29,159: position ids start at padding_idx + 1; such models also
                           have max_pos = 514, type_vocab = 1, ln_eps = 1e-5)                  */
    int32_t precision;  /* MX_PREC_BF16 (0, the default: bf16 operands, f32 accumulation -- the ingest path) |
                           MX_PREC_BF16X3: every GEMM operand carried as hi + lo bf16 (three MFMA products per
                           f32 product), f32 hidden state and f32 attention -- scores BETWEEN embeddings within
                           1e-3 of the f64 evaluation of the f32 CPU path (embedding.rs:109) on every draw of
                           checkpoint-like weights tried (2e-5 typical, 6e-4 at worst), where the bf16 path moves
                           them by 1e-2 ... 4e-2; about 3.2x slower.  What the loaders pick
                           for CLS-pooled hidden-768 models |
                           MX_PREC_MIXED (round 6): the attention block (Q, K, V, scores, PV, out-projection) as in
                           MX_PREC_BF16X3, the MLP's two GEMMs as TWO fp16 products per product (fp16 weights x fp16 hi + lo
                           activations): 15-17 % faster than MX_PREC_BF16X3; scores within 5.3e-4 on nine of eleven draws of
                           checkpoint-like weights, 1.1e-3 on the other two, where MX_PREC_BF16X3 has 6e-4
                           (profiles/r6_precision_modes_over_seeds.txt) -- "about 1e-3 at worst", opt-in |
                           MX_PREC_MIXED1 (round 6): MX_PREC_MIXED with the MLP on ONE fp16 product per product (weights, LayerNorm
                           output and GELU output one fp16 value each; residual stream, GEMM results and LayerNorm stay f32) and P
                           as one bf16 value in P.V: 0.39 / 0.48 of the bf16 rate, scores 1.5e-5 ... 2e-2 by draw -- an order
                           of magnitude better than bf16, NOT a mode that holds 1e-3; opt-in                           */
} mx_encoder_cfg;
enum { MX_PREC_BF16 = 0, MX_PREC_BF16X3 = 1, MX_PREC_MIXED = 2, MX_PREC_MIXED1 = 3 };
/* sizeof(mx_encoder_cfg) of the library that is loaded: the struct grew a trailing field (`precision`) and may again; a shim
 * built against an older header compares this with its own sizeof at start-up instead of letting the library read past
 * its struct (same handshake as mx_index_stats_size). */
size_t mx_encoder_cfg_size(void);

/*
 * Weight blob: f32, little-endian, tensors concatenated in this order (HF BertModel names,
 * Linear weights [out, in]):
 *   embeddings.word_embeddings.weight [vocab,H]; embeddings.position_embeddings.weight [max_pos,H];
 *   embeddings.token_type_embeddings.weight [type_vocab,H]; embeddings.LayerNorm.{weight,bias} [H];
 *   then per layer l: attention.self.{query,key,value}.{weight [H,H], bias [H]} (q,k,v in turn);
 *   attention.output.dense.{weight [H,H], bias}; attention.output.LayerNorm.{weight,bias};
 *   intermediate.dense.{weight [F,H], bias [F]}; output.dense.{weight [H,F], bias [H]};
 *   output.LayerNorm.{weight,bias}.
 * mx_encoder_weight_bytes() gives the exact size.  memex_amd/weights.py packs it from a
 * state-dict / safetensors file.
 */
size_t mx_encoder_weight_bytes(const mx_encoder_cfg *cfg);

/* Replaces SentenceEmbeddingsBuilder::remote(..).create_model() (embedding.rs:99-100). */
int mx_encoder_create(const mx_encoder_cfg *cfg, const void *weights, size_t nbytes, int device,
                      mx_encoder **out);
void mx_encoder_destroy(mx_encoder *enc); /* drops one reference; frees HBM when the last one goes */
/*
 * Keyed, ref-counted form: the reference spawns an embedder -- loads the
 * checkpoint -- for every HTTP search and every ingest task (collections/handlers.rs:61-63,
 * worker/tasks.rs:17).  The first mx_encoder_open(key, ...) uploads the weights; later opens of the
 * same key return the SAME resident encoder in O(1) (weights may then be NULL).  Calls on a shared
 * handle are serialised inside.  key NULL / "" = mx_encoder_create.
 */
int mx_encoder_open(const char *key, const mx_encoder_cfg *cfg, const void *weights, size_t nbytes, int device,
                    mx_encoder **out);
/* Stream contract as for the index (mx_index_wait_stream): order the next *_device call after the
 * work enqueued so far on `hip_stream`. */
int mx_encoder_wait_stream(mx_encoder *enc, void *hip_stream);

/*
 * Replaces model.encode(&segments) (embedding.rs:109) after tokenisation: ids [B, S] row-major
 * ([CLS] .. [SEP] already added, padded with anything past lens[b]); lens[b] in [1, S];
 * out [B, hidden] f32 (pooled, L2-normalised when cfg.normalize).  bf16 MFMA arithmetic with f32
 * accumulation; agrees with the f32 CPU path to |delta cosine| <= 1e-3 (north_star tolerance).
 * A row's embedding does not depend on its neighbours in a call, but it does depend, in its last bits, on
 * how many packed rows share its pass (calls are cut into passes of <= 1024 sequences / 131072 rows): three
 * kernel sets round at the same points and sum in f32 in different orders -- hidden 384: passes of <= 2048
 * rows (query-time and small documents: encoder_small.hip) / larger ones; hidden 768: passes of <= 4096 rows (the two Add & LayerNorm
 * GEMMs split over k) / below / from 32768 rows (the latter round the pre-LayerNorm sum to bf16 once more).  The same text embedded alone and inside a bulk ingest call
 * agrees to 1 - cos ~ 1e-7 .. 1e-5 (tests/test_encoder_gpu.py::test_a_row_across_the_pass_size_regimes),
 * not bit for bit; calls of the same shape are bit-reproducible.  MX_PREC_BF16X3 uses one kernel set.
 * (The attention of a head-dim-32 model comes in three forms chosen per pass from its longest sequence and its size -- one
 * head per work item, two, or a kernel of its own for short sequences: the same arithmetic per head and the same bits, except
 * that in the two-head form a head whose exp2 row sums leave the fast path's range -- logits of several hundred -- takes
 * its pair partner through the running-maximum loop with it.)
 */
int mx_encoder_encode(mx_encoder *enc, const int32_t *ids, const int32_t *lens, int B, int S, float *out);
int mx_encoder_encode_device(mx_encoder *enc, const int32_t *d_ids, const int32_t *d_lens, int B, int S,
                             float *d_out);

typedef struct mx_encoder_stats {
    uint64_t calls;
    uint64_t sequences;
    uint64_t tokens;   /* non-padding tokens processed */
    double flops;      /* algorithmic flops: tokens * L * (8H^2 + 4HF) + attention 4*S_b*H per token */
    double gpu_ms;     /* HIP-event time (profiling on) */
} mx_encoder_stats;
int mx_encoder_set_profiling(mx_encoder *enc, int on);
int mx_encoder_get_stats(mx_encoder *enc, mx_encoder_stats *out);
int mx_encoder_reset_stats(mx_encoder *enc);

/* =====================================================================================
 * WordPiece / byte-level BPE tokenizer + sliding-window segmenter (host code; replaces the `tokenizers` crate
 * calls of segment_text, lib/libmemex/src/llm/embedding.rs:155-198, and the tokenisation rust-bert
 * performs inside model.encode, embedding.rs:109)
 * ===================================================================================== */
typedef struct mx_tokenizer mx_tokenizer;

/* vocab: BERT vocab.txt (one token per line; needs [PAD] [UNK] [CLS] [SEP]).  lowercase = 1 for
 * the uncased MiniLM / bge models.  Replaces Tokenizer::from_pretrained (embedding.rs:163). */
int mx_tokenizer_create(const char *vocab_path, int lowercase, mx_tokenizer **out);
int mx_tokenizer_create_from_memory(const char *vocab, size_t nbytes, int lowercase, mx_tokenizer **out);
/* Byte-level BPE: the tokenizer of all-distilroberta-v1, the third model segment_text accepts (embedding.rs:159): GPT-2
 * pre-tokenizer regex, bytes -> printable code points, BPE over vocab.json + merges.txt (needs <s> </s> <pad>), <s> .. </s>
 * as the special tokens, ByteLevel decoder.  Every call below works on either kind of handle. */
int mx_tokenizer_create_bpe(const char *vocab_json_path, const char *merges_path, mx_tokenizer **out);
int mx_tokenizer_create_bpe_from_memory(const char *vocab_json, size_t n_vocab, const char *merges, size_t n_merges,
                                        mx_tokenizer **out);
/* tokenizer.json -- the file Tokenizer::from_pretrained itself reads (embedding.rs:163): model.vocab (+ model.merges) and the
 * normalizer's lowercase flag select one of the two kinds above.  The stacks of the reference's three models are accepted
 * (BertNormalizer / BertPreTokenizer / WordPiece "##" / WordPiece decoder; ByteLevel / BPE / ByteLevel decoder); any other
 * component -> MX_EUNSUPPORTED.  The file's truncation / padding blocks are ignored (segment_text overrides the first,
 * embedding.rs:172-176, and decodes with skip_special_tokens, :182,189). */
int mx_tokenizer_create_from_json(const char *tokenizer_json_path, mx_tokenizer **out);
int mx_tokenizer_create_from_json_memory(const char *json, size_t nbytes, mx_tokenizer **out);
void mx_tokenizer_destroy(mx_tokenizer *tok);
int mx_tokenizer_vocab_size(mx_tokenizer *tok, int *n);

/* tokenizer.encode(text, add_special_tokens) (embedding.rs:181).  *n = number of ids produced
 * (may exceed cap: only the first cap are written). */
int mx_tokenizer_encode(mx_tokenizer *tok, const char *text, int add_special_tokens, int32_t *ids, int cap, int *n);
/* tokenizer.decode(ids, skip_special_tokens) (embedding.rs:182,189); *nbytes includes the NUL. */
int mx_tokenizer_decode(mx_tokenizer *tok, const int32_t *ids, int n, int skip_special_tokens, char *out, size_t cap,
                        size_t *nbytes);
/* segment_text (embedding.rs:155-198): windows of max_length tokens starting every
 * max_length - stride tokens, each decoded back to text (first window also gets " ' " -> "'").
 * out receives the segments as consecutive NUL-terminated strings; *nbytes = total bytes. */
int mx_tokenizer_segment(mx_tokenizer *tok, const char *text, int max_length, int stride, char *out, size_t cap,
                         size_t *nbytes, int *n_segments);
/* segment_text for a batch of documents (what the ingest worker's queue holds: tasks.rs:17-19, up to five tasks at a time,
 * worker/lib.rs:36): the documents are dealt to host threads.  out: the windows of text 0, then of text 1, ..., each
 * NUL-terminated; n_segments[i] = windows of text i.  *nbytes is always the size needed; out is filled only when cap covers
 * it (call again with a larger buffer otherwise). */
int mx_tokenizer_segment_batch(mx_tokenizer *tok, const char *const *texts, int n_texts, int max_length, int stride,
                               char *out, size_t cap, size_t *nbytes, int32_t *n_segments);
/* Test hook: the WordPiece encoder stage by stage (normalize -> pre-tokenize -> WordPiece, as embedding.rs:181's encode is
 * usually described) -- mx_tokenizer_encode runs the same steps in one pass; tests hold the two against each other. */
int mx_tokenizer_encode_staged(mx_tokenizer *tok, const char *text, int32_t *ids, int cap, int *n);
/* Batch for mx_encoder_encode: [CLS] .. [SEP], truncated to max_seq_length, padded with [PAD] to
 * row pitch s_cap; lens[b] = tokens incl. specials; *S = longest row (<= s_cap or MX_EINVAL). */
int mx_tokenizer_encode_batch(mx_tokenizer *tok, const char *const *texts, int B, int max_seq_length, int32_t *ids,
                              int s_cap, int32_t *lens, int *S);

#ifdef __cplusplus
}
#endif
#endif /* MEMEX_HIP_H */
