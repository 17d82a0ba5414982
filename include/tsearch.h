/*
 * tsearch.h - C ABI of libtsearch.so, the MI355X (gfx950) brute-force theorem-search engine.
 *
 * The reference (uw-math-ai/TheoremSearch) has no FFI of its own: its hot path is Python
 * calling third-party Python.  These entry points are what a ctypes
 * binding for that path binds; each one names the reference call it stands in for.  The only
 * caller in this repository is theoremsearch_amd/_ffi.py; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every function returns an int status (TS_OK or a negative TS_ERR_*);
 *     ts_last_error() gives the thread-local message of the last failure.
 *   - the caller owns every host buffer it passes; the library owns the device memory behind
 *     the opaque handles.
 *   - rows are row-major; dtype codes TS_F32 (IEEE binary32) and TS_BF16 (bfloat16 bit patterns,
 *     uint16_t).
 *   - search results: for every query, k entries ordered by score descending, then index
 *     ascending; NaN scores are never returned; when fewer than k rows qualify the tail is
 *     (score = -inf, index = -1).
 *   - a stream argument is a hipStream_t passed as void* (NULL = the index's own stream).  With
 *     host output buffers the call returns after the results have landed; with device buffers
 *     it returns after enqueueing the work on that stream.  The index's own stream is a blocking
 *     stream: it is ordered with the legacy null stream, so NULL is also the right value for work
 *     that lives on the default stream of the caller (torch.cuda.current_stream().cuda_stream is 0
 *     there).  A call that arrives on a different stream than the previous call on the same handle
 *     is ordered behind that call by the library (the handle's scratch buffers are shared); for
 *     that the library records an event at the END of every call on the stream of that call and never touches
 *     that stream again: a caller-owned stream may be destroyed as soon as the caller's own work on it is done.
 *     A call made while its stream is being captured into a HIP graph records no such event: the caller orders replays of
 *     that graph against other calls on the handle itself.
 *   - device queries may be read in place (no copy is made when they already have the form the kernels multiply):
 *     they must stay unchanged until the work the call enqueued has run.
 *   - one handle may be used from several threads (the Streamlit apps share one model and one
 *     library across session threads, streamlit_app.py:52); calls on one handle are serialised
 *     inside, different handles are independent.
 */
#ifndef TSEARCH_H
#define TSEARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library itself is built with hidden visibility: what this header declares is what it exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define TS_VERSION 100 /* 0.1.0 */

#define TS_OK 0
#define TS_ERR_INVALID (-1)     /* bad argument */
#define TS_ERR_HIP (-2)         /* a HIP runtime call failed (message has hipGetErrorString) */
#define TS_ERR_NOMEM (-3)       /* device or host allocation failed */
#define TS_ERR_NODEVICE (-4)    /* no usable gfx950 device */
#define TS_ERR_UNSUPPORTED (-5) /* valid request this build has no kernel for */
#define TS_ERR_INTERNAL (-6)

#define TS_F32 0
#define TS_BF16 1

#define TS_METRIC_IP 0  /* inner product: pgvector "<#>" on stored-normalised rows, streamlit_app.py:275-282 */
#define TS_METRIC_COS 1 /* cosine: sentence_transformers.util.cos_sim, app_showcase_model.py:93 */

#define TS_ALGO_AUTO 0
#define TS_ALGO_SCAN 1 /* streaming dot-product scan with per-wave running top-k (any dtype, any d) */
#define TS_ALGO_MFMA 2 /* MFMA contraction with thresholded candidate selection (bf16 or fp32 index, d = 384, 512, 768 or 1024) */

#define TS_MAX_K 256

typedef struct ts_index ts_index;
typedef struct ts_timer ts_timer;

/* Per-call counters of ts_search_ex (all optional; zero-filled first). */
typedef struct ts_search_stats {
    int32_t algo;            /* TS_ALGO_SCAN or TS_ALGO_MFMA actually used */
    int32_t levels;          /* MFMA path: number of threshold levels run */
    int32_t fallback_queries; /* MFMA path: queries re-run through the scan (candidate overflow); -1 = not read back */
    int32_t reserved;
    int64_t candidates;      /* MFMA path: candidates appended in the last level; -1 = not read back */
} ts_search_stats;

/* ---- library / device -------------------------------------------------------------------- */

int ts_version(void);
const char *ts_last_error(void);
/* Number of visible HIP devices (0 on a CPU-only host; never fails for "no device"). */
int ts_device_count(int *count);
int ts_device_info(int device, char *name, int name_len, int64_t *total_mem_bytes, int32_t *compute_units);
int ts_device_synchronize(int device);

/* ---- the index: the [N x d] theorem-embedding matrix resident in HBM -------------------------
 * Stands in for the corpus tensor of the in-process form (torch.load of corpus_embeddings.pt,
 * app_showcase_model.py:52) and for the theorem_embedding_* tables of the in-database form
 * (rds_schema.sql:43-56).  dtype is the storage type; metric TS_METRIC_COS L2-normalises every
 * row once at upload (x / max(||x||, 1e-12), what util.cos_sim redoes on every call), then
 * rounds to bf16 when dtype is TS_BF16.  row_offset is the global id of row 0 (sharded use).
 */
int ts_index_create(int device, int64_t n, int32_t d, int dtype, int metric, ts_index **out);
/* Waits for the handle's work in flight, then frees it.  TS_ERR_UNSUPPORTED while views of it are alive (ts_index_view). */
int ts_index_destroy(ts_index *ix);
int ts_index_set_row_offset(ts_index *ix, int64_t row_offset);
/* The index's own HIP stream (hipStream_t as void*): what stream = NULL means in the calls below. */
int ts_index_stream(const ts_index *ix, void **stream);
/* Makes `stream` (hipStream_t as void*, not NULL) wait for the end of the LAST call on this handle, whatever stream that
 * call ran on (stream-ordered, no host wait): a side stream that consumes a search's device results - the exchange + merge
 * of the sharded search - needs no event of its own on the search's stream. */
int ts_index_wait_order(ts_index *ix, void *stream);
/* The ONE host-only entry (SURVEY.md 8b "ts_search_cpu"; BASELINE.json configs[0], the reference's CPU-runnable case:
 * util.cos_sim + argsort over ~1k theorems, compare_embeddings.py:24-31,55-92).  Stateless and explicit: no device entry
 * falls back to it, nothing selects it by itself; Python reaches it only through TheoremIndex(..., device=-1).
 * rows [n x d] and queries [nq x d] are HOST arrays (fp32, or bf16 bit patterns), prepared on every call exactly as
 * ts_index_upload / ts_search prepare them on the device (metric cos: x / max(||x||, 1e-12) with the norm in fp64; store_dtype
 * TS_BF16: rounded to nearest even); scores are fp32 dot products of the prepared values; results in the library's order
 * (score descending, row ascending; NaN never ranks), padded with (-inf, -1).  threads <= 0: one per hardware thread. */
int ts_search_cpu(const void *rows, int rows_dtype, int64_t n, int32_t d, int store_dtype, int metric,
                  const void *queries, int q_dtype, int32_t nq, int32_t k, float *out_scores, int64_t *out_idx,
                  int32_t threads);
/* Device-to-device copy of `bytes` bytes on `device`, enqueued on `stream` (hipStream_t as void*; NULL = the legacy null
 * stream): lets a Python caller that only holds raw device addresses stage a query batch in a buffer of its own before a
 * search on another stream reads it (ShardedSearcher with several searches in flight: the caller's stream then waits for
 * this copy, not for the whole search). */
int ts_copy_device(int device, void *dst, const void *src, int64_t bytes, void *stream);
/* Waits until everything this handle has enqueued (on its own stream and on the stream of its last call) is done. */
int ts_index_synchronize(ts_index *ix);
int ts_index_info(const ts_index *ix, int64_t *n, int32_t *d, int32_t *dtype, int32_t *metric,
                  int64_t *ld_elems, int64_t *row_offset, void **device_rows);

/* Tuning / diagnostic options of one handle (the TS_* knobs of DESIGN.md section 8, e.g. "TS_MFMA_FIRST_ROWS").  Their
 * initial values are read from the environment once, when the handle is created; afterwards only these calls change
 * them (the search path never calls getenv).  reset = back to the built-in default.  The reference has no counterpart. */
int ts_index_set_option(ts_index *ix, const char *name, int32_t value);
int ts_index_reset_option(ts_index *ix, const char *name);

/* Rows [row0, row0 + nrows) from host memory (src_dtype TS_F32 or TS_BF16, dense [nrows x d]).
 * Replaces building the corpus tensor (app_create_embeddings.py:81-89) and the per-row INSERT /
 * upsert of vectors (parsed_papers_to_vector_rds/rds.py:37-91, ec2/generate_embeddings/__main__.py:85-99). */
int ts_index_upload(ts_index *ix, const void *host_rows, int src_dtype, int64_t row0, int64_t nrows);
/* The same from device memory (row stride src_ld elements), enqueued on `stream`: encoder output
 * goes straight into the index without a host hop. */
int ts_index_upload_device(ts_index *ix, const void *dev_rows, int src_dtype, int64_t src_ld,
                           int64_t row0, int64_t nrows, void *stream);
/* Growth ("incremental append mirroring the upsert-by-slogan_id semantics",
 * ec2/generate_embeddings/__main__.py:85-99 - a slogan_id that is not in the table yet is INSERTed).  ts_index_reserve
 * makes room for `capacity_rows` rows without changing n; ts_index_append* add nrows rows behind the last one (growing
 * the allocation 1.5x when it is full: one device-to-device move of the rows) and return the global id of the first
 * new row.  Not on views / subset indexes, and refused (TS_ERR_UNSUPPORTED) while views of the index are alive.
 * Views and subsets made earlier do not see the new rows. */
int ts_index_reserve(ts_index *ix, int64_t capacity_rows);
int ts_index_append(ts_index *ix, const void *host_rows, int src_dtype, int64_t nrows, int64_t *first_row);
int ts_index_append_device(ts_index *ix, const void *dev_rows, int src_dtype, int64_t src_ld, int64_t nrows,
                           void *stream, int64_t *first_row);
/* Zero-copy: the index adopts rows that already sit in device memory - e.g. the tensor an encoder
 * wrote.  They must be what the kernels multiply: the index's storage dtype, row stride = ld (d padded to 64 elements),
 * normalised already when the metric is cosine; the allocation must hold capacity_rows >= n rounded up to 256 rows (whole
 * 32-row tiles are read; rows past n are never returned).  The caller keeps ownership: the memory must stay alive and
 * unchanged while searches run; the index's own allocation is released; an attached index cannot grow. */
int ts_index_attach_device(ts_index *ix, void *dev_rows, int64_t capacity_rows);
/* Stored rows back to the host in the storage dtype, dense [nrows x d] (what the kernels multiply). */
int ts_index_download(ts_index *ix, void *host_rows, int64_t row0, int64_t nrows);

/* ---- search --------------------------------------------------------------------------------
 * out_scores [nq x k] float, out_idx [nq x k] int64 (global ids = local row + row_offset).
 * Replaces  util.cos_sim(q, db)[0] + np.argsort(-s)[:5]        (app_scratchpad.py:129-130)
 *           util.cos_sim(q, db)[0] + torch.topk(s, k, sorted)   (app_showcase_model.py:93-96)
 *           ORDER BY e.embedding <#> q ASC LIMIT k              (streamlit_app.py:282-283)
 * queries: [nq x d] dense, q_dtype TS_F32 or TS_BF16, in host (q_on_device = 0) or device memory.
 * With metric COS the queries are L2-normalised first; with a bf16 index they are rounded to bf16.
 * 1 <= k <= TS_MAX_K.
 */
int ts_search(ts_index *ix, const void *queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
              float *out_scores, int64_t *out_idx, int out_on_device, void *stream);
int ts_search_ex(ts_index *ix, const void *queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                 float *out_scores, int64_t *out_idx, int out_on_device, void *stream, int algo,
                 ts_search_stats *stats);

/* Parses pgvector's text form of vectors - "[0.1,-2e-3,...]", one per row, whatever stands between the rows (ids, tabs,
 * newlines: the output of COPY (SELECT slogan_id, embedding ...) TO STDOUT or of SELECT embedding::text) - into a dense
 * fp32 matrix, each value by one strtof as pgvector's own vector_in does.  Feeds ts_index_upload from the RDS tables
 * the upsert pipeline fills (ec2/generate_embeddings/embeddings.py:27-40 sends embedding.tolist(); rds_schema.sql
 * embedding_* tables) without a Python float per value.  Stops after max_rows rows or at the last complete row of the
 * buffer; *consumed = bytes up to the end of that row, so a stream can be fed in pieces.  Host only, no device. */
int ts_parse_pgvector_text(const char *text, int64_t len, int32_t d, float *out, int64_t max_rows, int64_t *rows_parsed,
                           int64_t *consumed);

/* A second HANDLE on the rows of `src` (no copy): its own stream, scratch buffers and lock, so that two searches of the
 * same corpus can be in flight on two streams - the small kernels at the head of one search (query preparation, the
 * threshold sample and its select) then overlap the tail of the previous one (final select, re-run check, merge).
 * What a serving loop over independent query batches wants; the reference has no counterpart (one process, one query
 * at a time, streamlit_app.py:165-173).  Read-only; must be destroyed before `src`; set_row_offset / subsets of `src`
 * made later are not seen by the view. */
int ts_index_view(ts_index *src, ts_index **out);

/* A second index holding copies of the given rows of `src` (global ids, strictly ascending, host memory); searches
 * of it return the ORIGINAL global ids, in the same canonical order.  The filtered search for query batches and for
 * filters that stay fixed over many searches (a sidebar state, app_showcase_model.py:96-129; the WHERE clause of
 * streamlit_app.py:175-283): the copy costs 2 * nrows * ld * elem bytes of HBM traffic once, after which every
 * search runs the unfiltered kernels over nrows rows.  Rows are copied as stored (no second normalisation).
 * Uploading into a subset index and ts_rank_of on it are not supported. */
int ts_index_subset(ts_index *src, const int64_t *rows, int64_t nrows, ts_index **out);

/* Search restricted to the rows whose bit is set in row_mask (uint32 words, bit r & 31 of word r >> 5 = row r,
 * ceil(n / 32) words, host or device memory): the k best ALLOWED rows, exactly.  Stands in for the metadata
 * filters of the apps - the WHERE clause in front of ORDER BY ... LIMIT k (streamlit_app.py:175-243,253-283) and
 * the post-filter loop over the top-200 (app_showcase_model.py:96-129), which can return fewer than top_k hits
 * although more exist.  The caller evaluates the predicates to the bitmask; the scan kernel tests the bit before
 * a row can enter a top-k list (N / 8 extra bytes per pass). */
int ts_search_filtered(ts_index *ix, const void *queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                       const uint32_t *row_mask, int mask_on_device, float *out_scores, int64_t *out_idx,
                       int out_on_device, void *stream);

/* The same with an algorithm hint (TS_ALGO_SCAN is honoured; TS_ALGO_MFMA is refused with TS_ERR_UNSUPPORTED when the mask is
 * too sparse or lives on the device; TS_ALGO_AUTO decides by the mask's density) and the per-call counters. */
int ts_search_filtered_ex(ts_index *ix, const void *queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                          const uint32_t *row_mask, int mask_on_device, float *out_scores, int64_t *out_idx,
                          int out_on_device, void *stream, int algo, ts_search_stats *stats);

/* Top-k of  score(row) + weight * bias[row]  over ALL rows of the index (or all rows row_mask allows; row_mask may be NULL):
 * the citation-weighted ranking of streamlit_app.py:348-364 -
 *     weighted_score = similarity + w * CASE WHEN citations > 0 THEN ln(citations) ELSE 0 END   ORDER BY weighted_score DESC
 * - computed over the whole corpus instead of over the max(50, 10 k) nearest rows the SQL ranks (streamlit_app.py:317): the
 * pool form misses a heavily cited theorem that is not among the 10 k nearest; this one cannot (SURVEY.md section 8f rank 4:
 * "score + w * ln(citations) as a fused per-row bias, one fp32 side array").  bias: float[n] in host or device memory, one
 * value per row of THIS index (the caller puts ln(citations) or 0 there); the term is added in fp32 where the key of a row
 * is made (fmaf(weight, bias[row], score)), so the order is weighted score descending, then row ascending.  out_scores
 * receives the weighted scores, out_sims (optional, same shape and place) the raw similarities.  Runs on the scan kernel
 * (four queries per pass at the HBM rate; n * 4 more bytes per pass); not on subset indexes. */
int ts_search_biased(ts_index *ix, const void *queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                     const float *bias, int bias_on_device, float weight, const uint32_t *row_mask, int mask_on_device,
                     float *out_scores, float *out_sims, int64_t *out_idx, int out_on_device, void *stream);

/* Rank of one given row per query in the canonical order of that query's scores over the whole index (0 = best):
 * the number of rows whose (score, -row) beats the target's.  One streaming pass that counts; replaces ranking the
 * full [nq x N] matrix and looking the relevant document up - np.argsort(-sim_matrix) followed by the position of
 * the exact document in mrr_at_k with k=None (compare_embeddings.py:96-123) - for corpora whose score matrix
 * does not fit.  target_rows are global ids (row_offset applies); out_rank[i] = -1 and out_score[i] = NaN when
 * the row is not in this index or its score is NaN.  Host pointers; out_score may be NULL. */
int ts_rank_of(ts_index *ix, const void *queries, int q_dtype, int q_on_device, int32_t nq, const int64_t *target_rows,
               int64_t *out_rank, float *out_score, void *stream);

/* The sharded form of ts_rank_of: how many rows of THIS index rank before a document with the given score and global id
 * (score descending, global id ascending), wherever that document lives.  The shard that holds the document gets its
 * score from ts_rank_of; the sum of ts_count_above over all shards is the document's rank in the whole corpus (the
 * document itself never counts).  out_counts[i] = -1 for a NaN score.  Host pointers. */
int ts_count_above(ts_index *ix, const void *queries, int q_dtype, int q_on_device, int32_t nq, const float *target_scores,
                   const int64_t *target_ids, int64_t *out_counts, void *stream);

/* Full [nq x n] fp32 score matrix (small N only): util.cos_sim(q_emb, s_emb) of
 * compare_embeddings.py:24,61.  out row stride is n. */
int ts_scores(ts_index *ix, const void *queries, int q_dtype, int q_on_device, int32_t nq,
              float *out, int out_on_device, void *stream);

/* Merge `nparts` partial top-k lists per query (e.g. the all-gathered per-shard results,
 * SURVEY.md section 8e) into the global top-k_out under the same ordering rule.
 * scores/idx: [nparts x nq x k_in]; entries with idx < 0 are padding. */
int ts_merge_topk(int device, const float *scores, const int64_t *idx, int32_t nparts, int32_t nq,
                  int32_t k_in, int32_t k_out, float *out_scores, int64_t *out_idx, int on_device,
                  void *stream);

/* The same for results that were exchanged as one packed block per part (one all-gather instead of
 * two): part p starts at packed + p * part_stride_bytes and holds its scores [nq x k_in] float at
 * offset 0 and its ids [nq x k_in] int64 at idx_offset_bytes (a multiple of 8).  Device memory only;
 * enqueued on `stream`. */
int ts_merge_topk_packed(int device, const void *packed, int64_t part_stride_bytes, int64_t idx_offset_bytes,
                         int32_t nparts, int32_t nq, int32_t k_in, int32_t k_out, float *out_scores,
                         int64_t *out_idx, void *stream);

/* ---- row-sharded search over RCCL (BASELINE.json configs[3]) -------------------------------
 * The corpus row-shards over the GPUs of one node: GPU g holds rows [g N / G, (g+1) N / G) in its own ts_index whose
 * row_offset makes the ids global; every GPU searches its shard for the (replicated) query batch; ONE ncclAllGather
 * over xGMI exchanges the packed per-shard top-k (12 * nq * k bytes per rank) and a merge kernel reduces the G * k
 * candidates per query.  The exchange lives in this library (RCCL bound by dlopen at first use; TS_RCCL_LIB names the
 * file when the process has not loaded one yet): no torch.distributed, no MPI on the search path.
 *
 * ts_comm_*: one member per PROCESS (one process per GPU, the torch.distributed.run model).  Rank 0 calls
 * ts_comm_unique_id and hands the 128 bytes to the other ranks by whatever channel the launcher has (a file, a TCP
 * store); every rank then calls ts_comm_create (collective).  ts_comm_search enqueues on `stream` the local search of
 * `shard`, the all-gather and the merge; every rank receives the global answer (device or host buffers as in
 * ts_search).  ts_comm_allgather is the bare collective on device buffers (recv holds world * bytes). */
typedef struct ts_comm ts_comm;
#define TS_COMM_ID_BYTES 128
int ts_comm_unique_id(void *id_out, int32_t id_bytes);
int ts_comm_create(int device, int32_t world, int32_t rank, const void *id, int32_t id_bytes, ts_comm **out);
int ts_comm_destroy(ts_comm *c);
int ts_comm_info(const ts_comm *c, int32_t *world, int32_t *rank, int32_t *device);
int ts_comm_allgather(ts_comm *c, const void *send_dev, void *recv_dev, int64_t bytes, void *stream);
int ts_comm_search(ts_comm *c, ts_index *shard, const void *queries, int q_dtype, int q_on_device, int32_t nq, int32_t k,
                   float *out_scores, int64_t *out_idx, int out_on_device, void *stream);

/* ts_shards_*: ONE process drives all the GPUs (ncclCommInitAll, one stream per device) - what a serving process such
 * as the Streamlit app would hold instead of one ts_index.  devices = NULL means 0 .. ngpu-1; a device id may repeat
 * (several shards on one GPU, for rehearsing the path on a one-GPU box: the exchange then uses device copies, RCCL
 * refuses two ranks on one GPU).  ts_shards_upload routes global rows to their shards; ts_shards_shard lends the
 * ts_index of shard g (rows [*lo, *hi)) for device uploads / options / profiling; ts_shards_search takes host queries
 * and returns host results. */
typedef struct ts_shards ts_shards;
int ts_shards_create(int32_t ngpu, const int32_t *devices, int64_t n_total, int32_t d, int dtype, int metric,
                     ts_shards **out);
int ts_shards_destroy(ts_shards *s);
int ts_shards_info(const ts_shards *s, int32_t *ngpu, int64_t *n_total, int32_t *uses_rccl);
int ts_shards_shard(ts_shards *s, int32_t g, ts_index **ix, int64_t *lo, int64_t *hi);
int ts_shards_upload(ts_shards *s, const void *host_rows, int src_dtype, int64_t row0, int64_t nrows);
int ts_shards_search(ts_shards *s, const void *queries, int q_dtype, int32_t nq, int32_t k, float *out_scores,
                     int64_t *out_idx);

/* ---- encoder epilogue ------------------------------------------------
 * Pooling + optional L2 normalisation + cast of a transformer's last hidden state, fused in one kernel that
 * writes straight into a buffer the search (or ts_index_upload_device) consumes - what sentence-transformers
 * does with mean pooling / last-token pooling + F.normalize + .to(dtype) after the forward
 * (SentenceTransformer.encode(..., normalize_embeddings=True): parsed_papers_to_vector_rds/embeddings.py:31-37,
 * ec2/generate_embeddings/embeddings.py:24-30, streamlit_app.py:173).
 * hidden: device [n x seq x d], h_dtype TS_F32 | TS_BF16, dense.  attention_mask: device int64 [n x seq]
 * (1 = token, 0 = padding).  pooling: TS_POOL_MEAN (sum of unmasked tokens / max(count, 1e-9)),
 * TS_POOL_LAST (last unmasked token), TS_POOL_CLS (token 0).  out: device [n x out_ld], out_dtype TS_F32 | TS_BF16. */
#define TS_POOL_MEAN 0
#define TS_POOL_LAST 1
#define TS_POOL_CLS 2
int ts_pool_normalize(int device, const void *hidden, int h_dtype, const int64_t *attention_mask, int64_t n,
                      int32_t seq, int32_t d, int pooling, int normalize, void *out, int out_dtype,
                      int64_t out_ld, void *stream);

/* Residual add + LayerNorm of the encoder, one kernel: out = LayerNorm(a + b) * gamma + beta over rows of d elements
 * (BertSelfOutput / BertOutput of the sentence-transformer the reference loads, compare_embeddings.py:11-12: 25 add +
 * 25 layer_norm launches per BERT-base forward in PyTorch).  a, b, out: device [rows x d] dense; gamma, beta: device [d];
 * all of `dtype` (TS_F32 | TS_BF16), 16-byte aligned; the sum, the mean and the variance are taken in fp32.
 * d a multiple of 8 (bf16) / 4 (fp32), at most 2048 / 1024.  `out` may alias a or b. */
int ts_add_layernorm(int device, const void *a, const void *b, const void *gamma, const void *beta, float eps, int64_t rows,
                     int32_t d, int dtype, void *out, void *stream);

/* The encoder's input layer, one kernel: out[i] = LayerNorm(word[ids[i]] + type[type_ids[i]] + pos[i % seq]) * gamma + beta
 * (BertEmbeddings of the sentence-transformer the reference loads, compare_embeddings.py:11-12: three gathers, two adds and a
 * layer_norm launch per forward in PyTorch).  ids, type_ids: device int64 [tokens] (type_ids may be NULL: row 0); word / pos /
 * type: device tables [n_word | n_pos | n_type][d]; gamma, beta: device [d]; out: device [tokens][d]; all of `dtype`
 * (TS_F32 | TS_BF16), 16-byte aligned; sums, mean and variance in fp32.  d as for ts_add_layernorm.  Ids outside a table are
 * clamped to it. */
int ts_embed_layernorm(int device, const int64_t *ids, const int64_t *type_ids, const void *word, const void *pos, const void *type,
                       int64_t n_word, int64_t n_pos, int64_t n_type, const void *gamma, const void *beta, float eps,
                       int64_t tokens, int32_t seq, int32_t d, int dtype, void *out, void *stream);

/* Self-attention of the encoder for short sequences, one kernel: out = softmax(Q K^T / sqrt(64) + key mask) V per (sequence,
 * head), bf16, straight from the fused query / key / value projection (BertSelfAttention of the sentence-transformer the
 * reference loads, compare_embeddings.py:11-12; a query is one sentence: app_showcase_model.py:92).  qkv: device bf16
 * [batch][seq][3][heads][64] (the output of one GEMM over the concatenated projection weights); attention_mask: device int64
 * [batch][seq], 0 = padding key, or NULL; out: device bf16 [batch][seq][heads * 64].  head_dim must be 64 and seq at most 128
 * (TS_ERR_UNSUPPORTED otherwise: the caller keeps its library attention); up to 64 tokens every score tile of a (sequence, head)
 * is held at once, 65 .. 128 tokens walk the query tiles against K / V fragments held in registers.  Scores and softmax in fp32, probabilities rounded to
 * bf16 for the second product (as flash attention does). */
int ts_attention_short(int device, const void *qkv, const int64_t *attention_mask, int32_t batch, int32_t seq, int32_t heads,
                      int32_t head_dim, void *out, void *stream);

/* fp32 attention of the encoder at the reference's fp32 storage (SentenceTransformer(name) without a dtype, streamlit_app.py:55,173),
 * head size 64 / 128 / 256 up to 512 / 256 / 128 tokens: softmax(Q K^T * scale + key mask [+ causal]) V on the exact-fp32 matrix
 * instructions, straight from the stacked projection's output qkv [batch][seq][(q_heads + 2 kv_heads) * head_dim] (query heads, key
 * heads, value heads) to out [batch][seq][q_heads * head_dim]; grouped-query when kv_heads < q_heads.  attention_mask: int64
 * [batch][seq] key mask or NULL.  pieces (may be NULL): also the bf16 pieces [batch * seq][3 * q_heads * head_dim] of the output
 * (ts_split_pieces pattern 0) for the fp32-class GEMM behind it; out may then be NULL (only the pieces are written).  Replaces torch's scaled_dot_product_attention and the four
 * layout copies around it in the three fused forwards. */
int ts_attention_float(int device, const void *qkv, const void *qkv_bias, const int64_t *attention_mask, int32_t batch, int32_t seq,
                       int32_t q_heads, int32_t kv_heads, int32_t head_dim, int causal, float scale, void *out, void *pieces, void *stream);
/* The same for the decoder-style encoder the production app embeds with (Qwen/Qwen3-Embedding-0.6B, streamlit_app.py:55;
 * Qwen3Attention: grouped-query, causal, head size 128): out = softmax(Q K^T / sqrt(128) + causal + key mask) V per (sequence,
 * query head); query head h reads key / value head h / (q_heads / kv_heads).  qkv: device bf16 [batch * seq][(q_heads + 2 kv_heads)
 * * 128] - query heads, key heads, value heads of each token, the output of one GEMM over the stacked projection weights after
 * ts_qk_norm_rope; attention_mask: device int64 [batch][seq], 0 = padding key, or NULL; out: device bf16 [batch * seq][q_heads * 128].
 * head_dim must be 128 and seq at most 128 (TS_ERR_UNSUPPORTED otherwise: the caller keeps its library attention); up to 64
 * tokens every score tile of a (sequence, head) is held at once, 65 .. 128 tokens walk the query tiles against K fragments held in
 * registers and a V^T image in LDS.  A query row
 * without a single allowed key (a padding token on the left of a causal sequence) comes back as zeros. */
int ts_attention_gqa(int device, const void *qkv, const int64_t *attention_mask, int32_t batch, int32_t seq, int32_t q_heads,
                    int32_t kv_heads, int32_t head_dim, int causal, void *out, void *stream);

/* The decoder-style encoder the production app embeds with (Qwen/Qwen3-Embedding-0.6B, streamlit_app.py:55;
 * ec2/generate_embeddings/embedders.py:1-4): RMSNorm, rotary positions, grouped-query attention, gated MLP.  Three kernels for
 * what its layers do around the GEMMs; each follows the roundings of the torch modules it replaces.
 *
 * ts_add_rmsnorm: s = a + b (rounded to `dtype`; b may be NULL: s = a), out_norm = gamma * round(s * rsqrt(mean(s^2) + eps))
 * over rows of d elements (Qwen3DecoderLayer's residual add followed by the next Qwen3RMSNorm); out_sum (may be NULL) receives
 * s, the new residual.  d as for ts_add_layernorm; buffers 16-byte aligned; out_sum may alias a or b. */
int ts_add_rmsnorm(int device, const void *a, const void *b, const void *gamma, float eps, int64_t rows, int32_t d, int dtype,
                   void *out_sum, void *out_norm, void *stream);
/* Per-head RMSNorm of queries and keys + rotary position embedding, in place on the fused projection's output
 * (Qwen3Attention: q_norm / k_norm over the head, apply_rotary_pos_emb).  qkv: device [tokens][(q_heads + 2 kv_heads) * 128]
 * (query heads, key heads, value heads; values untouched); q_weight, k_weight: device [128]; cos_table, sin_table: device
 * [seq][128] of `dtype` (fp32 angles cast to the model's type, as Qwen3RotaryEmbedding does); token t has position t % seq.
 * head_dim must be 128 (TS_ERR_UNSUPPORTED otherwise). */
int ts_qk_norm_rope(int device, void *qkv, const void *q_weight, const void *k_weight, const void *cos_table, const void *sin_table,
                    float eps, int64_t tokens, int32_t seq, int32_t q_heads, int32_t kv_heads, int32_t head_dim, int dtype,
                    void *stream);
/* fp32 values as bf16 pieces for an fp32-class GEMM on the bf16 matrix pipe (encoder forward at the reference's fp32 storage,
 * streamlit_app.py:55,173): hi = bf16(x), lo = bf16(x - hi); out [rows x 3k] bf16 = [hi | lo | hi] (pattern 0, activations) or
 * [hi | hi | lo] (pattern 1, weights), so that ONE bf16 GEMM with fp32 accumulation over depth 3k yields
 * x_hi w_hi + x_lo w_hi + x_hi w_lo - the fp32 product up to ~2^-17 |x||w| per term.  x fp32 [rows x k], k a multiple of 4. */
int ts_split_pieces(int device, const void *x, int64_t rows, int32_t k, int pattern, void *out, void *stream);
/* The producers of the encoder forward writing the pieces of their fp32 output themselves ([hi | lo | hi], 3 d bf16 per row), next
 * to the fp32 output the residual path needs: the GEMM that follows reads them without a ts_split_pieces pass in between.  Same
 * arithmetic as ts_add_layernorm / ts_add_rmsnorm / ts_gemma_norm with dtype TS_F32.  a_bias / bias / qkv_bias (each may be NULL): the
 * bias of the GEMM that produced the input, added on the way in - that GEMM then runs without one (torch.addmm with an output type
 * copies the broadcast bias into its result before the GEMM: one more pass over the output per GEMM). */
int ts_add_layernorm_pieces(int device, const void *a, const void *a_bias, const void *b, const void *gamma, const void *beta, float eps,
                            int64_t rows, int32_t d, void *out, void *pieces, void *stream);
int ts_add_rmsnorm_pieces(int device, const void *a, const void *b, const void *gamma, float eps, int64_t rows, int32_t d,
                          void *out_sum, void *out_norm, void *pieces, void *stream);
int ts_gemma_norm_pieces(int device, const void *y, const void *x, const void *w_post, const void *w_next, float eps, int64_t rows,
                         int32_t d, void *out_sum, void *out_norm, void *pieces, void *stream);
/* fp32 activation straight into pieces [rows x 3n]: kind 0 = gelu (erf form) of x [rows x n] (BertIntermediate behind its GEMM + bias);
 * kind 1 = silu(gate) * up, kind 2 = gelu_tanh(gate) * up of x [rows x 2n] (gate columns, then up columns). */
int ts_act_pieces(int device, const void *x, const void *bias, int64_t rows, int32_t n, int kind, void *pieces, void *stream);
/* Gated-MLP activation (Qwen3MLP: SiLU(gate_proj(x)) * up_proj(x)) on the output of ONE GEMM over the stacked gate / up
 * weights: gate_up device [rows][2 * inter] (gate columns, then up columns) -> out device [rows][inter]. */
int ts_swiglu(int device, const void *gate_up, int64_t rows, int32_t inter, int dtype, void *out, void *stream);

/* The reference's second embedder, google/embeddinggemma-300m (ec2/generate_embeddings/embedders.py:1-4; Gemma3TextModel:
 * "sandwich" RMSNorms around every sublayer, q / k norms over heads of 256, rotary positions, GeGLU).  Three kernels for what its
 * layers do around the GEMMs, with the roundings of the torch modules they replace (Gemma3RMSNorm: v * rsqrt(mean(v^2) + eps) *
 * (1 + w) in fp32, rounded once).
 *
 * ts_gemma_norm: s = x + norm(y; w_post) (the residual add in `dtype`; y may be NULL: s = x), out_norm = norm(s; w_next) over rows
 * of d elements - the post-sublayer norm, the residual add and the pre-norm of what follows in one kernel; out_sum (may be NULL)
 * receives s.  d as for ts_add_layernorm; buffers 16-byte aligned. */
int ts_gemma_norm(int device, const void *y, const void *x, const void *w_post, const void *w_next, float eps, int64_t rows,
                  int32_t d, int dtype, void *out_sum, void *out_norm, void *stream);
/* ts_qk_norm_rope for Gemma3Attention: heads of 256, Gemma3RMSNorm (weights q_weight / k_weight [256]), rotate_half at 128;
 * cos / sin: [seq][256] of `dtype` (the layer type's table: sliding and full attention layers use different bases). */
int ts_gemma_qk_norm_rope(int device, void *qkv, const void *q_weight, const void *k_weight, const void *cos_table,
                          const void *sin_table, float eps, int64_t tokens, int32_t seq, int32_t q_heads, int32_t kv_heads,
                          int32_t head_dim, int dtype, void *stream);
/* Gemma3MLP's activation on the fused projection's output: out = gelu_tanh(gate) * up (gate_up [rows][2 * inter]: gate columns,
 * then up columns), the activation rounded to `dtype` before the product as torch does. */
int ts_geglu(int device, const void *gate_up, int64_t rows, int32_t inter, int dtype, void *out, void *stream);

/* ---- kernel timing inside the library ----------------------------------------------------------
 * With profiling enabled, every launch of the dominant kernel of a search (the full-corpus pass of
 * the MFMA path, or the scan kernel) is bracketed by a hipEvent pair on the stream it runs on.
 * ts_index_profile_read waits for the recorded events and returns the number of bracketed launches,
 * their summed duration and the corpus rows one launch covers; it then clears the record. */
int ts_index_profile_enable(ts_index *ix, int enable);
int ts_index_profile_read(ts_index *ix, int64_t *launches, double *total_ms, int64_t *rows_per_launch);

/* Clock probe of the MFMA full pass (diagnostic build of the kernel, selected with option TS_MFMA_VARIANT = 3; timing of
 * that build is not representative): the kernel stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its
 * tile loop; this returns the median over workgroups of the last probed launch: in-kernel clock in GHz, shader cycles
 * per unit (32 rows x 384 k), units per workgroup.  Zeros when no probe has run.
 * (MI355X_MICROARCH.md "DVFS give-back" item 6.) */
int ts_index_probe_read(ts_index *ix, double *ghz, double *cycles_per_unit, double *units_per_workgroup);

/* ---- timing on a given stream (hipEvent pairs; bench.py measures kernels with these) ------- */
int ts_timer_create(int device, ts_timer **out);
int ts_timer_start(ts_timer *t, void *stream);
int ts_timer_stop(ts_timer *t, void *stream);
int ts_timer_elapsed_ms(ts_timer *t, float *ms); /* synchronises on the stop event */
int ts_timer_destroy(ts_timer *t);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* TSEARCH_H */
