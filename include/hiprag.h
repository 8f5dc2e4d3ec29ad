/*
 * hiprag.h -- C-ABI of libhiprag.so, the MI355X (gfx950) hybrid-retrieval hot path.
 *
 * This is the drop-in boundary for batd-htplus/intool-rag's retrieval path.  The reference is pure
 * Python and reaches its native code through third-party wheels; each entry point below names the
 * reference call site (file:line under the reference repo) whose native work it replaces.  The Python
 * host side (intool-rag_amd/hiprag/_native.py) binds exactly these symbols with ctypes; INTEGRATION.md
 * shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns int32 status: 0 = OK, <0 = HIPRAG_E_*; hiprag_last_error() gives a
 *     thread-local message for the last failure on the calling thread.
 *   - handles are opaque uint64; the library owns all device memory behind a handle until *_destroy.
 *   - "_dev" variants take DEVICE pointers and a hipStream_t (as void*); they enqueue work and return
 *     without synchronising.  Non-"_dev" variants take HOST pointers, copy, run and synchronise.
 *   - callers may call from any thread (asyncio.to_thread workers, rag/providers/hf/embeddings.py:53,76):
 *     each handle serialises its own calls with an internal mutex and sets its device on entry.
 *   - no Python callbacks, no exceptions across the boundary, no torch types.
 */
#ifndef HIPRAG_H
#define HIPRAG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPRAG_OK 0
#define HIPRAG_E_INVALID (-1)  /* bad argument */
#define HIPRAG_E_HIP (-2)      /* HIP runtime error (message has hipGetErrorString) */
#define HIPRAG_E_HANDLE (-3)   /* unknown / destroyed handle */
#define HIPRAG_E_IO (-4)       /* file error */
#define HIPRAG_E_NOMEM (-5)
#define HIPRAG_E_UNSUPPORTED (-6)

#define HIPRAG_METRIC_IP 0 /* larger score first  (faiss.IndexFlatIP order)  */
#define HIPRAG_METRIC_L2 1 /* smaller squared-L2 first (faiss.IndexFlatL2 order; rag/storage/faiss_index.py:123) */

/* ---- library ---------------------------------------------------------------------------------------- */
int32_t hiprag_version(void);
const char* hiprag_last_error(void);
int32_t hiprag_device_count(int32_t* out_count);
int32_t hiprag_device_sync(int32_t device);
/* The device's SCAN STREAM: one high-priority hipStream_t per device, owned by the library (valid until hiprag_shutdown).
 * The hybrid calls run their dense leg on it; hosts that chain scans themselves (hipidx_search_begin_dev,
 * hiphybrid_shard_begin_dev) pass it as scan_stream so that the kernels meant to run BESIDE a scan -- the finish of the
 * previous launch, a BM25 leg, an all-gather -- do not take the scan's CUs first: a scan workgroup needs an empty CU, the
 * others are many small workgroups, and the dispatcher serves the high-priority queue first (hipidx_set_spare_cus). */
int32_t hiprag_scan_stream(int32_t device, void** out_stream);
/* The device's two TAIL STREAMS (which = 0 / 1; normal priority, owned by the library): where the kernels that run beside a
 * scan belong -- the finish of the previous launch, an exchange, a merge.  One pair per device for the whole process: HIP
 * hands its four hardware queues per priority to streams in order of first use and lets later streams share them, and a
 * process that creates tail streams per index or per wrapper object slows its later pipelines down by that alone.  The
 * library's own pipeline (hipidx_search_dev) uses which = 0. */
int32_t hiprag_tail_stream(int32_t device, int32_t which, void** out_stream);
/* Optional process-level bracket: init checks that n_devices GPUs are visible (<= 0: at least one) and
 * creates their contexts up front; shutdown synchronises every device and drops every handle still registered (their
 * device memory goes with them) -- the reference has no counterpart, its indices live until the process exits
 * (`_INDEX_CACHE`, rag/storage/faiss_index.py:24). */
int32_t hiprag_init(int32_t n_devices);
int32_t hiprag_shutdown(void);

/* HIP-event timing on an arbitrary stream (bench.py measures kernels on the stream they run on). */
/* Calibration, not part of the path: GB/s this device sustains on a read-only stream with the dense scan's partition
 * (every wave its own contiguous range of 64 KiB blocks, non-temporal 16-byte loads) over a zeroed scratch buffer of
 * `bytes`, `reps` timed launches of four passes each.  bench.py reports it beside the scan's achieved rate. */
int32_t hiprag_probe_read_gbps(int32_t device, int64_t bytes, int32_t reps, double* out_gbps);

int32_t hiprag_event_create(uint64_t* out_event);
int32_t hiprag_event_record(uint64_t event, void* stream);
int32_t hiprag_event_elapsed_ms(uint64_t start, uint64_t stop, float* out_ms); /* synchronises on stop */
int32_t hiprag_event_destroy(uint64_t event);

/* ---- dense flat index (replaces faiss.IndexFlatL2 / IndexFlatIP) -------------------------------------
 * create   <- faiss.IndexFlatL2(d)                      rag/storage/faiss_index.py:123
 * add      <- index.add(float32[n,d])                   rag/storage/faiss_index.py:124
 * search   <- index.search(float32[nq,d], k)            rag/storage/faiss_index.py:83, rag/agent/search_engine.py:45
 * ntotal/d <- index.ntotal / index.d                    rag/storage/faiss_index.py:97,103
 * save/load<- faiss.write_index / faiss.read_index      rag/storage/faiss_index.py:133,54
 *
 * Results: exact top-k under (better score first, lower id first on ties), where the score of a row is the
 * fp64-accumulated inner product / squared L2 distance of the fp32 inputs; out_scores is that value rounded
 * to fp32.  Slots past ntotal are padded FAISS-style: id -1, score -FLT_MAX (IP) / FLT_MAX (L2).
 * Returned ids are row numbers in insertion order plus the index's id_base (row-sharded indices).
 * Inputs must be finite.
 */
/* One handle = one GPU's rows (`device`).  SURVEY 8b sketched a `n_shards` argument here; this library shards the way the
 * hardware is driven instead -- one process per GPU, each with its own handle over a contiguous row range
 * (`hipidx_set_id_base` makes its ids global), partial top-k merged after ONE all-gather by `hiprag_merge_topk_dev`
 * (hiprag/sharded.py).  The same holds for `hipbm25_create` (document-range shards, `hipbm25_set_id_base`). */
int32_t hipidx_create(int32_t d, int32_t metric, int32_t device, uint64_t* out_handle);
int32_t hipidx_destroy(uint64_t h);
int32_t hipidx_add(uint64_t h, const float* x_host, int64_t n);
/* `hipidx_add_dev` enqueues its re-tiling on `stream` and returns at once: `x_dev` must stay valid until that work has run.
 * The library orders everything that reads the rows behind it -- a later add that re-allocates, `hipidx_save`,
 * `hipidx_reconstruct` (host waits) and searches on any stream (device-side wait) -- so no caller-side synchronisation
 * is needed between an add and the next call on the same handle. */
int32_t hipidx_add_dev(uint64_t h, const float* x_dev, int64_t n, void* stream);
int32_t hipidx_ntotal(uint64_t h, int64_t* out_n);
int32_t hipidx_dim(uint64_t h, int32_t* out_d);
int32_t hipidx_metric(uint64_t h, int32_t* out_metric);
int32_t hipidx_set_id_base(uint64_t h, int64_t id_base);
int32_t hipidx_search(uint64_t h, const float* q_host, int32_t nq, int32_t k, float* out_scores, int64_t* out_ids);
/* device variant: out_scores64_dev [nq,k] double (exact, for cross-shard merges), out_scores_dev [nq,k]
 * float (may be NULL), out_ids_dev [nq,k] int64.  Enqueues and returns; q_dev and the outputs must stay valid until `stream`
 * has passed the call's work.  A batch of more queries than one launch takes (hipidx_launch_queries) is PIPELINED inside the
 * library: the scans run back to back on the device's scan stream (hiprag_scan_stream) with 48 CUs left out of their grids,
 * the finish of every launch but the last beside the next scan (hipidx_gate_tail_dev), `stream` ahead of the first scan
 * and behind the last finish -- 1M x 1024 rows, 16384 queries per call: 202 k queries/s; a 125 k-row shard: 1.32 M.  Such a
 * call uses every workspace slot: do not mix it with a begin / finish pipeline in flight on the same index. */
int32_t hipidx_search_dev(uint64_t h, const float* q_dev, int32_t nq, int32_t k, double* out_scores64_dev,
                          float* out_scores_dev, int64_t* out_ids_dev, void* stream);
/* Queries one scan pass serves: 64.  HIPRAG_SCAN_MODE picks the scan's operands:
 *   bf16 (default)  the scan streams a bf16 FILTER COPY of the rows (2 B per element, kept beside the fp32 rows: 6 B per
 *                   element of HBM in all) against bf16 query tiles
 *   q64             streams the fp32 rows themselves, split on the fly into bf16 hi + lo, against bf16 query tiles (N*d*4
 *                   bytes per pass: the accounting of SURVEY 8(d)); no extra memory is read
 * Both return the same exact results -- the scan only nominates candidate groups (a per-query candidate list, filtered
 * inside the scan against a bound it maintains itself); every candidate the finish needs is re-scored in fp64 from the fp32
 * rows and a certificate (or, where it fails, the exhaustive path) proves nothing was missed.  The modes differ in how many
 * bytes a pass streams and how wide the certificate's error bound is.  Any other value is rejected by hipidx_create. */
int32_t hipidx_pass_queries(uint64_t h, int32_t* out_n);
/* Queries one scan LAUNCH takes (a multiple of the pass size): the scan kernel runs launch/pass passes back to back
 * inside one launch -- each pass streams the index once for its own query tile -- so that no kernel boundary (45-60 us
 * of idle GPU) separates them.  Sized by the index so that a launch lasts about 2.6 ms: in the default mode 8 passes
 * (512 queries) at 1M x 1024, 16 (the cap, 1024 queries) from half that down; it changes when rows are added.
 * HIPRAG_LAUNCH_QUERIES fixes it. */
int32_t hipidx_launch_queries(uint64_t h, int32_t* out_n);
/* Two-phase form of one launch (nq <= hipidx_launch_queries) for callers that pipeline: begin = index scan into
 * workspace `slot` (0..7, allocated on first use); finish = two kernels out of that slot (candidate ranking + fp64 re-score
 * + top-k + certificate in one; the exhaustive path for flagged queries).  k <= 128 takes this path; deeper k (to 1000) is
 * answered by the exhaustive path alone.  begin(slot s) of a later call must be ordered after finish(slot s) of the call that
 * used it (stream order or an event); the two phases may run on different streams if finish waits for begin.
 * search_dev == begin + finish on slot 0, repeated for every hipidx_launch_queries queries. */
int32_t hipidx_search_begin_dev(uint64_t h, const float* q_dev, int32_t nq, int32_t k, int32_t slot, void* stream);
int32_t hipidx_search_finish_dev(uint64_t h, const float* q_dev, int32_t nq, int32_t k, int32_t slot,
                                 double* out_scores64_dev, float* out_scores_dev, int64_t* out_ids_dev, void* stream);
/* Leave n CUs out of the scan grid (default 0).  A scan workgroup takes a CU's whole LDS, so kernels of OTHER streams --
 * the finish of the previous launch, the BM25 leg of a hybrid call, at N > 1 the RCCL all-gather, whose ranks spin until
 * every peer has joined -- only run on CUs the scan leaves (or between scans).  The scan is HBM-bound and does not need
 * every CU: at 1M x 1024 rows a launch takes the same time on 192 workgroups as on 256 and 5 % longer on 160.  Takes
 * effect with the next launch, no synchronisation (the finish reads a launch's lists, never its partition).  The hybrid
 * calls and the row-sharded hosts set it themselves (hiphybrid_search*: 96 for the dense leg of the call). */
int32_t hipidx_set_spare_cus(uint64_t h, int32_t n);
int32_t hipidx_get_spare_cus(uint64_t h, int32_t* out_n);
/* The START GATE: make `stream` (a tail stream) wait until the scan launched AFTER the one of `slot` has STARTED, i.e. until
 * every workgroup of that next scan holds its CU (the scan counts its workgroups in and raises a word; a one-wave kernel on
 * `stream` polls it).  Enqueued on the tail stream between the wait for the slot's own scan and hipidx_search_finish_dev, it
 * makes the finish (and whatever follows: an all-gather, a merge) be dispatched onto the CUs the next scan has left
 * (hipidx_set_spare_cus), beside it, instead of racing it for CUs: which of two queues that become ready at the same moment
 * is served first is not something a host can steer (measured: 0.89-1.23 M queries/s on a 125 k-row shard for one and the
 * same arrangement of streams, depending only on which hardware queues the streams happened to get).  The next scan must
 * have been LAUNCHED already (error otherwise) -- so a pipelined host enqueues the tails of step i right after it has launched
 * the scan of step i + 1, and without the gate when it needs step i's result before another scan is due (hiprag/sharded.py;
 * hipidx_search_dev does the same for a batch of several launches).  The wait is BOUNDED (1 ms: the gate opens ~30 us after
 * the previous scan has ended): where kernels are serialised -- counter collection, a debugger -- the awaited scan cannot
 * start while the wait runs, and the gate then simply times out. */
int32_t hipidx_gate_tail_dev(uint64_t h, int32_t slot, void* stream);
/* make sure slot 0's search workspace for k exists so that search / search_dev never allocate */
int32_t hipidx_reserve_search(uint64_t h, int32_t k);
/* Capacity for n_rows rows now (one allocation of each of the index's buffers; rows added later beyond it grow it by half
 * again as usual).  A caller that knows the size of the collection -- create_faiss_index's one add of all embeddings,
 * rag/storage/faiss_index.py:121-124, or a chunked ingest -- avoids the allocate / copy / free cycles of growth. */
int32_t hipidx_reserve_rows(uint64_t h, int64_t n_rows);
int32_t hipidx_reconstruct(uint64_t h, int64_t row, float* out_host); /* row as stored (tests, export) */
int32_t hipidx_save(uint64_t h, const char* path);
int32_t hipidx_load(const char* path, int32_t device, uint64_t* out_handle);

typedef struct hipidx_stats {
    int64_t passes;            /* scan passes so far (one per <= pass_queries queries; several per launch) */
    int64_t queries;           /* queries answered */
    int64_t fallback_queries;  /* queries that took the exhaustive path (certificate failed: massive ties, overflowing lists) */
    int64_t bytes_per_pass;    /* bytes of index the scan kernel reads per pass (algorithmic) */
    int64_t timed_passes;      /* scan LAUNCHES averaged into avg_scan_ms (at most the last 512) */
    float avg_scan_ms;         /* mean HIP-event duration of the scan kernel since timing was enabled, else -1 */
    float avg_scan_wall_ms;    /* same launches on the GPU wall clock, stamped inside the kernel: first wave in -> last wave out */
    float avg_scan_gap_ms;     /* mean idle time between consecutive timed scans (last wave out -> next first wave in) */
    int64_t launches;          /* scan kernel launches so far (each runs 1..launch_queries/pass_queries passes back to back) */
    int64_t roundb_queries;    /* queries whose re-scored prefix the finish had to extend once (more groups within eps of the k-th score) */
    int64_t list_entries;      /* candidate-list entries the scans wrote, summed over all queries answered by the finish */
    int64_t ranked_entries;    /* of those, entries at or above the final bound (what the finish ranks), summed */
    int64_t rescored_groups;   /* 16-row groups whose tagged quad was re-scored in fp64, summed */
} hipidx_stats;
int32_t hipidx_get_stats(uint64_t h, hipidx_stats* out);
/* on = n > 0: HIP events (on the launch stream) around every n-th scan launch, in-kernel wall-clock stamps on every launch;
 * 0 = off.  Two event records cost ~20 us of dispatch bubble between chained launches, hence the sampling.  get_stats syncs. */
int32_t hipidx_enable_timing(uint64_t h, int32_t on);

/* ---- IVF-Flat (BASELINE north_star: "the flat-IP / IVF distance scan"; the reference builds faiss.IndexFlatL2 only,
 *      rag/storage/faiss_index.py:123 -- this is the approximate, low-latency one-query mode on top of the flat index) -------
 * rows_h       a flat index whose rows are stored PERMUTED by inverted list: list l = stored rows
 *              [list_offsets[l], list_offsets[l + 1]), every list starting on a 32-row block (pad with any rows);
 *              orig_ids_host[stored row] = the id results carry, -1 for padding rows
 * centroids_h  a flat index (same d, metric, device) over the nlist centroids, row l = centroid of list l
 * A search scores, exactly (fp64 from the fp32 rows, the flat index's own re-score), every row of the nprobe lists whose
 * centroids rank best for the query under the index's metric, and returns their top k in the canonical order.  It is
 * approximate unless nprobe >= nlist, where the result equals the flat index's bit for bit.  k <= 256, nprobe <= 1000.
 * The handle shares the two flat indexes (destroy it before them). */
int32_t hipivf_create(uint64_t rows_h, uint64_t centroids_h, const int64_t* list_offsets_host, const int64_t* orig_ids_host,
                      int32_t nlist, uint64_t* out_handle);
int32_t hipivf_destroy(uint64_t h);
int32_t hipivf_search_dev(uint64_t h, const float* q_dev, int32_t nq, int32_t k, int32_t nprobe, double* out_scores64_dev,
                          float* out_scores_dev, int64_t* out_ids_dev, void* stream);
int32_t hipivf_info(uint64_t h, int32_t* out_nlist, int64_t* out_stored_rows, int64_t* out_longest_list);

/* ---- partial top-k merge (multi-GPU: after one all-gather of per-shard partial results) ---------------
 * in_scores64 / in_ids: n_parts blocks of [nq, k_in] (device), block p starting part_stride ELEMENTS after block
 * p-1 (0 = dense, nq*k_in) so both arrays can live interleaved in one all-gathered buffer.  Canonical comparator
 * as above; ids < 0 are padding.
 * New capability (the reference is single-process); correctness criterion: sharded == unsharded, bit for bit. */
int32_t hiprag_merge_topk_dev(const double* in_scores64_dev, const int64_t* in_ids_dev, int32_t n_parts, int32_t nq,
                              int32_t k_in, int32_t k_out, int64_t part_stride, int32_t metric,
                              double* out_scores64_dev, float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* ---- BM25 term-at-a-time sparse scoring (named by the reference's README.md:54-58 and rag/config.py:43-45,
 *      implemented nowhere in it; spec in DESIGN.md) -------------------------------------------------------
 * Postings are CSR over term ids, doc ids ascending inside a list, impacts precomputed in fp32 by the host.
 * score(doc) = sum over query terms IN QUERY ORDER of impact (fp32 adds); docs with score <= 0 are excluded;
 * order (score desc, doc id asc); padding id -1 / score -FLT_MAX. */
int32_t hipbm25_create(int64_t n_docs, int64_t n_terms, const uint64_t* offsets_host, const uint32_t* doc_ids_host,
                       const float* impacts_host, int32_t device, uint64_t* out_handle);
int32_t hipbm25_destroy(uint64_t h);
int32_t hipbm25_set_id_base(uint64_t h, int64_t id_base);
int32_t hipbm25_search(uint64_t h, const uint32_t* term_ids_host, const int32_t* q_offsets_host, int32_t nq, int32_t k,
                       float* out_scores, int64_t* out_ids);
int32_t hipbm25_search_dev(uint64_t h, const uint32_t* term_ids_host, const int32_t* q_offsets_host, int32_t nq,
                           int32_t k, double* out_scores64_dev, float* out_scores_dev, int64_t* out_ids_dev,
                           void* stream);
typedef struct hipbm25_stats {
    int64_t queries;
    int64_t postings_touched; /* sum of df over all query terms so far */
    int64_t bytes_algorithmic; /* 8 B per posting + 2*4*N per query (zero + select read) */
} hipbm25_stats;
int32_t hipbm25_get_stats(uint64_t h, hipbm25_stats* out);

/* ---- reciprocal-rank fusion (README.md:54-58 "hybrid"; weights rag/config.py:44-45) ---------------------
 * s(d) = w_a/(c + rank_a(d)) + w_b/(c + rank_b(d)), ranks 1-based, a missing list contributes +0;
 * IEEE fp32 in exactly that order; order (s desc, id asc); ids < 0 in the inputs are padding. */
int32_t hiprrf_fuse(const int64_t* ids_a_host, const int64_t* ids_b_host, int32_t nq, int32_t depth_a, int32_t depth_b,
                    int32_t k, float c, float w_a, float w_b, float* out_scores, int64_t* out_ids);
int32_t hiprrf_fuse_dev(const int64_t* ids_a_dev, const int64_t* ids_b_dev, int32_t nq, int32_t depth_a,
                        int32_t depth_b, int32_t k, float c, float w_a, float w_b, float* out_scores_dev,
                        int64_t* out_ids_dev, void* stream);

/* ---- hybrid fast path: one call = dense top-`depth` + BM25 top-`depth` + RRF -> top-k -------------------
 * What rag/query/retriever.py does per query batch with three calls, for hosts that bind the C-ABI directly: host
 * queries and term lists in, fused fp32 scores / ids out; the two result lists never leave the GPU.  Both handles must
 * live on the same device.  Row-sharded serving uses hiphybrid_shard_begin_dev / _end_dev below (the all-gather sits between
 * search and fusion -- ranks are global, so fusion has to follow the merge).
 * The two legs are independent and bound by different things (the dense scan by HBM, BM25 by LDS round trips), so they run
 * BESIDE each other: the dense leg on the device's scan stream (hiprag_scan_stream) with 96 CUs left out of its scan grid,
 * the BM25 leg on the caller's stream, its workgroups filling those CUs, RRF behind both (1M chunks, 256 queries per call:
 * 140-154 k hybrid queries/s against 123-137 k with the legs one after the other; identical results). */
int32_t hiphybrid_search(uint64_t dense_h, uint64_t bm25_h, const float* q_host, const uint32_t* term_ids_host,
                         const int32_t* q_offsets_host, int32_t nq, int32_t depth, int32_t k, float c, float w_dense,
                         float w_sparse, float* out_scores, int64_t* out_ids);
/* The same with the queries, the two intermediate lists and the results in device memory, ordered on `stream` (the legs
 * start where `stream` is at the call; the fusion is enqueued on `stream` behind both legs; nothing synchronises).
 * lists_dev: int64 [4][nq][depth] the caller provides = dense fp64 score bits | dense ids | BM25 fp64 score bits | BM25 ids
 * (left there for callers that also want the per-leg lists, rag/query/retriever.py keeps both). */
int32_t hiphybrid_search_dev(uint64_t dense_h, uint64_t bm25_h, const float* q_dev, const uint32_t* term_ids_host,
                             const int32_t* q_offsets_host, int32_t nq, int32_t depth, int32_t k, float c, float w_dense,
                             float w_sparse, int64_t* lists_dev, float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* ---- row-sharded hybrid step: the two halves around the caller's ONE all-gather (SURVEY 8b `hiphybrid_search(...)`, 8e) ----
 * One process per GPU holds the rows AND the postings of one contiguous document range (hipidx_set_id_base /
 * hipbm25_set_id_base make the ids global).  A batch is
 *   hiphybrid_shard_begin_dev   local dense top-`depth` + local BM25 top-`depth` -> pack_dev, int64 [2 legs][2][nq][depth]
 *                               (leg 0 dense, leg 1 BM25; [0] = fp64 score bits, [1] = ids).  The index scan is enqueued on
 *                               scan_stream (callers chain their scans there), everything after it on tail_stream, ordered
 *                               behind the scan by a library-owned event; `slot` (0..7) as in hipidx_search_begin_dev.
 *                               scratch_f32_dev: 2 * nq * depth floats.  For the tails to run BESIDE the next scan, pass
 *                               the device's scan stream (hiprag_scan_stream) and leave the tails CUs
 *                               (hipidx_set_spare_cus(dense_h, 96)): see hipidx_set_spare_cus and hiphybrid_search.

 *   (caller)                    all-gather of pack_dev over the ranks -> gathered_dev [n_parts][2][2][nq][depth]: RCCL, MPI,
 *                               whatever the host has; 4 * nq * depth * 8 bytes per rank
 *   hiphybrid_shard_end_dev     each leg merged over the parts with the canonical comparator (better score, then lower id),
 *                               RRF over the two GLOBAL lists -> out_scores_dev / out_ids_dev [nq][k], identical on every rank.
 *                               scratch_dev: 4 * nq * depth int64.
 * hiprag/sharded.py (ShardedHybrid) is one client of these two calls; a host without Python needs nothing else.
 * The reference is a single CPU process (rag/storage/faiss_index.py:83, rag/Dockerfile:23-26): new capability. */
int32_t hiphybrid_shard_begin_dev(uint64_t dense_h, uint64_t bm25_h, const float* q_dev, const uint32_t* term_ids_host,
                                  const int32_t* q_offsets_host, int32_t nq, int32_t depth, int32_t slot, int64_t* pack_dev,
                                  float* scratch_f32_dev, void* scan_stream, void* tail_stream);
int32_t hiphybrid_shard_end_dev(const int64_t* gathered_dev, int32_t n_parts, int32_t nq, int32_t depth, int32_t k,
                                int32_t dense_metric, float c, float w_dense, float w_sparse, int64_t* scratch_dev,
                                float* out_scores_dev, int64_t* out_ids_dev, void* stream);

/* ---- batch encoder: XLM-RoBERTa-large architecture (BGE-M3 embeddings, bge-reranker-v2-m3 cross-encoder) ---------
 * forward     <- HuggingFaceEmbeddings.embed_query / embed_documents (sentence-transformers encode, CLS pooling,
 *                normalize_embeddings=True)            rag/providers/hf/embeddings.py:32-35,54,77
 * score_pairs <- the cross-encoder the reference only configures (RERANKER_MODEL)      rag/config.py:25-27
 * Weights are DEVICE pointers owned by the caller (torch tensors) and must outlive the handle: matrices bf16 in
 * torch.nn.Linear layout [out, in]; biases and LayerNorm parameters fp32.  Token ids / lengths are HOST arrays
 * ([nseq, max_len] int32, right-padded; lengths include BOS/EOS).  Outputs are DEVICE fp32: forward -> [nseq, hidden]
 * L2-normalised CLS embeddings (zeros for length-0 rows); score_pairs -> [nseq] logits.
 * Batches of at most HIPENC_SMALL_ROWS (environment, read at create; default 384, 0 = never) padded token rows -- one
 * query through embed_single, a few short texts -- run the GEMMs as weight-streaming workgroups instead of 128 x 128
 * tiles (1.3 ms instead of 3.3 ms for one query on the 24-layer model); same arithmetic, fixed summation order. */
typedef struct hipenc_config {
    int32_t vocab, hidden, layers, heads, ffn, max_pos, pad_id;
    float ln_eps;
} hipenc_config;
typedef struct hipenc_layer_weights {
    const void *wqkv, *bqkv;     /* [3H, H] bf16 = cat(query, key, value).weight; [3H] f32 */
    const void *wo, *bo;         /* attention.output.dense */
    const void *ln1_g, *ln1_b;   /* attention.output.LayerNorm */
    const void *w1, *b1;         /* intermediate.dense [F, H] */
    const void *w2, *b2;         /* output.dense [H, F] */
    const void *ln2_g, *ln2_b;   /* output.LayerNorm */
} hipenc_layer_weights;
typedef struct hipenc_weights {
    const void *word_emb, *pos_emb, *type_emb;   /* [V,H], [max_pos,H], [H] (row 0 of token_type_embeddings) bf16 */
    const void *emb_ln_g, *emb_ln_b;             /* f32 */
    const hipenc_layer_weights* layers;          /* host array of `layers` entries */
    const void *cls_dense_w, *cls_dense_b, *cls_out_w, *cls_out_b; /* optional head: [H,H] bf16, [H] f32, [H] bf16, [1] f32 */
} hipenc_weights;
int32_t hipenc_create(const hipenc_config* cfg, const hipenc_weights* weights, int32_t device, uint64_t* out_handle);
int32_t hipenc_destroy(uint64_t h);
int32_t hipenc_forward(uint64_t h, const int32_t* token_ids_host, const int32_t* seq_lens_host, int32_t nseq,
                       int32_t max_len, float* out_dev, void* stream);
int32_t hipenc_score_pairs(uint64_t h, const int32_t* token_ids_host, const int32_t* seq_lens_host, int32_t nseq,
                           int32_t max_len, float* out_logits_dev, void* stream);
/* One linear layer of the encoder in isolation: C = A[M,K] * W[N,K]^T with the fused epilogue the encoder uses at that place
 * (0 = QKV: bias, q scaled by 1/8, head-major q / k into out / out_k and TRANSPOSED v into out_vt; 1 = bias + exact-erf GELU
 * -> bf16 [M,N]; 2 = bias + bf16 residual -> f32 [M,N]; 3 = the same sum rounded to bf16 [M,N], the pre-LayerNorm
 * form of the big-batch path).  All pointers are device memory.  impl 0 picks the kernel the way
 * hipenc_forward does, 1 forces 128 x 128 tiles, 2 the persistent 256 x 256 tiles.  For kernel tests and benchmarks
 * (tests/test_encoder_gpu.py, tools/bench_gemm.py): the nn.Linear calls inside sentence-transformers' forward
 * (rag/providers/hf/embeddings.py:54,77) are what it stands for. */
int32_t hipenc_linear(const void* a_dev, const void* w_dev, const float* bias_dev, int32_t M, int32_t N, int32_t K,
                      int32_t epilogue, const void* resid_dev, void* out_dev, void* out_k_dev, void* out_vt_dev, int32_t S,
                      int32_t heads, int32_t impl, void* stream);
int32_t hipenc_last_flops(uint64_t h, double* out_flops); /* algorithmic FLOPs of the last forward */

#ifdef __cplusplus
}
#endif
#endif /* HIPRAG_H */
