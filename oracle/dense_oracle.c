/*
 * oracle/dense_oracle.c -- CPU restatement of the reference's dense search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the reported CPU baseline.  The product path
 * (intool-rag_amd/csrc) never links or calls it.
 *
 * What it restates
 * ----------------
 * The reference searches with faiss.IndexFlatL2 through
 *   /root/reference/rag/storage/faiss_index.py:81-83   (q = float32[1,d]; D,I = index.search(q,k))
 *   /root/reference/rag/storage/faiss_index.py:123-124 (IndexFlatL2(d); index.add(float32[n,d]))
 * faiss-cpu==1.7.4 (rag/requirements.txt:25) is a third-party wheel that is absent from
 * /root/reference and from this image, so the arithmetic is restated from its published
 * behaviour:
 *   IndexFlatL2.search : k smallest squared-L2 distances, ascending; missing -> id -1, dist FLT_MAX
 *   IndexFlatIP.search : k largest inner products, descending;       missing -> id -1, score -FLT_MAX
 * PARITY UNPINNED for the fp32 summation order inside FAISS (no FAISS, no reference tests,
 * no golden vectors exist for it).  The oracle therefore defines truth as:
 *   score = fp64-accumulated sum over the fp32 inputs (products of two fp32 are exact in fp64),
 *   order = (better score first, lower id first on exact ties), reported score = (float)fp64.
 * The wrapper semantics above FAISS (score transform, clamp, -1 passthrough, enrichment,
 * page ranking) ARE pinned: see oracle/hybrid_oracle.py and tests/golden/.
 *
 * Two entry points:
 *   oracle_flat_search_f64 : the truth used by every parity test.
 *   oracle_flat_search_f32 : reference-faithful timing twin -- one query at a time, one thread,
 *                            fp32 8-lane partial sums + bounded heap, the shape of FAISS's nq<20
 *                            non-BLAS path; used only as bench.py's cpu_baseline ("port").
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define METRIC_IP 0
#define METRIC_L2 1

typedef struct {
    double s;
    int64_t id;
} cand_t;

/* "a ranks before b": IP -> larger score first; L2 -> smaller distance first; ties -> lower id. */
static inline int before(int metric, double sa, int64_t ia, double sb, int64_t ib)
{
    if (sa != sb) return metric == METRIC_IP ? (sa > sb) : (sa < sb);
    return ia < ib;
}

/* Bounded "worst-on-top" heap of the k best candidates seen so far. */
static void sift_down(cand_t* h, int n, int i, int metric)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w = i;
        /* w = the WORST of (i, l, r) -> root holds the worst of the kept set */
        if (l < n && before(metric, h[w].s, h[w].id, h[l].s, h[l].id)) w = l;
        if (r < n && before(metric, h[w].s, h[w].id, h[r].s, h[r].id)) w = r;
        if (w == i) return;
        cand_t t = h[i]; h[i] = h[w]; h[w] = t;
        i = w;
    }
}

static void sift_up(cand_t* h, int i, int metric)
{
    while (i > 0) {
        int p = (i - 1) / 2;
        if (before(metric, h[p].s, h[p].id, h[i].s, h[i].id)) {
            cand_t t = h[i]; h[i] = h[p]; h[p] = t;
            i = p;
        } else return;
    }
}

static int cmp_metric; /* qsort context (single-threaded use only) */
static int cand_cmp(const void* a, const void* b)
{
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (before(cmp_metric, x->s, x->id, y->s, y->id)) return -1;
    if (before(cmp_metric, y->s, y->id, x->s, x->id)) return 1;
    return 0;
}

static double dot_f64(const float* a, const float* b, int d)
{
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += (double)a[i] * (double)b[i];
    return s;
}

static double l2_f64(const float* a, const float* b, int d)
{
    double s = 0.0;
    for (int i = 0; i < d; ++i) { double t = (double)a[i] - (double)b[i]; s += t * t; }
    return s;
}

/*
 * Truth.  x: [n,d] row-major fp32, q: [nq,d].  out_scores: [nq,k] fp32, out_scores64: [nq,k] (may be
 * NULL), out_ids: [nq,k] int64.  id_base is added to every returned id (row-sharded indices).
 * Slots past n are padded FAISS-style: id -1, score -FLT_MAX (IP) / FLT_MAX (L2).
 */
int oracle_flat_search_f64(const float* x, int64_t n, int32_t d, const float* q, int32_t nq, int32_t k,
                           int32_t metric, int64_t id_base, float* out_scores, double* out_scores64,
                           int64_t* out_ids)
{
    if (d <= 0 || k <= 0 || nq < 0 || n < 0) return -1;
    cand_t* heap = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
    if (!heap) return -2;
    for (int32_t b = 0; b < nq; ++b) {
        const float* qb = q + (size_t)b * d;
        int hn = 0;
        for (int64_t i = 0; i < n; ++i) {
            const float* xi = x + (size_t)i * d;
            double s = metric == METRIC_IP ? dot_f64(xi, qb, d) : l2_f64(xi, qb, d);
            if (hn < k) {
                heap[hn].s = s; heap[hn].id = i; sift_up(heap, hn, metric); ++hn;
            } else if (before(metric, s, i, heap[0].s, heap[0].id)) {
                heap[0].s = s; heap[0].id = i; sift_down(heap, hn, 0, metric);
            }
        }
        cmp_metric = metric;
        qsort(heap, (size_t)hn, sizeof(cand_t), cand_cmp);
        for (int j = 0; j < k; ++j) {
            size_t o = (size_t)b * k + j;
            if (j < hn) {
                out_scores[o] = (float)heap[j].s;
                if (out_scores64) out_scores64[o] = heap[j].s;
                out_ids[o] = heap[j].id + id_base;
            } else {
                out_scores[o] = metric == METRIC_IP ? -FLT_MAX : FLT_MAX;
                if (out_scores64) out_scores64[o] = metric == METRIC_IP ? -DBL_MAX : DBL_MAX;
                out_ids[o] = -1;
            }
        }
    }
    free(heap);
    return 0;
}

/* All n exact fp64 scores of one query (tests use it to inspect gaps / margins). */
int oracle_all_scores_f64(const float* x, int64_t n, int32_t d, const float* q, int32_t metric, double* out)
{
    for (int64_t i = 0; i < n; ++i)
        out[i] = metric == METRIC_IP ? dot_f64(x + (size_t)i * d, q, d) : l2_f64(x + (size_t)i * d, q, d);
    return 0;
}

/* ---- reference-faithful timing twin (fp32, single thread, one query per call) ------------------- */

typedef struct {
    float s;
    int64_t id;
} cand32_t;

static inline int before32(int metric, float sa, int64_t ia, float sb, int64_t ib)
{
    if (sa != sb) return metric == METRIC_IP ? (sa > sb) : (sa < sb);
    return ia < ib;
}

static void sift_down32(cand32_t* h, int n, int i, int metric)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && before32(metric, h[w].s, h[w].id, h[l].s, h[l].id)) w = l;
        if (r < n && before32(metric, h[w].s, h[w].id, h[r].s, h[r].id)) w = r;
        if (w == i) return;
        cand32_t t = h[i]; h[i] = h[w]; h[w] = t;
        i = w;
    }
}

static void sift_up32(cand32_t* h, int i, int metric)
{
    while (i > 0) {
        int p = (i - 1) / 2;
        if (before32(metric, h[p].s, h[p].id, h[i].s, h[i].id)) {
            cand32_t t = h[i]; h[i] = h[p]; h[p] = t;
            i = p;
        } else return;
    }
}

static int cmp_metric32;
static int cand32_cmp(const void* a, const void* b)
{
    const cand32_t* x = (const cand32_t*)a; const cand32_t* y = (const cand32_t*)b;
    if (before32(cmp_metric32, x->s, x->id, y->s, y->id)) return -1;
    if (before32(cmp_metric32, y->s, y->id, x->s, x->id)) return 1;
    return 0;
}

/* 8 partial sums, the shape of an AVX2 fvec_* kernel; the compiler vectorises it at -O3. */
static float dot_f32(const float* a, const float* b, int d)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= d; i += 8)
        for (int j = 0; j < 8; ++j) acc[j] += a[i + j] * b[i + j];
    float s = ((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]));
    for (; i < d; ++i) s += a[i] * b[i];
    return s;
}

static float l2_f32(const float* a, const float* b, int d)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= d; i += 8)
        for (int j = 0; j < 8; ++j) { float t = a[i + j] - b[i + j]; acc[j] += t * t; }
    float s = ((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]));
    for (; i < d; ++i) { float t = a[i] - b[i]; s += t * t; }
    return s;
}

int oracle_flat_search_f32(const float* x, int64_t n, int32_t d, const float* q, int32_t nq, int32_t k,
                           int32_t metric, float* out_scores, int64_t* out_ids)
{
    if (d <= 0 || k <= 0 || nq < 0 || n < 0) return -1;
    cand32_t* heap = (cand32_t*)malloc(sizeof(cand32_t) * (size_t)k);
    if (!heap) return -2;
    for (int32_t b = 0; b < nq; ++b) { /* one query at a time, as faiss_index.py:81-83 does */
        const float* qb = q + (size_t)b * d;
        int hn = 0;
        for (int64_t i = 0; i < n; ++i) {
            const float* xi = x + (size_t)i * d;
            float s = metric == METRIC_IP ? dot_f32(xi, qb, d) : l2_f32(xi, qb, d);
            if (hn < k) {
                heap[hn].s = s; heap[hn].id = i; sift_up32(heap, hn, metric); ++hn;
            } else if (before32(metric, s, i, heap[0].s, heap[0].id)) {
                heap[0].s = s; heap[0].id = i; sift_down32(heap, hn, 0, metric);
            }
        }
        cmp_metric32 = metric;
        qsort(heap, (size_t)hn, sizeof(cand32_t), cand32_cmp);
        for (int j = 0; j < k; ++j) {
            size_t o = (size_t)b * k + j;
            if (j < hn) { out_scores[o] = heap[j].s; out_ids[o] = heap[j].id; }
            else { out_scores[o] = metric == METRIC_IP ? -FLT_MAX : FLT_MAX; out_ids[o] = -1; }
        }
    }
    free(heap);
    return 0;
}
