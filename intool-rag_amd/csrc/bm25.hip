// bm25.hip -- BM25 term-at-a-time (TAAT) sparse scoring on MI355X.
//
// The reference names hybrid BM25 search (README.md:54-58; rag/config.py:43-45 HYBRID_SEARCH_ENABLED / BM25_WEIGHT)
// but contains no implementation; the specification is restated in oracle/hybrid_oracle.py (bm25_impacts,
// bm25_scores_taat) and DESIGN.md.  Determinism rule: the host precomputes one fp32 IMPACT per posting
// (idf * tf-normalisation, fp64 -> fp32 once); the GPU only performs acc[doc] += impact, one query-term slot per
// kernel launch, in query order.  A document occurs at most once in a posting list, so a slot needs no atomics and
// the fp32 sum of every document is formed in exactly the oracle's order.
//
// Layout: CSR postings in HBM as two SoA streams (doc_ids u32, impacts f32) -- 8 B per posting, read once per use,
// coalesced 16 B per lane; accumulators acc[query][doc] fp32 stay L2 / Infinity-Cache resident (4 MB per query per
// 1M docs).  Bound: HBM (posting streams) + the accumulator zero/select passes; bytes reported by hipbm25_get_stats.
#include <cfloat>
#include <vector>

#include "common.h"
#include "topk_device.h"

namespace hiprag {
namespace {

constexpr int kBatch = 32;            // queries scored concurrently
constexpr int kTaatThreads = 256;
constexpr int kTaatPerThread = 8;     // postings per thread
constexpr int kTaatChunk = kTaatThreads * kTaatPerThread;
constexpr int kMaxK = 1000;

struct SlotRange {  // posting range of one (query, term slot)
    unsigned long long lo, hi;
};

// grid (chunks, nq): chunk c of query b's posting list for this slot.
__global__ __launch_bounds__(kTaatThreads) void taat_kernel(const u32* __restrict__ doc_ids, const float* __restrict__ impacts,
                                                           const SlotRange* __restrict__ ranges, float* __restrict__ acc,
                                                           i64 n_docs)
{
    const int b = blockIdx.y;
    const SlotRange r = ranges[b];
    const u64 start = r.lo + (u64)blockIdx.x * kTaatChunk;
    if (start >= r.hi) return;
    float* accb = acc + (i64)b * n_docs;  // n_docs here is the padded row stride
    // strided by thread so that each load instruction is a contiguous 1 KiB (u32/f32 x 256 threads)
#pragma unroll
    for (int j = 0; j < kTaatPerThread; ++j) {
        const u64 i = start + (u64)j * kTaatThreads + threadIdx.x;
        if (i < r.hi) {
            const u32 d = doc_ids[i];
            accb[d] = accb[d] + impacts[i];
        }
    }
}

struct FinishArgs {
    const u64* ck;
    const i64* ci;
    double* out64;
    float* out32;
    i64* out_ids;
    i64 ncand, id_base;
    int k;
};

__global__ __launch_bounds__(256) void bm25_finish_kernel(FinishArgs a)
{
    extern __shared__ unsigned char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    i64* ids = reinterpret_cast<i64*>(keys + kTile);
    u64* selk = reinterpret_cast<u64*>(ids + kTile);
    i64* seli = reinterpret_cast<i64*>(selk + a.k);
    KeyId* red = reinterpret_cast<KeyId*>(seli + a.k);
    const int q = blockIdx.x;
    const u64* sk = a.ck + (i64)q * a.ncand;
    const i64* si = a.ci + (i64)q * a.ncand;
    wg_stream_topk<256>([&](i64 i, u64& k, i64& id) { k = sk[i]; id = si[i]; }, a.ncand, a.k, keys, ids, red, selk, seli);
    for (int r = threadIdx.x; r < a.k; r += 256) {
        const i64 o = (i64)q * a.k + r;
        const u64 k = selk[r];
        const float s = k ? unord32((u32)(k >> 32)) : -FLT_MAX;
        if (a.out64) a.out64[o] = k ? (double)s : -DBL_MAX;
        if (a.out32) a.out32[o] = s;
        a.out_ids[o] = k ? seli[r] + a.id_base : -1;
    }
}

struct Bm25Index {
    std::mutex mu;
    int device = 0;
    i64 n_docs = 0, n_terms = 0, n_postings = 0, id_base = 0;
    std::vector<uint64_t> offsets;  // host copy: planning happens on the host
    DevBuf doc_ids, impacts, acc, ranges, ck, ci, o64, o32, oid;
    int ws_k = 0;
    i64 queries = 0, postings_touched = 0, bytes_alg = 0;

    i64 nchunks() const { return std::max<i64>(1, (n_docs + kTile - 1) / kTile); }
    i64 stride() const { return std::max<i64>(4, (n_docs + 3) / 4 * 4); }       // accumulator row stride (16-B aligned rows)
    i64 nlists() const { return (std::max<i64>(1, (n_docs + kSelPerWave - 1) / kSelPerWave) + 3) / 4 * 4; }

    int32_t reserve(int k)
    {
        int32_t rc;
        if ((rc = acc.reserve((size_t)kBatch * stride() * sizeof(float)))) return rc;
        if (k > ws_k) {
            const i64 per_q = std::max(nchunks(), nlists());
            if ((rc = ck.reserve((size_t)kBatch * per_q * k * sizeof(u64)))) return rc;
            if ((rc = ci.reserve((size_t)kBatch * per_q * k * sizeof(i64)))) return rc;
            ws_k = k;
        }
        return HIPRAG_OK;
    }

    int32_t search_dev(const uint32_t* terms, const int32_t* qoff, int nq, int k, double* o64p, float* o32p, i64* oidp,
                       hipStream_t st)
    {
        int32_t rc = reserve(k);
        if (rc) return rc;
        for (int q0 = 0; q0 < nq; q0 += kBatch) {
            const int m = std::min(kBatch, nq - q0);
            int max_terms = 0;
            for (int b = 0; b < m; ++b) max_terms = std::max(max_terms, qoff[q0 + b + 1] - qoff[q0 + b]);
            // slot-major plan: ranges[slot][b]
            std::vector<SlotRange> plan((size_t)std::max(max_terms, 1) * m);
            std::vector<u64> slot_max(std::max(max_terms, 1), 0);
            for (int s = 0; s < max_terms; ++s)
                for (int b = 0; b < m; ++b) {
                    SlotRange r{0, 0};
                    const int nt = qoff[q0 + b + 1] - qoff[q0 + b];
                    if (s < nt) {
                        const uint32_t t = terms[qoff[q0 + b] + s];
                        if ((i64)t < n_terms) { r.lo = offsets[t]; r.hi = offsets[t + 1]; }  // unknown terms score nothing
                    }
                    plan[(size_t)s * m + b] = r;
                    slot_max[s] = std::max<u64>(slot_max[s], r.hi - r.lo);
                    postings_touched += (i64)(r.hi - r.lo);
                    bytes_alg += (i64)(r.hi - r.lo) * 8;
                }
            if ((rc = ranges.reserve(plan.size() * sizeof(SlotRange)))) return rc;
            HR_CHECK_HIP(hipMemcpyAsync(ranges.p, plan.data(), plan.size() * sizeof(SlotRange), hipMemcpyHostToDevice, st));
            HR_CHECK_HIP(hipStreamSynchronize(st));  // plan is a stack-local vector; tiny copy
            HR_CHECK_HIP(hipMemsetAsync(acc.p, 0, (size_t)m * stride() * sizeof(float), st));
            for (int s = 0; s < max_terms; ++s) {
                if (slot_max[s] == 0) continue;
                const unsigned gx = (unsigned)((slot_max[s] + kTaatChunk - 1) / kTaatChunk);
                hipLaunchKernelGGL(taat_kernel, dim3(gx, m), dim3(kTaatThreads), 0, st, doc_ids.as<u32>(), impacts.as<float>(),
                                   ranges.as<SlotRange>() + (size_t)s * m, acc.as<float>(), stride());
            }
            double* o64q = o64p ? o64p + (i64)q0 * k : nullptr;
            float* o32q = o32p ? o32p + (i64)q0 * k : nullptr;
            i64* oidq = oidp + (i64)q0 * k;
            const i64 wave_cand = nlists() * k;
            if (k <= 64 && wave_cand <= 16 * 64 * 16) {
                // fast selectors (topk_device.h): one wave filters 4096 accumulators, then a 16-wave merge per query
                hipLaunchKernelGGL(select_wave_kernel<true>, dim3((unsigned)(nlists() / 4), m), dim3(256), 0, st,
                                   (const float*)acc.as<float>(), (i64)stride(), (i64)n_docs, k, ck.as<u64>(), ci.as<i64>());
                const i64 per_lane = (wave_cand + 1023) / 1024;
                auto mk = merge_packed_kernel<16>;
                if (per_lane <= 1) mk = merge_packed_kernel<1>;
                else if (per_lane <= 2) mk = merge_packed_kernel<2>;
                else if (per_lane <= 4) mk = merge_packed_kernel<4>;
                else if (per_lane <= 8) mk = merge_packed_kernel<8>;
                hipLaunchKernelGGL(mk, dim3(m), dim3(1024), 0, st, (const u64*)ck.as<u64>(), (const i64*)ci.as<i64>(), wave_cand, k,
                                   id_base, o64q, o32q, oidq);
            } else {
                hipLaunchKernelGGL(select_f32_kernel<true>, dim3((unsigned)nchunks(), m), dim3(256), 0, st,
                                   (const float*)acc.as<float>(), (i64)stride(), (i64)n_docs, k, ck.as<u64>(), ci.as<i64>());
                FinishArgs fa;
                fa.ck = ck.as<u64>(); fa.ci = ci.as<i64>();
                fa.out64 = o64q; fa.out32 = o32q; fa.out_ids = oidq;
                fa.ncand = nchunks() * k; fa.id_base = id_base; fa.k = k;
                const size_t lds = (size_t)kTile * 16 + (size_t)k * 16 + 2 * 4 * sizeof(KeyId);
                HR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bm25_finish_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(bm25_finish_kernel, dim3(m), dim3(256), lds, st, fa);
            }
            HR_CHECK_HIP(hipGetLastError());
            queries += m;
            bytes_alg += (i64)m * n_docs * 8;
        }
        return HIPRAG_OK;
    }
};

Registry<Bm25Index>& reg()
{
    static Registry<Bm25Index> r;
    return r;
}

#define GET_BM25(h)                                                                          \
    std::shared_ptr<Bm25Index> ix = reg().get(h);                                            \
    if (!ix) { set_error("unknown bm25 handle %llu", (unsigned long long)(h)); return HIPRAG_E_HANDLE; } \
    std::lock_guard<std::mutex> guard(ix->mu);                                               \
    HR_CHECK_HIP(hipSetDevice(ix->device))

}  // namespace
}  // namespace hiprag

using namespace hiprag;

extern "C" {

int32_t hipbm25_create(int64_t n_docs, int64_t n_terms, const uint64_t* offsets_host, const uint32_t* doc_ids_host,
                       const float* impacts_host, int32_t device, uint64_t* out_handle)
{
    HR_REQUIRE(out_handle && offsets_host, "null argument");
    HR_REQUIRE(n_docs >= 0 && n_terms >= 0, "negative sizes");
    HR_REQUIRE(n_docs < (1ll << 32), "doc ids are u32: n_docs must be < 2^32 per shard");
    const uint64_t P = offsets_host[n_terms];
    HR_REQUIRE(offsets_host[0] == 0, "offsets[0] must be 0");
    for (int64_t t = 0; t < n_terms; ++t)
        HR_REQUIRE(offsets_host[t] <= offsets_host[t + 1], "offsets must be non-decreasing (term %lld)", (long long)t);
    HR_REQUIRE(P == 0 || (doc_ids_host && impacts_host), "null postings");
    for (uint64_t i = 0; i < P; ++i)
        HR_REQUIRE((int64_t)doc_ids_host[i] < n_docs, "posting %llu has doc id %u >= n_docs", (unsigned long long)i, doc_ids_host[i]);
    auto ix = std::make_shared<Bm25Index>();
    ix->device = device;
    ix->n_docs = n_docs;
    ix->n_terms = n_terms;
    ix->n_postings = (i64)P;
    ix->offsets.assign(offsets_host, offsets_host + n_terms + 1);
    HR_CHECK_HIP(hipSetDevice(device));
    int32_t rc;
    if ((rc = ix->doc_ids.reserve(std::max<size_t>(16, P * sizeof(u32))))) return rc;
    if ((rc = ix->impacts.reserve(std::max<size_t>(16, P * sizeof(float))))) return rc;
    if (P) {
        HR_CHECK_HIP(hipMemcpy(ix->doc_ids.p, doc_ids_host, P * sizeof(u32), hipMemcpyHostToDevice));
        HR_CHECK_HIP(hipMemcpy(ix->impacts.p, impacts_host, P * sizeof(float), hipMemcpyHostToDevice));
    }
    *out_handle = reg().put(ix);
    return HIPRAG_OK;
}

int32_t hipbm25_destroy(uint64_t h)
{
    std::shared_ptr<Bm25Index> ix = reg().get(h);
    if (!ix) { set_error("unknown bm25 handle"); return HIPRAG_E_HANDLE; }
    {
        std::lock_guard<std::mutex> guard(ix->mu);
        (void)hipSetDevice(ix->device);
        (void)hipDeviceSynchronize();
    }
    reg().erase(h);
    return HIPRAG_OK;
}

int32_t hipbm25_set_id_base(uint64_t h, int64_t id_base)
{
    GET_BM25(h);
    ix->id_base = id_base;
    return HIPRAG_OK;
}

static int32_t validate_queries(const uint32_t* term_ids_host, const int32_t* q_offsets_host, int32_t nq, int32_t k)
{
    HR_REQUIRE(nq >= 0 && k > 0 && k <= kMaxK, "bad nq/k (k must be in 1..%d)", kMaxK);
    if (nq == 0) return HIPRAG_OK;
    HR_REQUIRE(q_offsets_host, "null q_offsets");
    HR_REQUIRE(q_offsets_host[0] >= 0, "q_offsets must start >= 0");
    for (int b = 0; b < nq; ++b) HR_REQUIRE(q_offsets_host[b] <= q_offsets_host[b + 1], "q_offsets must be non-decreasing");
    HR_REQUIRE(term_ids_host || q_offsets_host[nq] == q_offsets_host[0], "null term ids");
    return HIPRAG_OK;
}

int32_t hipbm25_search_dev(uint64_t h, const uint32_t* term_ids_host, const int32_t* q_offsets_host, int32_t nq, int32_t k,
                           double* out_scores64_dev, float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    GET_BM25(h);
    int32_t rc = validate_queries(term_ids_host, q_offsets_host, nq, k);
    if (rc || nq == 0) return rc;
    HR_REQUIRE(out_ids_dev, "null output");
    return ix->search_dev(term_ids_host, q_offsets_host, nq, k, out_scores64_dev, out_scores_dev, (i64*)out_ids_dev,
                          (hipStream_t)stream);
}

int32_t hipbm25_search(uint64_t h, const uint32_t* term_ids_host, const int32_t* q_offsets_host, int32_t nq, int32_t k,
                       float* out_scores, int64_t* out_ids)
{
    GET_BM25(h);
    int32_t rc = validate_queries(term_ids_host, q_offsets_host, nq, k);
    if (rc || nq == 0) return rc;
    HR_REQUIRE(out_scores && out_ids, "null output");
    if ((rc = ix->o32.reserve((size_t)nq * k * sizeof(float)))) return rc;
    if ((rc = ix->oid.reserve((size_t)nq * k * sizeof(i64)))) return rc;
    rc = ix->search_dev(term_ids_host, q_offsets_host, nq, k, nullptr, ix->o32.as<float>(), ix->oid.as<i64>(), nullptr);
    if (rc) return rc;
    HR_CHECK_HIP(hipMemcpy(out_scores, ix->o32.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost));
    HR_CHECK_HIP(hipMemcpy(out_ids, ix->oid.p, (size_t)nq * k * sizeof(i64), hipMemcpyDeviceToHost));
    return HIPRAG_OK;
}

int32_t hipbm25_get_stats(uint64_t h, hipbm25_stats* out)
{
    GET_BM25(h);
    HR_REQUIRE(out, "null out");
    out->queries = ix->queries;
    out->postings_touched = ix->postings_touched;
    out->bytes_algorithmic = ix->bytes_alg;
    return HIPRAG_OK;
}

}  // extern "C"
