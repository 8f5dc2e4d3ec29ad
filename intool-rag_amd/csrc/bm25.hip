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
#include <algorithm>
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
    // strided by thread so that each load instruction is a contiguous 1 KiB (u32/f32 x 256 threads).  A document
    // occurs at most once in a posting list, so the eight read-modify-writes of a thread never alias: all eight
    // accumulator loads go out together (written as three separate loops -- left as acc[d] += x the compiler must assume
    // aliasing and serialises eight L2 round trips per thread, which made the kernel latency-bound at ~1 TB/s).
    u32 d[kTaatPerThread];
    float im[kTaatPerThread], a[kTaatPerThread];
#pragma unroll
    for (int j = 0; j < kTaatPerThread; ++j) {
        const u64 i = start + (u64)j * kTaatThreads + threadIdx.x;
        const bool ok = i < r.hi;
        d[j] = ok ? doc_ids[i] : 0xFFFFFFFFu;
        im[j] = ok ? impacts[i] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kTaatPerThread; ++j) a[j] = d[j] != 0xFFFFFFFFu ? accb[d[j]] : 0.f;
#pragma unroll
    for (int j = 0; j < kTaatPerThread; ++j)
        if (d[j] != 0xFFFFFFFFu) accb[d[j]] = a[j] + im[j];
}

// ---------------------------------------------------------------------------------------------------------
// Tiled TAAT (the default path, k <= 64): one workgroup per (document tile, query) keeps the tile's fp32 accumulators in
// LDS (9216 documents = 36 KiB: FOUR workgroups per CU.  A workgroup spends its ~17 us mostly waiting -- six term slots x
// (barrier + LDS read-modify-write round trip), then the pass over the accumulators -- so workgroups per CU are the lever:
// 16384-document tiles (two per CU) 370 k queries/s at 1M documents, 12288 (three) 400-418 k, 9216 (four) 411-417 k, 8192
// (four, more tiles) 398 k; 13312 fit three on paper only), walks the query's term slots IN ORDER -- for each slot it streams only the postings
// whose document falls into the tile -- and then selects the tile's top-k straight out of LDS.  HBM traffic is the posting
// streams alone: no accumulator array to zero, scatter into and scan again (the global-accumulator form below moves
// 8 B per posting plus up to two 128-B lines per touched accumulator through L2, and ran at ~1.2 TB/s).
// Where a list's tile range starts comes from a SKIP TABLE built at create time for every list with >= kSkipMinDf
// postings (the posting offset at each tile boundary, u32 relative to the list start, (ntiles + 1) entries); shorter
// lists are read whole by every tile and filtered -- 109 tiles x < 2048 postings costs less than a table lookup chain.
// Determinism: a document occurs at most once per list, so within a slot no two threads touch the same accumulator and
// the barrier between slots keeps every document's fp32 sum in query-term order, exactly the oracle's.
// Selection: grid (nq, ntiles), queries fastest.  Two bounds per query, zeroed per launch: hist[q] (score histogram of the
// documents emitted so far, see the kernel) and theta[q], which carries the best K-th score any finished
// wave of query q has published, as order-preserving u32 bits; a wave whose own maximum is below it emits nothing, one
// with few survivors above it skips the threshold bisection.  A bound is only ever a score K documents already reached,
// so dropping what lies strictly below it cannot change the merged top-k (ties at the bound are kept).
// ---------------------------------------------------------------------------------------------------------
constexpr int kTileDocs = 9216;
constexpr int kTileThreads = 256;   // workgroup of the tiled kernel
constexpr int kTileWaves = kTileThreads / 64;
constexpr int kHistBuckets = 2048;  // score buckets of the per-query bound histogram (+ 1 word: highest bucket used)
constexpr u64 kSkipMinDf = 2048;
constexpr int kMaxSlots = 64;      // query terms the tiled kernel takes (longer queries use the global-accumulator form)

struct TileSlot {   // posting range of one (query, term slot); skip = first entry of the list's skip table or -1
    unsigned long long lo, hi;
    long long skip;
};

// IDX32: fewer than 2^30 postings in all, so posting indices and their byte offsets fit 32 bits -- the stream's index
// arithmetic (a third of its instructions as 64-bit adds, compares and selects) becomes single 32-bit operations.
template <bool IDX32>
__global__ __launch_bounds__(kTileThreads) void taat_tile_kernel(const u32* __restrict__ doc_ids, const float* __restrict__ impacts,
                                                        const TileSlot* __restrict__ slots, const int* __restrict__ nslots,
                                                        const u32* __restrict__ skip, int max_slots, i64 n_docs, int K1,
                                                        u64* __restrict__ ck, i64* __restrict__ ci, u32* __restrict__ theta,
                                                        u32* __restrict__ hist)
{
    extern __shared__ float tacc[];  // kTileDocs accumulators
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // queries fastest: the tiles of one query are spread over the whole launch, so the bound its early tiles publish
    // (theta, below) is there when the later ones select; neighbours in time share a tile's postings of common terms
    const int q = blockIdx.x, tile = blockIdx.y;
    const u32 tlo = (u32)tile * kTileDocs;
    const u32 tlen = (u32)min((i64)kTileDocs, n_docs - (i64)tlo);
    // posting ranges of this tile for every slot, resolved up front (slot descriptor -> skip table is a chain of two
    // dependent global loads; done per slot inside the loop it cost ~4 us of latency six times per workgroup)
    __shared__ unsigned long long ra[kMaxSlots], rb[kMaxSlots];
    const int ns = min(nslots[q], kMaxSlots);
    if (tid < ns) {
        const TileSlot sl = slots[(i64)q * max_slots + tid];
        u64 a = sl.lo, b = sl.hi;
        if (sl.skip >= 0) {
            a = sl.lo + skip[sl.skip + tile];
            b = sl.lo + skip[sl.skip + tile + 1];
        }
        ra[tid] = a;
        rb[tid] = b;
    }
    for (int i = tid; i < kTileDocs / 4; i += kTileThreads) reinterpret_cast<float4*>(tacc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // The slots form ONE stream of 2048-posting chunks: the loads of the next chunks -- of the same slot or of the next
    // ones -- are in flight while this chunk's accumulator updates run; a workgroup barrier separates slots only.
    constexpr int U = 2048 / kTileThreads;
    constexpr u64 CH = (u64)U * kTileThreads;
    // Loads are UNCONDITIONAL (lanes past the end of a range read posting 0, the buffers never have fewer than four
    // entries) and nothing touches the loaded values before the accumulate step: a load behind a lane mask or a
    // branch is compiled into branch + load + wait, i.e. one memory round trip per posting (this loop used to spend
    // 16 us per workgroup that way), and it makes the number of loads in flight unknown where the paths join.
    auto fetch = [&](u64 i0, u64 b, u32 (&d)[U], float (&im)[U]) {
#pragma unroll
        for (int j = 0; j < U; ++j) {   // eight independent 1 KiB-per-instruction loads of each stream
            if (IDX32) {
                const u32 i = (u32)i0 + (u32)j * (u32)kTileThreads + (u32)tid;
                const u32 ic = i < (u32)b ? i : 0u;
                d[j] = doc_ids[ic];
                im[j] = impacts[ic];
            } else {
                const u64 i = i0 + (u64)j * kTileThreads + tid;
                const u64 ic = i < b ? i : 0;
                d[j] = doc_ids[ic];
                im[j] = impacts[ic];
            }
        }
    };
    float wmax = 0.f;
    int cs = 0;              // cursor: next chunk to fetch = [cpos, ..) of slot cs
    u64 cpos = ns > 0 ? ra[0] : 0;
    auto next_chunk = [&](int& slot, u64& pos, u64& bound) -> bool {
        while (cs < ns && cpos >= rb[cs]) { ++cs; if (cs < ns) cpos = ra[cs]; }
        if (cs >= ns) { pos = 0; bound = 0; return false; }
        slot = cs; pos = cpos; bound = rb[cs];
        cpos += CH;
        return true;
    };
    // D chunks in flight (a tile's share of a list is usually one or two chunks, so this is "the next slots are already
    // on their way")
    constexpr int D = 4;
    u32 dR[D][U];
    float mR[D][U];
    int sR[D];
    bool hR[D];
    u64 pR[D], bR[D];
#pragma unroll
    for (int r = 0; r < D; ++r) {
        sR[r] = 0;
        hR[r] = next_chunk(sR[r], pR[r], bR[r]);
        fetch(pR[r], bR[r], dR[r], mR[r]);
    }
    while (hR[0]) {
#pragma unroll
        for (int r = 0; r < D; ++r) {
            if (!hR[r]) break;   // chunks are handed out in order: the first empty ring entry ends the stream
            // a chunk lies inside one list, so its documents are distinct: read all eight accumulators, then write
            // them (as `+=` the compiler must assume aliasing and runs eight dependent LDS round trips)
            u32 dd[U];
            float cur[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const u32 x = dR[r][j] - tlo;
                const bool inr = IDX32 ? ((u32)pR[r] + (u32)j * (u32)kTileThreads + (u32)tid < (u32)bR[r])
                                       : (pR[r] + (u64)j * kTileThreads + tid < bR[r]);
                dd[j] = (inr && x < tlen) ? x : 0xFFFFFFFFu;
                cur[j] = tacc[dd[j] != 0xFFFFFFFFu ? dd[j] : 0u];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (dd[j] != 0xFFFFFFFFu) {
                    const float nv = cur[j] + mR[r][j];
                    tacc[dd[j]] = nv;
                    wmax = fmaxf(wmax, nv);   // impacts are positive: the largest value ever written is the tile's best final score
                }
            }
            const int nr = (r + 1) % D;              // the chunk that follows in stream order (static after unrolling)
            const bool boundary = !hR[nr] || sR[nr] != sR[r];
            hR[r] = next_chunk(sR[r], pR[r], bR[r]); // re-arm this entry before the barrier: its loads fly through it
            fetch(pR[r], bR[r], dR[r], mR[r]);
            if (boundary) __syncthreads();           // slot boundary: later slots add to the same documents
        }
    }
    __shared__ float smax[kTileWaves];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
    if (lane == 0) smax[wv] = wmax;
    __syncthreads();
    float tile_top = 0.f;
#pragma unroll
    for (int w = 0; w < kTileWaves; ++w) tile_top = fmaxf(tile_top, smax[w]);
    // ---- tile top-K1 out of LDS: wave w filters accumulators [4096 w, 4096 (w + 1)) like select_wave_kernel<true> ----
    // theta[q] = the best K1-th score any finished wave of this query has published (ordered-integer form, grows
    // monotonically): a lower bound of the query's final K1-th score, so nothing below it can reach the result.  The
    // first tiles of a query pay the full selection; later ones start from a threshold that is already almost final
    // and keep only a handful of documents.  Which tiles profit depends on timing, the merged top-k does not.
    u32 th = __hip_atomic_load(theta + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // A second, much tighter bound: hist[q] counts the documents every finished wave of this query has emitted, by score
    // bucket (sign-free float bits >> 20: exponent and three mantissa bits, 12.5 % wide); hist[q][kHistBuckets] is the
    // highest bucket used.  If the buckets from b upwards hold K1 documents, K1 documents score at least b's lower edge.
    // theta alone is the best K1-th score of ONE wave's 2304 documents -- about the 10,000th best of 1M -- and left every
    // wave ~50 survivors to compact and rank (and often a bisection first); this bound follows the query's true K1-th
    // score to within a bucket after the first few tiles and leaves most waves none.
    if (hist) {
        const u32* hq = hist + (size_t)q * (kHistBuckets + 1);
        const u32 hm = __hip_atomic_load(hq + kHistBuckets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hm != 0) {
            const int b = (int)hm - 63 + lane;   // lanes cover the 64 buckets up to the highest one: a factor 256 in score
            u32 suffix = b > 0 ? __hip_atomic_load(hq + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u32 up = (u32)__shfl_down((int)suffix, off);
                if (lane + off < 64) suffix += up;
            }
            const unsigned long long okm = __ballot(suffix >= (u32)K1);
            if (okm) {
                const int l = 63 - __builtin_clzll(okm);
                const u32 hb = 0x80000000u | ((u32)((int)hm - 63 + l) << 20);   // ord32 image of the bucket's lower edge
                th = hb > th ? hb : th;
            }
        }
    }
    // the whole tile is below the bound: no pass over the accumulators at all (about 40 % of the tiles once the bound has
    // converged -- a query's best ~100 documents leave that share of its 109 tiles without any of them)
    if (th != 0 && (!(tile_top > 0.f) || th > ord32(tile_top))) {
        const i64 o0 = (((i64)q * gridDim.y + tile) * kTileWaves + wv) * K1;
        if (lane < K1) { ck[o0 + lane] = 0; ci[o0 + lane] = -1; }
        return;
    }
    constexpr int NV = kTileDocs / (kTileWaves * 256);   // float4 per lane: a wave filters its share of the tile
    float v[NV * 4];
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const float4 x = reinterpret_cast<const float4*>(tacc)[wv * (kTileDocs / kTileWaves / 4) + it * 64 + lane];
        v[it * 4 + 0] = x.x; v[it * 4 + 1] = x.y; v[it * 4 + 2] = x.z; v[it * 4 + 3] = x.w;
    }
    float m = -INFINITY;
#pragma unroll
    for (int n = 0; n < NV * 4; ++n) m = fmaxf(m, v[n] > 0.f ? v[n] : -INFINITY);
    const u32 mo = m == -INFINITY ? 0u : ord32(m);
    u32 best = 0;
    u32 top = mo;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) top = max(top, (u32)__shfl_xor((int)top, off));
#pragma unroll
    for (int n = 0; n < NV * 4; ++n) v[n] = v[n] > 0.f ? v[n] : -INFINITY;
    if (th != 0 && th > top) {
        best = th;                       // nothing in this wave's range can reach the result
    } else if (th != 0) {
        const float tf = unord32(th);
        int c = 0;
#pragma unroll
        for (int n = 0; n < NV * 4; ++n) c += __popcll(__ballot(v[n] >= tf));
        best = c > K1 + (K1 >> 1) ? wave_bisect_threshold(v, th, top + 1, K1, 10) : th;
    } else {
        for (int j = 0; j < 64; ++j) {
            const u32 mj = (u32)__builtin_amdgcn_readlane((int)mo, j);
            const int cnt = __popcll(__ballot(mo >= mj));
            if (mj != 0 && cnt >= K1 && mj > best) best = mj;
        }
        if (best) best = wave_bisect_threshold(v, best, top + 1, K1, 10);
    }
    // Survivors: positive accumulators at or above the threshold.  When there are at most 128 of them (the bisection
    // aims at K1 .. 1.5 K1, a bound inherited from other tiles leaves fewer) they are COMPACTED into this wave's own
    // quarter of the tile (its accumulators are in registers by now; no other wave reads that quarter) and RANKED --
    // every lane counts the keys above its own one or two, keys are unique (score bits | document), so rank = final
    // position: ~64 ballots + ~m broadcast reads instead of 64 conditional serial inserts into a sorted wave list.
    const float thrf = best ? unord32(best) : 0.f;
    const i64 o = (((i64)q * gridDim.y + tile) * kTileWaves + wv) * K1;
    if (tlen < (u32)kTileDocs) {   // last, partial tile only: accumulators past the end of the collection never count
#pragma unroll
        for (int n = 0; n < NV * 4; ++n) {
            const u32 local = (u32)wv * (u32)(kTileDocs / kTileWaves) + (u32)(n >> 2) * 256u + (u32)lane * 4u + (u32)(n & 3);
            if (local >= tlen) v[n] = -INFINITY;
        }
    }
    // one pass: a survivor (v >= threshold; non-positive accumulators are -inf by now) is rare -- ~75 of a wave's 4096 --
    // so most of the 64 steps are one compare and one scalar branch
    u64* comp = reinterpret_cast<u64*>(tacc + wv * (kTileDocs / kTileWaves));
    int ms = 0;
#pragma unroll
    for (int n = 0; n < NV * 4; ++n) {
        const bool keep = v[n] >= thrf;
        const unsigned long long mask = __ballot(keep);
        if (mask) {
            if (keep) {
                const int pos = ms + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                const u32 local = (u32)wv * (u32)(kTileDocs / kTileWaves) + (u32)(n >> 2) * 256u + (u32)lane * 4u + (u32)(n & 3);
                if (pos < 128) comp[pos] = pack_key(v[n], tlo + local);
            }
            ms += __popcll(mask);
        }
    }
    if (ms <= 128) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS writes are done before it reads them back
        const u64 k0 = lane < ms ? comp[lane] : 0ull, k1 = 64 + lane < ms ? comp[64 + lane] : 0ull;
        int r0 = 0, r1 = 0;
#pragma unroll 4
        for (int j = 0; j < ms; ++j) {
            const u64 kj = comp[j];   // same address in every lane: an LDS broadcast
            r0 += kj > k0;
            r1 += kj > k1;
        }
        if (lane < K1 && lane >= ms) { ck[o + lane] = 0; ci[o + lane] = -1; }
        if (k0 != 0 && r0 < K1) { ck[o + r0] = k0 & 0xFFFFFFFF00000000ull; ci[o + r0] = (i64)packed_index(k0); }
        if (k1 != 0 && r1 < K1) { ck[o + r1] = k1 & 0xFFFFFFFF00000000ull; ci[o + r1] = (i64)packed_index(k1); }
        if (hist) {
            u32* hq = hist + (size_t)q * (kHistBuckets + 1);
            u32 bmax = 0;
            if (k0 != 0 && r0 < K1) { const u32 b = ((u32)(k0 >> 32) >> 20) & (kHistBuckets - 1); atomicAdd(hq + b, 1u); bmax = b; }
            if (k1 != 0 && r1 < K1) { const u32 b = ((u32)(k1 >> 32) >> 20) & (kHistBuckets - 1); atomicAdd(hq + b, 1u); bmax = max(bmax, b); }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) bmax = max(bmax, (u32)__shfl_xor((int)bmax, off));
            if (lane == 0 && bmax != 0) atomicMax(hq + kHistBuckets, bmax);
        }
        // this wave holds K1 documents at or above the key of rank K1 - 1: publish the bound
        if (k0 != 0 && r0 == K1 - 1) atomicMax(theta + q, (u32)(k0 >> 32));
        if (k1 != 0 && r1 == K1 - 1) atomicMax(theta + q, (u32)(k1 >> 32));
    } else {
        WaveListPacked L;
        L.init();
        u64 tau = best ? ((u64)best << 32) - 1 : 0;
#pragma unroll
        for (int n = 0; n < NV * 4; ++n) {
            const u32 local = (u32)wv * (u32)(kTileDocs / kTileWaves) + (u32)(n >> 2) * 256u + (u32)lane * 4u + (u32)(n & 3);
            u64 c = 0;
            if (v[n] > 0.f && local < tlen) c = pack_key(v[n], tlo + local);
            tau = L.offer(c, K1, tau);
        }
        if (lane < K1) {
            ck[o + lane] = L.e & 0xFFFFFFFF00000000ull;
            ci[o + lane] = L.e ? (i64)packed_index(L.e) : -1;
        }
        if (hist) {
            u32* hq = hist + (size_t)q * (kHistBuckets + 1);
            u32 bmax = 0;
            if (lane < K1 && L.e != 0) { const u32 b = ((u32)(L.e >> 32) >> 20) & (kHistBuckets - 1); atomicAdd(hq + b, 1u); bmax = b; }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) bmax = max(bmax, (u32)__shfl_xor((int)bmax, off));
            if (lane == 0 && bmax != 0) atomicMax(hq + kHistBuckets, bmax);
        }
        const u64 kth = L.kth(K1);   // this wave holds K1 documents at or above it: publish the bound
        if (lane == 0 && kth != 0) atomicMax(theta + q, (u32)(kth >> 32));
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
    DevBuf doc_ids, impacts, acc, ranges, ck, ci, o64, o32, oid, skip_dev, slots_dev, nslots_dev, theta_dev, hist_dev;
    std::vector<i64> skip_index;         // per term: first entry of its skip table, or -1 (short lists)
    // Host staging of the query plans: a ring of pinned buffers, each with the event of its last copy.  A call fills the next
    // buffer and enqueues its copies without waiting for anything but THAT buffer's previous copy (kStages calls ago), so a
    // host that pipelines batches (ShardedHybrid: the BM25 leg of step i beside the scan of step i + 1) is not held until
    // the previous call's kernels have run -- with one pageable staging vector every call blocked on its predecessor.
    static constexpr int kStages = 4;
    struct Stage {
        PinBuf slots, nslots;
        hipEvent_t ev = nullptr;
        bool used = false;
    };
    Stage stages[kStages];
    unsigned stage_next = 0;
    bool force_global = false;           // HIPBM25_GLOBAL_ACC=1: the global-accumulator form for every k (A/B runs)
    int ws_k = 0;
    i64 queries = 0, postings_touched = 0, bytes_alg = 0;

    i64 ntiles() const { return std::max<i64>(1, (n_docs + kTileDocs - 1) / kTileDocs); }
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

    // tiled path: the whole query batch in one TAAT launch + one merge launch
    int32_t search_tiled(const uint32_t* terms, const int32_t* qoff, int nq, int k, double* o64p, float* o32p, i64* oidp,
                         hipStream_t st)
    {
        int32_t rc;
        int max_slots = 1;
        for (int b = 0; b < nq; ++b) max_slots = std::max(max_slots, qoff[b + 1] - qoff[b]);
        Stage& sg = stages[stage_next++ % kStages];
        if (sg.used) HR_CHECK_HIP(hipEventSynchronize(sg.ev));   // this buffer's previous copy (kStages calls ago) has left it
        const size_t n_slots = (size_t)nq * max_slots;
        if ((rc = sg.slots.reserve(n_slots * sizeof(TileSlot)))) return rc;
        if ((rc = sg.nslots.reserve((size_t)nq * sizeof(int)))) return rc;
        TileSlot* plan_slots = sg.slots.as<TileSlot>();
        int* plan_nslots = sg.nslots.as<int>();
        for (size_t i = 0; i < n_slots; ++i) plan_slots[i] = TileSlot{0, 0, -1};
        for (int b = 0; b < nq; ++b) {
            const int nt = qoff[b + 1] - qoff[b];
            plan_nslots[b] = nt;
            for (int s = 0; s < nt; ++s) {
                const uint32_t t = terms[qoff[b] + s];
                TileSlot& sl = plan_slots[(size_t)b * max_slots + s];
                if ((i64)t < n_terms) { sl.lo = offsets[t]; sl.hi = offsets[t + 1]; sl.skip = skip_index[t]; }  // unknown terms score nothing
                postings_touched += (i64)(sl.hi - sl.lo);
                bytes_alg += (i64)(sl.hi - sl.lo) * 8;
            }
        }
        const i64 lists = ntiles() * kTileWaves;
        if ((rc = slots_dev.reserve(n_slots * sizeof(TileSlot)))) return rc;
        if ((rc = nslots_dev.reserve((size_t)nq * sizeof(int)))) return rc;
        if ((rc = ck.reserve((size_t)nq * lists * k * sizeof(u64)))) return rc;
        if ((rc = ci.reserve((size_t)nq * lists * k * sizeof(i64)))) return rc;
        HR_CHECK_HIP(hipMemcpyAsync(slots_dev.p, plan_slots, n_slots * sizeof(TileSlot), hipMemcpyHostToDevice, st));
        HR_CHECK_HIP(hipMemcpyAsync(nslots_dev.p, plan_nslots, (size_t)nq * sizeof(int), hipMemcpyHostToDevice, st));
        if (!sg.ev) HR_CHECK_HIP(hipEventCreateWithFlags(&sg.ev, hipEventDisableTiming));
        HR_CHECK_HIP(hipEventRecord(sg.ev, st));
        sg.used = true;
        if ((rc = theta_dev.reserve((size_t)nq * sizeof(u32)))) return rc;
        HR_CHECK_HIP(hipMemsetAsync(theta_dev.p, 0, (size_t)nq * sizeof(u32), st));
        u32* hist_p = nullptr;
        {
            const size_t hbytes = (size_t)nq * (kHistBuckets + 1) * sizeof(u32);
            if ((rc = hist_dev.reserve(hbytes))) return rc;
            HR_CHECK_HIP(hipMemsetAsync(hist_dev.p, 0, hbytes, st));
            hist_p = hist_dev.as<u32>();
        }
        static bool lds_ok = false;
        if (!lds_ok) {
            HR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(taat_tile_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             kTileDocs * (int)sizeof(float)));
            HR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(taat_tile_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             kTileDocs * (int)sizeof(float)));
            lds_ok = true;
        }
        auto tile_kernel = n_postings < ((i64)1 << 30) ? taat_tile_kernel<true> : taat_tile_kernel<false>;
        hipLaunchKernelGGL(tile_kernel, dim3(nq, (unsigned)ntiles()), dim3(kTileThreads), kTileDocs * sizeof(float), st,
                           doc_ids.as<u32>(), impacts.as<float>(), slots_dev.as<TileSlot>(), nslots_dev.as<int>(), skip_dev.as<u32>(),
                           max_slots, n_docs, k, ck.as<u64>(), ci.as<i64>(), theta_dev.as<u32>(), hist_p);
        const i64 wave_cand = lists * k;
        const i64 per_lane = (wave_cand + 1023) / 1024;
        auto mk = merge_packed_loop_kernel<8>;   // any size
        if (per_lane <= 16) mk = merge_packed_kernel<16>;
        if (per_lane <= 1) mk = merge_packed_kernel<1>;
        else if (per_lane <= 2) mk = merge_packed_kernel<2>;
        else if (per_lane <= 4) mk = merge_packed_kernel<4>;
        else if (per_lane <= 8) mk = merge_packed_kernel<8>;
        hipLaunchKernelGGL(mk, dim3(nq), dim3(1024), 0, st, (const u64*)ck.as<u64>(), (const i64*)ci.as<i64>(), wave_cand, k, id_base,
                           o64p, o32p, oidp);
        HR_CHECK_HIP(hipGetLastError());
        queries += nq;
        bytes_alg += (i64)nq * n_docs * 8;   // SURVEY 8d counts the accumulator zero + scan passes; this path keeps them in LDS
        return HIPRAG_OK;
    }

    // One workspace (ck / ci / theta / slots / acc) serves every call on this handle: a call on another stream than the
    // previous one is ordered behind it on the device, so that two streams never share the workspace in time.
    hipEvent_t prev_ev = nullptr;
    bool prev_ev_set = false;
    hipStream_t prev_stream = nullptr;

    int32_t search_dev(const uint32_t* terms, const int32_t* qoff, int nq, int k, double* o64p, float* o32p, i64* oidp,
                       hipStream_t st)
    {
        if (prev_ev_set && prev_stream != st) HR_CHECK_HIP(hipStreamWaitEvent(st, prev_ev, 0));
        const int32_t rc = search_dev_impl(terms, qoff, nq, k, o64p, o32p, oidp, st);
        if (rc) return rc;
        if (!prev_ev) HR_CHECK_HIP(hipEventCreateWithFlags(&prev_ev, hipEventDisableTiming));
        HR_CHECK_HIP(hipEventRecord(prev_ev, st));
        prev_ev_set = true;
        prev_stream = st;
        return HIPRAG_OK;
    }

    ~Bm25Index()
    {
        if (prev_ev) (void)hipEventDestroy(prev_ev);
        for (Stage& sg : stages)
            if (sg.ev) (void)hipEventDestroy(sg.ev);
    }

    int32_t search_dev_impl(const uint32_t* terms, const int32_t* qoff, int nq, int k, double* o64p, float* o32p, i64* oidp,
                            hipStream_t st)
    {
        int32_t rc;
        int longest = 0;
        for (int b = 0; b < nq; ++b) longest = std::max(longest, qoff[b + 1] - qoff[b]);
        // the tiled form for any collection whose per-query candidate lists (4 per tile x k) stay below a million entries --
        // 20 M documents at k = 50; the merge walks lists of any length (merge_packed_loop_kernel).  (Until late round 3 the
        // limit was 65536 entries = 3 M documents at k = 50: a 10M-document shard on one GPU fell back to the
        // global-accumulator form, 200 ms per 256 queries.)
        if (!force_global && k <= 64 && longest <= kMaxSlots && ntiles() * kTileWaves * k <= (i64)1 << 20)
            return search_tiled(terms, qoff, nq, k, o64p, o32p, oidp, st);
        if ((rc = reserve(k))) return rc;
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

size_t clear_bm25_registry() { return reg().clear(); }
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
    // every list strictly ascending by document: one posting per (term, document) is what makes the atomics-free
    // accumulation deterministic, and the skip tables are built by binary search
    for (int64_t t = 0; t < n_terms; ++t)
        for (uint64_t i = offsets_host[t] + 1; i < offsets_host[t + 1]; ++i)
            HR_REQUIRE(doc_ids_host[i - 1] < doc_ids_host[i], "posting list of term %lld is not strictly ascending by doc id", (long long)t);
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
    // skip tables of the long lists (taat_tile_kernel): posting offset at every tile boundary, relative to the list start
    {
        const i64 nt = ix->ntiles();
        ix->skip_index.assign((size_t)n_terms, -1);
        std::vector<u32> skip;
        for (int64_t t = 0; t < n_terms; ++t) {
            const uint64_t lo = offsets_host[t], hi = offsets_host[t + 1];
            if (hi - lo < kSkipMinDf) continue;
            HR_REQUIRE(hi - lo < (1ull << 32), "posting list of term %lld is too long for u32 skip offsets", (long long)t);
            ix->skip_index[(size_t)t] = (i64)skip.size();
            const uint32_t* b = doc_ids_host + lo;
            const uint32_t* e = doc_ids_host + hi;
            for (i64 tile = 0; tile <= nt; ++tile) {
                const uint64_t bound = (uint64_t)tile * kTileDocs;
                skip.push_back((u32)(std::lower_bound(b, e, bound, [](uint32_t d, uint64_t v) { return (uint64_t)d < v; }) - b));
            }
        }
        if ((rc = ix->skip_dev.reserve(std::max<size_t>(16, skip.size() * sizeof(u32))))) return rc;
        if (!skip.empty()) HR_CHECK_HIP(hipMemcpy(ix->skip_dev.p, skip.data(), skip.size() * sizeof(u32), hipMemcpyHostToDevice));
        const char* fg = getenv("HIPBM25_GLOBAL_ACC");
        ix->force_global = fg && fg[0] == '1';
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
