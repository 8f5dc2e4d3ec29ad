// topk_device.h -- workgroup-level exact top-K selection used by every select/merge kernel.
//
// Canonical order everywhere in libhiprag: larger key first, then LOWER id first.  Keys are order-preserving
// unsigned images of the scores (ord32 / ord64 below), key 0 means "empty slot" (it is below the image of every
// non-NaN value).  Selection runs in rounds: each thread keeps the best of the LDS entries it owns (a strided
// slice), one wave-wide xor-shuffle reduction and one barrier per round pick the winner, the owning thread
// clears that entry and rescans only its own slice.  Cost per round is a few hundred cycles; K is tens.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace hiprag {

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;

struct KeyId {
    u64 key;
    i64 id;
    int pos;
};

__device__ __forceinline__ u32 ord32(float f)
{
    u32 b = __float_as_uint(f);
    return b ^ ((b & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unord32(u32 o)
{
    u32 b = o ^ ((o & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(b);
}
__device__ __forceinline__ u64 ord64(double d)
{
    u64 b = (u64)__double_as_longlong(d);
    return b ^ ((b >> 63) ? ~0ull : (1ull << 63));
}
__device__ __forceinline__ double unord64(u64 o)
{
    u64 b = o ^ ((o >> 63) ? (1ull << 63) : ~0ull);
    return __longlong_as_double((i64)b);
}

__device__ __forceinline__ bool key_before(u64 ka, i64 ia, u64 kb, i64 ib)
{
    return ka > kb || (ka == kb && ia < ib);
}

__device__ __forceinline__ KeyId wave_best(KeyId v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        KeyId o;
        o.key = __shfl_xor(v.key, off);
        o.id = __shfl_xor(v.id, off);
        o.pos = __shfl_xor(v.pos, off);
        if (key_before(o.key, o.id, v.key, v.id)) v = o;
    }
    return v;
}

// Workgroup top-K over LDS arrays keys[n] / ids[n] (n entries, key 0 = empty).  DESTROYS keys (winners are
// cleared).  `red` is LDS scratch of 2*(NT/64) KeyId.  Calls emit(rank, key, id) on thread 0 for rank 0..K-1
// in canonical order; exhausted ranks are emitted with key 0, id -1.  All threads of the workgroup must call
// it; keys/ids must be visible (barrier) before the call.  Ends with a barrier.
template <int NT, typename Emit>
__device__ __forceinline__ void wg_topk_rounds(u64* keys, const i64* ids, int n, int K, KeyId* red, Emit emit)
{
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    auto rescan = [&]() {
        KeyId best;
        best.key = 0;
        best.id = 0x7FFFFFFFFFFFFFFFll;
        best.pos = -1;
        for (int pos = tid; pos < n; pos += NT) {
            u64 kk = keys[pos];
            if (kk != 0) {
                i64 ii = ids[pos];
                if (key_before(kk, ii, best.key, best.id)) { best.key = kk; best.id = ii; best.pos = pos; }
            }
        }
        return best;
    };

    KeyId local = rescan();
    for (int r = 0; r < K; ++r) {
        KeyId wb = wave_best(local);
        if (lane == 0) red[(r & 1) * NW + wave] = wb;
        __syncthreads();
        KeyId g = red[(r & 1) * NW];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            KeyId o = red[(r & 1) * NW + w];
            if (key_before(o.key, o.id, g.key, g.id)) g = o;
        }
        if (g.key == 0) {  // exhausted (uniform across the workgroup)
            if (tid == 0)
                for (int rr = r; rr < K; ++rr) emit(rr, (u64)0, (i64)-1);
            break;
        }
        if (tid == 0) emit(r, g.key, g.id);
        if (g.pos >= 0 && (g.pos % NT) == tid) {
            keys[g.pos] = 0;
            local = rescan();
        }
    }
    __syncthreads();
}


constexpr int kTile = 4096;  // LDS tile entries of the streaming selectors (64 KiB of key+id)

// Streaming selection: exact top-K of M candidates produced by load(idx, &key, &id), using one kTile-entry LDS tile
// (keys/ids) and carrying the running winners between tiles.  Winners land in selk/seli (LDS, K entries, canonical
// order, exhausted ranks key 0 / id -1).  K must be < kTile.
template <int NT, int TILE = kTile, typename Load>
__device__ __forceinline__ void wg_stream_topk(Load load, i64 M, int K, u64* keys, i64* ids, KeyId* red, u64* selk,
                                               i64* seli)
{
    const int tid = threadIdx.x;
    int filled = 0;
    i64 o = 0;
    for (;;) {
        i64 rem = M - o;
        const int take = (int)(rem < (i64)(TILE - filled) ? rem : (i64)(TILE - filled));
        for (int i = tid; i < take; i += NT) {
            u64 k;
            i64 id;
            load(o + i, k, id);
            keys[filled + i] = k;
            ids[filled + i] = id;
        }
        const int n = filled + take;
        o += take;
        __syncthreads();
        wg_topk_rounds<NT>(keys, ids, n, K, red, [&](int r, u64 k, i64 id) { selk[r] = k; seli[r] = id; });
        if (o >= M) break;
        for (int i = tid; i < K; i += NT) { keys[i] = selk[i]; ids[i] = seli[i]; }
        filled = K;
        __syncthreads();
    }
}

// Level-1 selector over a dense float array per query: grid (nchunks, nq), 256 threads.  Emits the top-K1 of each
// kTile-entry chunk as (key = ord32(v) << 32, id = position) to ck/ci[q][chunk][K1].  POSITIVE_ONLY drops v <= 0.
template <bool POSITIVE_ONLY>
__global__ __launch_bounds__(256) void select_f32_kernel(const float* __restrict__ vals, i64 stride, i64 n_total, int K1,
                                                        u64* __restrict__ ck, i64* __restrict__ ci)
{
    __shared__ u64 keys[kTile];
    __shared__ i64 ids[kTile];
    __shared__ KeyId red[2 * 4];
    const int q = blockIdx.y;
    const i64 base = (i64)blockIdx.x * kTile;
    i64 rem = n_total - base;
    const int n = (int)(rem < (i64)kTile ? (rem < 0 ? 0 : rem) : (i64)kTile);
    const float* src = vals + (i64)q * stride + base;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = src[i];
        keys[i] = (POSITIVE_ONLY && !(v > 0.f)) ? 0ull : ((u64)ord32(v) << 32);
        ids[i] = base + i;
    }
    __syncthreads();
    u64* ok = ck + ((i64)q * gridDim.x + blockIdx.x) * K1;
    i64* oi = ci + ((i64)q * gridDim.x + blockIdx.x) * K1;
    wg_topk_rounds<256>(keys, ids, n, K1, red, [&](int r, u64 k, i64 id) { ok[r] = k; oi[r] = id; });
}

// ------------------------------------------------------------------------------------------------------
// Wave-resident sorted list ("filter and insert"): exact top-K for K <= 64 with one compare per rejected element.
// Lane j of the wave holds the j-th best entry seen so far; a candidate that beats the current K-th entry is
// inserted by one wave-wide compare + a shift by one lane.  After warm-up almost every element of a stream fails
// the threshold test, so a wave filters a stream at close to load speed.
//
// Two entry flavours share the code:
//   packed  : u64 = (ord32(value) << 32) | (0xFFFFFFFF - index32): one integer compare orders (value desc, index asc)
//   pair    : (u64 key, i64 id) with the canonical comparator (fp64 scores, 64-bit ids)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 pack_key(float v, u32 idx) { return ((u64)ord32(v) << 32) | (u64)(0xFFFFFFFFu - idx); }
__device__ __forceinline__ u32 packed_index(u64 k) { return 0xFFFFFFFFu - (u32)(k & 0xFFFFFFFFull); }
__device__ __forceinline__ float packed_value(u64 k) { return unord32((u32)(k >> 32)); }

__device__ __forceinline__ u64 readlane_u64(u64 v, int lane)
{
    const u32 lo = __builtin_amdgcn_readlane((int)(u32)v, lane);
    const u32 hi = __builtin_amdgcn_readlane((int)(u32)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}

// lane j receives lane j-1's value (lane 0 keeps its own): DPP wave_shr:1, two VALU moves instead of LDS-crossbar
// shuffles.
__device__ __forceinline__ u32 lane_up1_u32(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ u64 lane_up1_u64(u64 v)
{
    return ((u64)lane_up1_u32((u32)(v >> 32)) << 32) | lane_up1_u32((u32)v);
}

struct WaveListPacked {
    u64 e;  // this lane's entry (0 = empty)
    __device__ __forceinline__ void init() { e = 0; }
    // insert a wave-uniform candidate c (c != 0); keeps lanes sorted descending
    __device__ __forceinline__ void insert(u64 c)
    {
        const int lane = threadIdx.x & 63;
        const u64 up = lane_up1_u64(e);
        const bool before = c > e;
        const bool before_prev = lane > 0 && c > up;
        if (before) e = before_prev ? up : c;
    }
    __device__ __forceinline__ u64 kth(int K) const { return readlane_u64(e, K - 1); }
    // Offer every lane's candidate (0 = nothing) against threshold tau: candidates must be > tau to enter, and tau
    // rises to the K-th entry once the list holds K entries above it.  Returns the new tau.
    __device__ __forceinline__ u64 offer(u64 c, int K, u64 tau)
    {
        unsigned long long mask = __ballot(c > tau);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const u64 cj = readlane_u64(c, j);
            if (cj > tau) {  // wave-uniform
                insert(cj);
                const u64 kk = kth(K);
                tau = kk > tau ? kk : tau;
            }
        }
        return tau;
    }
};

struct WaveListPair {
    u64 k;
    i64 id;
    __device__ __forceinline__ void init() { k = 0; id = 0x7FFFFFFFFFFFFFFFll; }
    __device__ __forceinline__ void insert(u64 ck, i64 cid)
    {
        const int lane = threadIdx.x & 63;
        const u64 upk = lane_up1_u64(k);
        const i64 upi = (i64)lane_up1_u64((u64)id);
        const bool before = key_before(ck, cid, k, id);
        const bool before_prev = lane > 0 && key_before(ck, cid, upk, upi);
        if (before) {
            k = before_prev ? upk : ck;
            id = before_prev ? upi : cid;
        }
    }
    // Offer every lane's candidate (key 0 = nothing); the K-th entry is the threshold.
    __device__ __forceinline__ void offer(u64 ck, i64 cid, int K)
    {
        u64 tk = readlane_u64(k, K - 1);
        i64 ti = (i64)readlane_u64((u64)id, K - 1);
        unsigned long long mask = __ballot(ck != 0 && key_before(ck, cid, tk, ti));
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const u64 cj = readlane_u64(ck, j);
            const i64 ij = (i64)readlane_u64((u64)cid, j);
            if (key_before(cj, ij, tk, ti)) {
                insert(cj, ij);
                tk = readlane_u64(k, K - 1);
                ti = (i64)readlane_u64((u64)id, K - 1);
            }
        }
    }
};

// Largest t among the 64 lane values m (0 = none) such that at least K lanes hold a value >= t; 0 if fewer than K
// lanes are non-empty.  K distinct lanes with value >= t prove t is a lower bound of the K-th largest element of
// whatever population the lane values were drawn from, so (t - 1) is a safe starting threshold for offer().
__device__ __forceinline__ u64 wave_kth_of_lanes(u64 m, int K)
{
    u64 best = 0;
    for (int j = 0; j < 64; ++j) {
        const u64 mj = readlane_u64(m, j);
        const int cnt = __popcll(__ballot(m >= mj));
        if (mj != 0 && cnt >= K && mj > best) best = mj;
    }
    return best;
}

// Exact top-K (K <= 64) of per-lane candidate arrays c[0..N) (packed keys, 0 = none) into a sorted wave list.
template <int N>
__device__ __forceinline__ void wave_topk_packed(const u64 (&c)[N], int K, WaveListPacked& L)
{
    u64 m = 0;
#pragma unroll
    for (int n = 0; n < N; ++n) m = c[n] > m ? c[n] : m;
    const u64 t0 = wave_kth_of_lanes(m, K);
    u64 tau = t0 ? t0 - 1 : 0;
    L.init();
#pragma unroll
    for (int n = 0; n < N; ++n) tau = L.offer(c[n], K, tau);
}

// Tighten a wave-wide selection threshold by bisection in the ordered-integer domain.  On entry at least K of the wave's
// N * 64 values v (excluded ones are -inf) are >= unord32(lo) and fewer than K are >= unord32(hi); on exit lo has moved up
// so that the count is still >= K but close to it.  Why: the lane-maxima threshold of the selectors (K-th largest of 64
// lane maxima) lets through several hundred of 4096 values when K is 33-50, and every one of them costs a serial
// sorted-list insertion (~100 cycles); ten bisection steps cost ~600 cycles each and cut the insertions to about K.
template <int N>
__device__ __forceinline__ u32 wave_bisect_threshold(const float (&v)[N], u32 lo, u32 hi, int K, int steps)
{
    for (int s = 0; s < steps && hi - lo > 1; ++s) {
        const u32 mid = lo + ((hi - lo) >> 1);
        const float t = unord32(mid);
        int c = 0;   // wave-wide count, kept on the scalar unit: one v_cmp per value, s_bcnt1 + s_add beside it
#pragma unroll
        for (int n = 0; n < N; ++n) c += __popcll(__ballot(v[n] >= t));
        if (c >= K) lo = mid;
        else hi = mid;
        if (c >= K && c <= K + (K >> 1)) break;   // close enough: the sorted-list stage takes the rest
    }
    return lo;
}

// Level-1 selector, K1 <= 64: grid (nslices, nq), 256 threads = 4 independent waves; wave w of slice s filters the
// contiguous kSelPerWave elements [(4s+w)*kSelPerWave, ...) of query q's float array (all loads issued up-front) and
// writes its sorted top-K1 as (key = ord32(v) << 32, id = index) to ck/ci[q][(4s+w)*K1 ..].
// POSITIVE_ONLY drops v <= 0.  `stride` and the array base must be 16-byte aligned multiples of 4 floats.
constexpr int kSelPerWave = 4096;
template <bool POSITIVE_ONLY>
__global__ __launch_bounds__(256) void select_wave_kernel(const float* __restrict__ vals, i64 stride, i64 n_total, int K1,
                                                         u64* __restrict__ ck, i64* __restrict__ ci)
{
    constexpr int NV = kSelPerWave / 256;  // float4 loads per lane
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.y;
    const i64 wave = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const i64 nwaves = (i64)gridDim.x * 4;
    const i64 lo = wave * kSelPerWave;
    const float* src = vals + (i64)q * stride;
    float v[NV * 4];
#pragma unroll
    for (int it = 0; it < NV; ++it) {
        const i64 i = lo + it * 256 + lane * 4;
        float4 x = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (i + 3 < n_total) {
            x = *reinterpret_cast<const float4*>(src + i);
        } else {
            if (i + 0 < n_total) x.x = src[i + 0];
            if (i + 1 < n_total) x.y = src[i + 1];
            if (i + 2 < n_total) x.z = src[i + 2];
        }
        v[it * 4 + 0] = x.x; v[it * 4 + 1] = x.y; v[it * 4 + 2] = x.z; v[it * 4 + 3] = x.w;
    }
    // float-domain pre-threshold: K1-th largest of the 64 lane maxima (padding / excluded values are -inf)
    float m = -INFINITY;
#pragma unroll
    for (int n = 0; n < NV * 4; ++n) {
        const float x = (POSITIVE_ONLY && !(v[n] > 0.f)) ? -INFINITY : v[n];
        m = fmaxf(m, x);
    }
    const u32 mo = m == -INFINITY ? 0u : ord32(m);
    u32 best = 0;
    for (int j = 0; j < 64; ++j) {
        const u32 mj = (u32)__builtin_amdgcn_readlane((int)mo, j);
        const int cnt = __popcll(__ballot(mo >= mj));
        if (mj != 0 && cnt >= K1 && mj > best) best = mj;
    }
    if (best) {
        if (POSITIVE_ONLY) {
#pragma unroll
            for (int n = 0; n < NV * 4; ++n) v[n] = v[n] > 0.f ? v[n] : -INFINITY;
        }
        u32 top = mo;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) top = max(top, (u32)__shfl_xor((int)top, off));
        best = wave_bisect_threshold(v, best, top + 1, K1, 10);
    }
    // elements with ord32(v) >= best can still be among the top K1; everything else is rejected by one compare
    WaveListPacked L;
    L.init();
    u64 tau = best ? ((u64)best << 32) - 1 : 0;
#pragma unroll
    for (int n = 0; n < NV * 4; ++n) {
        const i64 i = lo + (n >> 2) * 256 + lane * 4 + (n & 3);
        u64 c = 0;
        if (v[n] != -INFINITY && (!POSITIVE_ONLY || v[n] > 0.f)) c = pack_key(v[n], (u32)i);
        tau = L.offer(c, K1, tau);
    }
    if (lane < K1) {
        const i64 o = ((i64)q * nwaves + wave) * K1 + lane;
        ck[o] = L.e & 0xFFFFFFFF00000000ull;
        ci[o] = L.e ? (i64)packed_index(L.e) : -1;
    }
}

// Final merge of select_wave_kernel's per-wave lists for value/index problems (BM25): one 16-wave workgroup per query
// reduces ncand = nlists*K1 (key, id) candidates to the top-k; NPL = candidates per lane in the first level.
// Writes score (fp32, also as fp64 if out64) and id + id_base; exhausted ranks: -FLT_MAX / -1.
template <int NPL>
__global__ __launch_bounds__(1024) void merge_packed_kernel(const u64* __restrict__ ck, const i64* __restrict__ ci, i64 ncand,
                                                           int k, i64 id_base, double* __restrict__ out64,
                                                           float* __restrict__ out32, i64* __restrict__ out_ids)
{
    __shared__ u64 lists[16 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x;
    const u64* sk = ck + (i64)q * ncand;
    const i64* si = ci + (i64)q * ncand;
    {
        // unconditional loads of keys and ids (clamped index), combined afterwards: behind `if (i < ncand)` / `if (kk)`
        // every candidate cost two dependent, serialized memory round trips (32 of them at NPL = 16)
        u64 c[NPL];
        i64 cs[NPL];
#pragma unroll
        for (int n = 0; n < NPL; ++n) {
            const i64 i = min(((i64)wave * NPL + n) * 64 + lane, ncand - 1);
            c[n] = sk[i];
            cs[n] = si[i];
        }
#pragma unroll
        for (int n = 0; n < NPL; ++n) {
            const i64 i = ((i64)wave * NPL + n) * 64 + lane;
            c[n] = (i < ncand && c[n] != 0) ? (c[n] | (u64)(0xFFFFFFFFu - (u32)cs[n])) : 0;
        }
        WaveListPacked L;
        wave_topk_packed<NPL>(c, k, L);
        lists[wave * 64 + lane] = lane < k ? L.e : 0;
    }
    __syncthreads();
    if (wave == 0) {
        u64 c[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) c[n] = lists[n * 64 + lane];
        WaveListPacked L;
        wave_topk_packed<16>(c, k, L);
        if (lane < k) {
            const i64 o = (i64)q * k + lane;
            const bool ok = L.e != 0;
            const float s = ok ? packed_value(L.e) : -3.402823466e+38f;
            if (out64) out64[o] = ok ? (double)s : -1.7976931348623157e+308;
            if (out32) out32[o] = s;
            out_ids[o] = ok ? (i64)packed_index(L.e) + id_base : -1;
        }
    }
}

// The same merge for any number of candidates: every wave walks its sixteenth of them in rounds of 8 per lane and offers
// them to a running sorted list (most slots of a tiled BM25 search are empty: a bound shared by the tiles of a query
// leaves later tiles a handful of survivors, and an empty slot costs one compare).
template <int ROUND = 8>
__global__ __launch_bounds__(1024) void merge_packed_loop_kernel(const u64* __restrict__ ck, const i64* __restrict__ ci, i64 ncand,
                                                                int k, i64 id_base, double* __restrict__ out64,
                                                                float* __restrict__ out32, i64* __restrict__ out_ids)
{
    __shared__ u64 lists[16 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = blockIdx.x;
    const u64* sk = ck + (i64)q * ncand;
    const i64* si = ci + (i64)q * ncand;
    {
        const i64 per_wave = ((ncand + 15) / 16 + 63) / 64 * 64;
        const i64 lo = (i64)wave * per_wave, hi = min(lo + per_wave, ncand);
        WaveListPacked L;
        L.init();
        u64 tau = 0;
        for (i64 base = lo; base < hi; base += ROUND * 64) {
            u64 c[ROUND];
            i64 cs[ROUND];
#pragma unroll
            for (int n = 0; n < ROUND; ++n) {   // unconditional loads (clamped), masked afterwards
                const i64 i = min(base + n * 64 + lane, ncand - 1);
                c[n] = sk[i];
                cs[n] = si[i];
            }
#pragma unroll
            for (int n = 0; n < ROUND; ++n) {
                const i64 i = base + n * 64 + lane;
                const u64 key = (i < hi && c[n] != 0) ? (c[n] | (u64)(0xFFFFFFFFu - (u32)cs[n])) : 0;
                tau = L.offer(key, k, tau);
            }
        }
        lists[wave * 64 + lane] = lane < k ? L.e : 0;
    }
    __syncthreads();
    if (wave == 0) {
        u64 c[16];
#pragma unroll
        for (int n = 0; n < 16; ++n) c[n] = lists[n * 64 + lane];
        WaveListPacked L;
        wave_topk_packed<16>(c, k, L);
        if (lane < k) {
            const i64 o = (i64)q * k + lane;
            const bool ok = L.e != 0;
            const float s = ok ? packed_value(L.e) : -3.402823466e+38f;
            if (out64) out64[o] = ok ? (double)s : -1.7976931348623157e+308;
            if (out32) out32[o] = s;
            out_ids[o] = ok ? (i64)packed_index(L.e) + id_base : -1;
        }
    }
}

}  // namespace hiprag
