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
template <int NT, typename Load>
__device__ __forceinline__ void wg_stream_topk(Load load, i64 M, int K, u64* keys, i64* ids, KeyId* red, u64* selk,
                                               i64* seli)
{
    const int tid = threadIdx.x;
    int filled = 0;
    i64 o = 0;
    for (;;) {
        i64 rem = M - o;
        const int take = (int)(rem < (i64)(kTile - filled) ? rem : (i64)(kTile - filled));
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

}  // namespace hiprag
