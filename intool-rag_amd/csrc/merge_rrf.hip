// merge_rrf.hip -- (1) merge of per-shard partial top-k lists after the all-gather, (2) reciprocal-rank fusion.
//
// Neither exists in the reference (single CPU process, dense leg only): the merge is the exchange step of the
// row-sharded index, RRF is the fusion the reference names in README.md:54-58 with weights hinted
// by rag/config.py:44-45.  Specs restated in oracle/hybrid_oracle.py (merge_partial_topk, rrf_fuse).
// Both are latency-bound: a handful of KiB per query, one workgroup per query.
#include <cfloat>

#include "common.h"
#include "topk_device.h"

namespace hiprag {
namespace {

constexpr int kThreads = 256;

template <int METRIC>
__global__ __launch_bounds__(kThreads) void merge_topk_kernel(const double* __restrict__ in_s, const i64* __restrict__ in_i,
                                                             int n_parts, int nq, int k_in, int k_out, i64 part_stride,
                                                             double* __restrict__ out64, float* __restrict__ out32,
                                                             i64* __restrict__ out_ids)
{
    extern __shared__ unsigned char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    i64* ids = reinterpret_cast<i64*>(keys + kTile);
    u64* selk = reinterpret_cast<u64*>(ids + kTile);
    i64* seli = reinterpret_cast<i64*>(selk + k_out);
    KeyId* red = reinterpret_cast<KeyId*>(seli + k_out);
    const int q = blockIdx.x;
    const i64 M = (i64)n_parts * k_in;
    auto load = [&](i64 i, u64& k, i64& id) {
        const int part = (int)(i / k_in), j = (int)(i % k_in);
        const i64 o = (i64)part * part_stride + (i64)q * k_in + j;
        id = in_i[o];
        const double s = in_s[o];
        k = id < 0 ? 0ull : ord64(METRIC == HIPRAG_METRIC_IP ? s : -s);
    };
    wg_stream_topk<kThreads>(load, M, k_out, keys, ids, red, selk, seli);
    for (int r = threadIdx.x; r < k_out; r += kThreads) {
        const i64 o = (i64)q * k_out + r;
        const u64 k = selk[r];
        if (k == 0) {
            out64[o] = METRIC == HIPRAG_METRIC_IP ? -DBL_MAX : DBL_MAX;
            if (out32) out32[o] = METRIC == HIPRAG_METRIC_IP ? -FLT_MAX : FLT_MAX;
            out_ids[o] = -1;
        } else {
            const double s = METRIC == HIPRAG_METRIC_IP ? unord64(k) : -unord64(k);
            out64[o] = s;
            if (out32) out32[o] = (float)s;
            out_ids[o] = seli[r];
        }
    }
}

// Fast form for the usual exchange sizes (n_parts * k_in <= 64 * NPL, k_out <= 64): one wave per query, no LDS, so
// it runs beside a resident scan workgroup.  Lane-best keys give a starting threshold; then filter-and-insert.
template <int METRIC, int NPL>
__global__ __launch_bounds__(64) void merge_wave_kernel(const double* __restrict__ in_s, const i64* __restrict__ in_i,
                                                       int n_parts, int nq, int k_in, int k_out, i64 part_stride,
                                                       double* __restrict__ out64, float* __restrict__ out32,
                                                       i64* __restrict__ out_ids)
{
    const int q = blockIdx.x, lane = threadIdx.x;
    const int M = n_parts * k_in;
    u64 ck[NPL];
    i64 ci[NPL];
    u64 m = 0;
    // unconditional loads with a clamped index, then the masking: behind `if (i < M)` each candidate is a branch + two
    // loads + s_waitcnt vmcnt(0), one memory round trip per candidate on the tail of every multi-GPU step
    double cs[NPL];
#pragma unroll
    for (int n = 0; n < NPL; ++n) {
        const int i = min(n * 64 + lane, M - 1);
        const int part = i / k_in, j = i % k_in;
        const i64 o = (i64)part * part_stride + (i64)q * k_in + j;
        ci[n] = in_i[o];
        cs[n] = in_s[o];
    }
#pragma unroll
    for (int n = 0; n < NPL; ++n) {
        const bool ok = n * 64 + lane < M && ci[n] >= 0;
        ck[n] = ok ? ord64(METRIC == HIPRAG_METRIC_IP ? cs[n] : -cs[n]) : 0ull;
        if (!ok) ci[n] = -1;
        m = ck[n] > m ? ck[n] : m;
    }
    const u64 t0 = wave_kth_of_lanes(m, k_out);
    WaveListPair F;
    F.init();
#pragma unroll
    for (int n = 0; n < NPL; ++n) F.offer(ck[n] >= t0 ? ck[n] : 0ull, ci[n], k_out);
    if (lane < k_out) {
        const i64 o = (i64)q * k_out + lane;
        if (F.k == 0) {
            out64[o] = METRIC == HIPRAG_METRIC_IP ? -DBL_MAX : DBL_MAX;
            if (out32) out32[o] = METRIC == HIPRAG_METRIC_IP ? -FLT_MAX : FLT_MAX;
            out_ids[o] = -1;
        } else {
            const double s = METRIC == HIPRAG_METRIC_IP ? unord64(F.k) : -unord64(F.k);
            out64[o] = s;
            if (out32) out32[o] = (float)s;
            out_ids[o] = F.id;
        }
    }
}

// One wave per query.  LDS: the two input lists, then the fused (key,id) union.
__global__ __launch_bounds__(64) void rrf_kernel(const i64* __restrict__ ids_a, const i64* __restrict__ ids_b, int depth_a,
                                                int depth_b, int k, float c, float w_a, float w_b,
                                                float* __restrict__ out_s, i64* __restrict__ out_i)
{
    extern __shared__ unsigned char smem[];
    const int n = depth_a + depth_b;
    i64* la = reinterpret_cast<i64*>(smem);
    i64* lb = la + depth_a;
    u64* keys = reinterpret_cast<u64*>(lb + depth_b);
    i64* ids = reinterpret_cast<i64*>(keys + n);
    KeyId* red = reinterpret_cast<KeyId*>(ids + n);
    const int q = blockIdx.x, lane = threadIdx.x;
    for (int i = lane; i < depth_a; i += 64) la[i] = ids_a[(i64)q * depth_a + i];
    for (int i = lane; i < depth_b; i += 64) lb[i] = ids_b[(i64)q * depth_b + i];
    __syncthreads();
    // entries of list a: rank = first occurrence; partner rank looked up in b
    for (int i = lane; i < depth_a; i += 64) {
        const i64 d = la[i];
        u64 key = 0;
        if (d >= 0) {
            bool first = true;
            for (int j = 0; j < i; ++j) first = first && (la[j] != d);
            if (first) {
                int rb = 0;
                for (int j = depth_b - 1; j >= 0; --j) if (lb[j] == d) rb = j + 1;  // first occurrence in b
                const float ta = w_a / (c + (float)(i + 1));
                const float tb = rb ? w_b / (c + (float)rb) : 0.0f;
                key = (u64)ord32(ta + tb) << 32;
            }
        }
        keys[i] = key;
        ids[i] = d;
    }
    // entries only in list b
    for (int i = lane; i < depth_b; i += 64) {
        const i64 d = lb[i];
        u64 key = 0;
        if (d >= 0) {
            bool fresh = true;
            for (int j = 0; j < i; ++j) fresh = fresh && (lb[j] != d);
            for (int j = 0; j < depth_a; ++j) fresh = fresh && (la[j] != d);
            if (fresh) {
                const float ta = 0.0f;
                const float tb = w_b / (c + (float)(i + 1));
                key = (u64)ord32(ta + tb) << 32;
            }
        }
        keys[depth_a + i] = key;
        ids[depth_a + i] = d;
    }
    __syncthreads();
    wg_topk_rounds<64>(keys, ids, n, k, red, [&](int r, u64 kk, i64 id) {
        const i64 o = (i64)q * k + r;
        out_s[o] = kk ? unord32((u32)(kk >> 32)) : -FLT_MAX;
        out_i[o] = kk ? id : -1;
    });
}

}  // namespace
}  // namespace hiprag

using namespace hiprag;

extern "C" {

int32_t hiprag_merge_topk_dev(const double* in_scores64_dev, const int64_t* in_ids_dev, int32_t n_parts, int32_t nq,
                              int32_t k_in, int32_t k_out, int64_t part_stride, int32_t metric,
                              double* out_scores64_dev, float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    if (part_stride == 0) part_stride = (int64_t)nq * k_in;
    HR_REQUIRE(part_stride >= (int64_t)nq * k_in, "part_stride smaller than one part");
    HR_REQUIRE(n_parts > 0 && nq >= 0 && k_in > 0 && k_out > 0 && k_out < kTile, "bad merge shape");
    HR_REQUIRE(metric == HIPRAG_METRIC_IP || metric == HIPRAG_METRIC_L2, "unknown metric %d", metric);
    if (nq == 0) return HIPRAG_OK;
    HR_REQUIRE(in_scores64_dev && in_ids_dev && out_scores64_dev && out_ids_dev, "null device pointer");
    const int M = n_parts * k_in;
    if (k_out <= 64 && M <= 64 * 8) {
        const int npl = (M + 63) / 64;
        const bool ip = metric == HIPRAG_METRIC_IP;
        void (*kern)(const double*, const i64*, int, int, int, int, i64, double*, float*, i64*) =
            npl <= 1 ? (ip ? merge_wave_kernel<0, 1> : merge_wave_kernel<1, 1>)
          : npl <= 2 ? (ip ? merge_wave_kernel<0, 2> : merge_wave_kernel<1, 2>)
          : npl <= 4 ? (ip ? merge_wave_kernel<0, 4> : merge_wave_kernel<1, 4>)
                     : (ip ? merge_wave_kernel<0, 8> : merge_wave_kernel<1, 8>);
        hipLaunchKernelGGL(kern, dim3(nq), dim3(64), 0, (hipStream_t)stream, in_scores64_dev, (const i64*)in_ids_dev,
                           n_parts, nq, k_in, k_out, (i64)part_stride, out_scores64_dev, out_scores_dev, (i64*)out_ids_dev);
        HR_CHECK_HIP(hipGetLastError());
        return HIPRAG_OK;
    }
    const size_t lds = (size_t)kTile * 16 + (size_t)k_out * 16 + 2 * (kThreads / 64) * sizeof(KeyId);
    auto kern = metric == HIPRAG_METRIC_IP ? merge_topk_kernel<HIPRAG_METRIC_IP> : merge_topk_kernel<HIPRAG_METRIC_L2>;
    HR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(nq), dim3(kThreads), lds, (hipStream_t)stream, in_scores64_dev,
                       (const i64*)in_ids_dev, n_parts, nq, k_in, k_out, (i64)part_stride, out_scores64_dev, out_scores_dev,
                       (i64*)out_ids_dev);
    HR_CHECK_HIP(hipGetLastError());
    return HIPRAG_OK;
}

int32_t hiprrf_fuse_dev(const int64_t* ids_a_dev, const int64_t* ids_b_dev, int32_t nq, int32_t depth_a, int32_t depth_b,
                        int32_t k, float c, float w_a, float w_b, float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    HR_REQUIRE(nq >= 0 && depth_a >= 0 && depth_b >= 0 && k > 0, "bad rrf shape");
    HR_REQUIRE(depth_a + depth_b <= 4096, "rrf depth_a + depth_b must be <= 4096");
    HR_REQUIRE(c + 1.0f > 0.0f, "rrf constant c must keep c + rank positive");
    if (nq == 0) return HIPRAG_OK;
    HR_REQUIRE((ids_a_dev || depth_a == 0) && (ids_b_dev || depth_b == 0) && out_scores_dev && out_ids_dev,
               "null device pointer");
    const int n = depth_a + depth_b;
    const size_t lds = (size_t)n * 8 + (size_t)n * 16 + 2 * sizeof(KeyId) + 64;
    HR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(rrf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds));
    hipLaunchKernelGGL(rrf_kernel, dim3(nq), dim3(64), lds, (hipStream_t)stream, (const i64*)ids_a_dev,
                       (const i64*)ids_b_dev, depth_a, depth_b, k, c, w_a, w_b, out_scores_dev, (i64*)out_ids_dev);
    HR_CHECK_HIP(hipGetLastError());
    return HIPRAG_OK;
}

int32_t hiprrf_fuse(const int64_t* ids_a_host, const int64_t* ids_b_host, int32_t nq, int32_t depth_a, int32_t depth_b,
                    int32_t k, float c, float w_a, float w_b, float* out_scores, int64_t* out_ids)
{
    HR_REQUIRE(nq >= 0 && depth_a >= 0 && depth_b >= 0 && k > 0, "bad rrf shape");
    if (nq == 0) return HIPRAG_OK;
    DevBuf a, b, os, oi;
    int32_t rc;
    if ((rc = a.reserve(std::max<size_t>(8, (size_t)nq * depth_a * 8)))) return rc;
    if ((rc = b.reserve(std::max<size_t>(8, (size_t)nq * depth_b * 8)))) return rc;
    if ((rc = os.reserve((size_t)nq * k * 4))) return rc;
    if ((rc = oi.reserve((size_t)nq * k * 8))) return rc;
    if (depth_a) HR_CHECK_HIP(hipMemcpy(a.p, ids_a_host, (size_t)nq * depth_a * 8, hipMemcpyHostToDevice));
    if (depth_b) HR_CHECK_HIP(hipMemcpy(b.p, ids_b_host, (size_t)nq * depth_b * 8, hipMemcpyHostToDevice));
    rc = hiprrf_fuse_dev(a.as<int64_t>(), b.as<int64_t>(), nq, depth_a, depth_b, k, c, w_a, w_b, os.as<float>(),
                         oi.as<int64_t>(), nullptr);
    if (rc) return rc;
    HR_CHECK_HIP(hipMemcpy(out_scores, os.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    HR_CHECK_HIP(hipMemcpy(out_ids, oi.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
    return HIPRAG_OK;
}

}  // extern "C"
