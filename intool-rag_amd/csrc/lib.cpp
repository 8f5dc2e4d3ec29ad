// lib.cpp -- library-level entry points of libhiprag.so: error state, device queries, HIP-event timing.
#include "common.h"

#include <vector>

namespace hiprag {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

static Registry<hipEvent_t>& events()
{
    static Registry<hipEvent_t> r;
    return r;
}

// Helper streams of hiphybrid_search (the BM25 leg runs beside the dense leg): owned by the library, not by the calling
// thread -- a pool per process, entries tagged with their device, handed out under a mutex and destroyed by hiprag_shutdown.
struct SideStream {
    hipStream_t st = nullptr;
    hipEvent_t done = nullptr;
    int dev = -1;
};
class SidePool {
public:
    int32_t acquire(int dev, SideStream& out)
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].dev == dev) { out = free_[i]; free_.erase(free_.begin() + (long)i); return HIPRAG_OK; }
        }
        SideStream s;
        s.dev = dev;
        HR_CHECK_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
        hipError_t e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (e != hipSuccess) { (void)hipStreamDestroy(s.st); HR_CHECK_HIP(e); }
        out = s;
        return HIPRAG_OK;
    }
    void release(const SideStream& s)
    {
        std::lock_guard<std::mutex> g(mu_);
        free_.push_back(s);
    }
    void clear()   // hiprag_shutdown (every device has been synchronised): leases still out are destroyed by their holders' release -> here next time
    {
        std::lock_guard<std::mutex> g(mu_);
        for (const SideStream& s : free_) {
            if (hipSetDevice(s.dev) != hipSuccess) continue;
            (void)hipEventDestroy(s.done);
            (void)hipStreamDestroy(s.st);
        }
        free_.clear();
    }

private:
    std::mutex mu_;
    std::vector<SideStream> free_;
};
static SidePool& side_pool()
{
    static SidePool p;
    return p;
}
// One lease per hiphybrid_search call.  While `busy` the helper stream may still write into buffers the call's frame owns:
// every way out of the frame (error returns included) first waits for it, then hands the stream back.
struct SideLease {
    SideStream s;
    bool held = false, busy = false;
    ~SideLease()
    {
        if (!held) return;
        if (busy) (void)hipStreamSynchronize(s.st);
        side_pool().release(s);
    }
};

struct StepEvents {    // one event per (device, slot): orders a step's tail stream behind its scan
    std::mutex mu;
    std::unordered_map<long long, hipEvent_t> ev;
    int32_t get(int dev, int slot, hipEvent_t& out)
    {
        std::lock_guard<std::mutex> g(mu);
        const long long key = (long long)dev * 64 + slot;
        auto it = ev.find(key);
        if (it == ev.end()) {
            hipEvent_t e = nullptr;
            HR_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            it = ev.emplace(key, e).first;
        }
        out = it->second;
        return HIPRAG_OK;
    }
    void clear()
    {
        std::lock_guard<std::mutex> g(mu);
        for (auto& kv : ev) {
            if (hipSetDevice((int)(kv.first / 64)) == hipSuccess) (void)hipEventDestroy(kv.second);
        }
        ev.clear();
    }
};
static StepEvents& step_events()
{
    static StepEvents s;
    return s;
}

namespace {
// hiprag_probe_read_gbps: what this device streams through a statically partitioned read (the dense scan's access
// pattern without its work): every wave reads its own contiguous range of 64 KiB blocks, 16 KiB in flight per wave,
// non-temporal 16-byte loads.
typedef float probe_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void read_probe_kernel(const probe_f4* __restrict__ x, long long nblocks, int bpw, int passes,
                                                        float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long gw = (long long)blockIdx.x * 8 + wave;
    probe_f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int pass = 0; pass < passes; ++pass) {
        for (int i = 0; i < bpw; ++i) {
            const long long b = gw * bpw + i;
            if (b >= nblocks) break;
            const probe_f4* src = x + b * 4096 + lane;   // a block = 64 KiB = 4096 x 16 B
            for (int p = 0; p < 64; p += 16) {
                probe_f4 r[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) r[u] = __builtin_nontemporal_load(src + (p + u) * 64);
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += r[u];
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;   // never true for a zeroed buffer: keeps the loads
}
}  // namespace

}  // namespace hiprag

using namespace hiprag;

extern "C" {

int32_t hiprag_version(void) { return 100; }

const char* hiprag_last_error(void) { return g_last_error.c_str(); }

int32_t hiprag_device_count(int32_t* out_count)
{
    HR_REQUIRE(out_count, "null out");
    int n = 0;
    HR_CHECK_HIP(hipGetDeviceCount(&n));
    *out_count = n;
    return HIPRAG_OK;
}

int32_t hiprag_init(int32_t n_devices)
{
    int n = 0;
    HR_CHECK_HIP(hipGetDeviceCount(&n));
    const int want = n_devices > 0 ? n_devices : 1;
    HR_REQUIRE(n >= want, "hiprag_init: %d device(s) requested, %d visible", want, n);
    for (int dev = 0; dev < want; ++dev) {   // create the contexts now, not inside the first search
        HR_CHECK_HIP(hipSetDevice(dev));
        HR_CHECK_HIP(hipFree(nullptr));
    }
    HR_CHECK_HIP(hipSetDevice(0));
    return HIPRAG_OK;
}

int32_t hiprag_shutdown(void)
{
    int n = 0;
    HR_CHECK_HIP(hipGetDeviceCount(&n));
    for (int dev = 0; dev < n; ++dev) {
        if (hipSetDevice(dev) == hipSuccess) (void)hipDeviceSynchronize();
    }
    clear_encoder_registry();
    clear_bm25_registry();
    clear_dense_registry();
    events().clear();
    side_pool().clear();
    step_events().clear();
    return HIPRAG_OK;
}

int32_t hiprag_probe_read_gbps(int32_t device, int64_t bytes, int32_t reps, double* out_gbps)
{
    HR_REQUIRE(out_gbps && bytes >= (1 << 20) && reps > 0, "bad probe arguments");
    HR_CHECK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    HR_CHECK_HIP(hipGetDeviceProperties(&prop, device));
    const int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const long long nblocks = bytes / 65536;
    const long long waves = (long long)ncu * 8;
    const int bpw = (int)((nblocks + waves - 1) / waves), passes = 4;
    void* x = nullptr;
    float* out = nullptr;
    HR_CHECK_HIP(hipMalloc(&x, (size_t)nblocks * 65536));
    hipError_t e = hipMalloc(&out, 64);
    if (e != hipSuccess) { (void)hipFree(x); HR_CHECK_HIP(e); }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    e = hipMemset(x, 0, (size_t)nblocks * 65536);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) {
        for (int it = 0; it < 2; ++it)
            hipLaunchKernelGGL(read_probe_kernel, dim3(ncu), dim3(512), 0, 0, (const probe_f4*)x, nblocks, bpw, passes, out);
        (void)hipEventRecord(e0, 0);
        for (int it = 0; it < reps; ++it)
            hipLaunchKernelGGL(read_probe_kernel, dim3(ncu), dim3(512), 0, 0, (const probe_f4*)x, nblocks, bpw, passes, out);
        (void)hipEventRecord(e1, 0);
        e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(out);
    (void)hipFree(x);
    HR_CHECK_HIP(e);
    *out_gbps = (double)nblocks * 65536.0 * passes * reps / ((double)ms * 1e-3) / 1e9;
    return HIPRAG_OK;
}

int32_t hiphybrid_search(uint64_t dense_h, uint64_t bm25_h, const float* q_host, const uint32_t* term_ids_host,
                         const int32_t* q_offsets_host, int32_t nq, int32_t depth, int32_t k, float c, float w_dense,
                         float w_sparse, float* out_scores, int64_t* out_ids)
{
    HR_REQUIRE(nq >= 0 && depth > 0 && k > 0, "bad hybrid shape");
    if (nq == 0) return HIPRAG_OK;
    HR_REQUIRE(q_host && q_offsets_host && out_scores && out_ids, "null argument");
    int32_t d = 0, rc;
    if ((rc = hipidx_dim(dense_h, &d))) return rc;
    if ((rc = hipidx_reserve_search(dense_h, depth))) return rc;   // also makes the index's device current on this thread
    DevBuf q, s64a, ida, s64b, idb, os, oi;
    const size_t nd = (size_t)nq * depth;
    if ((rc = q.reserve((size_t)nq * d * sizeof(float)))) return rc;
    if ((rc = s64a.reserve(nd * 8))) return rc;
    if ((rc = ida.reserve(nd * 8))) return rc;
    if ((rc = s64b.reserve(nd * 8))) return rc;
    if ((rc = idb.reserve(nd * 8))) return rc;
    if ((rc = os.reserve((size_t)nq * k * sizeof(float)))) return rc;
    if ((rc = oi.reserve((size_t)nq * k * 8))) return rc;
    HR_CHECK_HIP(hipMemcpy(q.p, q_host, (size_t)nq * d * sizeof(float), hipMemcpyHostToDevice));
    // The two legs are independent and use different parts of the chip -- the dense scan is HBM-bound, BM25 waits on LDS
    // round trips and barriers -- so BM25 runs on a second stream beside the dense leg (its workgroups fill the gaps
    // around the scan's tail kernels): 123-127 k -> 130-131 k hybrid queries/s at 1M chunks, same results.
    int cur_dev = 0;
    HR_CHECK_HIP(hipGetDevice(&cur_dev));
    SideLease lease;   // declared AFTER the DevBufs: destroyed (= helper stream drained) before they are freed
    if ((rc = side_pool().acquire(cur_dev, lease.s))) return rc;
    lease.held = true;
    lease.busy = true;   // from the first BM25 launch until the main stream has been made to wait for the leg and has drained
    if ((rc = hipbm25_search_dev(bm25_h, term_ids_host, q_offsets_host, nq, depth, s64b.as<double>(), nullptr,
                                 idb.as<int64_t>(), lease.s.st))) return rc;
    HR_CHECK_HIP(hipEventRecord(lease.s.done, lease.s.st));
    if ((rc = hipidx_search_dev(dense_h, q.as<float>(), nq, depth, s64a.as<double>(), nullptr, ida.as<int64_t>(), nullptr))) return rc;
    HR_CHECK_HIP(hipStreamWaitEvent(nullptr, lease.s.done, 0));
    if ((rc = hiprrf_fuse_dev(ida.as<int64_t>(), idb.as<int64_t>(), nq, depth, depth, k, c, w_dense, w_sparse,
                              os.as<float>(), oi.as<int64_t>(), nullptr))) return rc;
    HR_CHECK_HIP(hipMemcpy(out_scores, os.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost));
    HR_CHECK_HIP(hipMemcpy(out_ids, oi.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
    lease.busy = false;   // the blocking copies above ran behind the fusion, which waited for the BM25 leg
    return HIPRAG_OK;
}

// ---- row-sharded hybrid step: the library's two halves around the caller's ONE all-gather ----------------

int32_t hiphybrid_shard_begin_dev(uint64_t dense_h, uint64_t bm25_h, const float* q_dev, const uint32_t* term_ids_host,
                                  const int32_t* q_offsets_host, int32_t nq, int32_t depth, int32_t slot, int64_t* pack_dev,
                                  float* scratch_f32_dev, void* scan_stream, void* tail_stream)
{
    HR_REQUIRE(nq > 0 && depth > 0, "bad hybrid shape");
    HR_REQUIRE(q_dev && q_offsets_host && pack_dev && scratch_f32_dev, "null argument");
    int32_t rc;
    // the HBM-heavy part stays on the caller's scan stream (scans chained one after the other) ...
    if ((rc = hipidx_search_begin_dev(dense_h, q_dev, nq, depth, slot, scan_stream))) return rc;   // makes the device current
    int dev = 0;
    HR_CHECK_HIP(hipGetDevice(&dev));
    hipEvent_t scanned = nullptr;
    if ((rc = step_events().get(dev, slot, scanned))) return rc;
    if (tail_stream != scan_stream) {
        HR_CHECK_HIP(hipEventRecord(scanned, (hipStream_t)scan_stream));
        HR_CHECK_HIP(hipStreamWaitEvent((hipStream_t)tail_stream, scanned, 0));
    }
    // ... everything latency-bound on the tail stream, beside later scans: the dense finish and the BM25 leg fill the two
    // halves of the pack, [leg][score bits | ids][nq][depth]
    const size_t leg = (size_t)2 * nq * depth;
    if ((rc = hipidx_search_finish_dev(dense_h, q_dev, nq, depth, slot, reinterpret_cast<double*>(pack_dev), scratch_f32_dev,
                                       pack_dev + (size_t)nq * depth, tail_stream)))
        return rc;
    return hipbm25_search_dev(bm25_h, term_ids_host, q_offsets_host, nq, depth, reinterpret_cast<double*>(pack_dev + leg),
                              scratch_f32_dev + (size_t)nq * depth, pack_dev + leg + (size_t)nq * depth, tail_stream);
}

int32_t hiphybrid_shard_end_dev(const int64_t* gathered_dev, int32_t n_parts, int32_t nq, int32_t depth, int32_t k,
                                int32_t dense_metric, float c, float w_dense, float w_sparse, int64_t* scratch_dev,
                                float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    HR_REQUIRE(n_parts > 0 && nq > 0 && depth > 0 && k > 0, "bad hybrid shape");
    HR_REQUIRE(gathered_dev && scratch_dev && out_scores_dev && out_ids_dev, "null argument");
    const size_t nd = (size_t)nq * depth, part = 4 * nd;      // one rank's pack: 2 legs x {scores, ids} x nq x depth
    int64_t* dl_s = scratch_dev;                              // merged dense leg: score bits | ids, then the sparse leg
    int64_t* dl_i = scratch_dev + nd;
    int64_t* sl_s = scratch_dev + 2 * nd;
    int64_t* sl_i = scratch_dev + 3 * nd;
    int32_t rc;
    // each leg merged GLOBALLY with the canonical comparator, then RRF over the two global lists (ranks are global: fusing
    // per shard and merging afterwards would be a different function)
    if ((rc = hiprag_merge_topk_dev(reinterpret_cast<const double*>(gathered_dev), gathered_dev + nd, n_parts, nq, depth, depth,
                                    (int64_t)part, dense_metric, reinterpret_cast<double*>(dl_s), nullptr, dl_i, stream)))
        return rc;
    if ((rc = hiprag_merge_topk_dev(reinterpret_cast<const double*>(gathered_dev + 2 * nd), gathered_dev + 3 * nd, n_parts, nq, depth,
                                    depth, (int64_t)part, HIPRAG_METRIC_IP, reinterpret_cast<double*>(sl_s), nullptr, sl_i, stream)))
        return rc;
    return hiprrf_fuse_dev(dl_i, sl_i, nq, depth, depth, k, c, w_dense, w_sparse, out_scores_dev, out_ids_dev, stream);
}

int32_t hiprag_device_sync(int32_t device)
{
    HR_CHECK_HIP(hipSetDevice(device));
    HR_CHECK_HIP(hipDeviceSynchronize());
    return HIPRAG_OK;
}

int32_t hiprag_event_create(uint64_t* out_event)
{
    HR_REQUIRE(out_event, "null out");
    auto ev = std::make_shared<hipEvent_t>();
    HR_CHECK_HIP(hipEventCreate(ev.get()));
    *out_event = events().put(ev);
    return HIPRAG_OK;
}

int32_t hiprag_event_record(uint64_t event, void* stream)
{
    auto ev = events().get(event);
    if (!ev) { set_error("unknown event handle"); return HIPRAG_E_HANDLE; }
    HR_CHECK_HIP(hipEventRecord(*ev, (hipStream_t)stream));
    return HIPRAG_OK;
}

int32_t hiprag_event_elapsed_ms(uint64_t start, uint64_t stop, float* out_ms)
{
    HR_REQUIRE(out_ms, "null out");
    auto a = events().get(start);
    auto b = events().get(stop);
    if (!a || !b) { set_error("unknown event handle"); return HIPRAG_E_HANDLE; }
    HR_CHECK_HIP(hipEventSynchronize(*b));
    HR_CHECK_HIP(hipEventElapsedTime(out_ms, *a, *b));
    return HIPRAG_OK;
}

int32_t hiprag_event_destroy(uint64_t event)
{
    auto ev = events().get(event);
    if (!ev) { set_error("unknown event handle"); return HIPRAG_E_HANDLE; }
    (void)hipEventDestroy(*ev);
    events().erase(event);
    return HIPRAG_OK;
}

}  // extern "C"
