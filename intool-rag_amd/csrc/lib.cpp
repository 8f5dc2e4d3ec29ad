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

// The device's SCAN STREAM: one high-priority stream per device, owned by the library (hiprag_scan_stream hands it to
// hosts that chain their own scans, hiprag/sharded.py).  A scan workgroup needs a whole CU (158 of 160 KiB of LDS) and a
// scan is launched on fewer workgroups than the chip has CUs (hipidx_set_spare_cus); the kernels that are to run beside
// it -- BM25 tiles, four to a CU, the finish of the previous launch -- are many small workgroups.  When a scan and such a
// kernel become ready together the dispatcher serves the high-priority queue first: the scan gets its CUs and the small
// workgroups fill the rest.  With equal priorities they take every CU first and the scan's workgroups start late, one by
// one, as CUs drain (the hybrid legs' times then simply add up: 123-137 k hybrid queries/s at 1M chunks, against 140-154 k
// beside each other on the same boxes).  Every dense leg of every hybrid call goes through this one stream, in the order
// the index's mutex admitted the calls (they share the index's workspace).
class ScanStreams {
public:
    int32_t get(int dev, hipStream_t& out)
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = hp_.find(dev);
        if (it == hp_.end()) {
            int least = 0, greatest = 0, cur = 0;
            hipStream_t hp = nullptr;
            HR_CHECK_HIP(hipGetDevice(&cur));
            HR_CHECK_HIP(hipSetDevice(dev));
            hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&hp, hipStreamNonBlocking, greatest);
            (void)hipSetDevice(cur);
            HR_CHECK_HIP(e);
            it = hp_.emplace(dev, hp).first;
        }
        out = it->second;
        return HIPRAG_OK;
    }
    // The device's TAIL STREAMS (two, normal priority): where the kernels that run beside a scan are enqueued -- the finish
    // of the previous launch, an exchange, a merge.  One pair per device for the whole process, not one per index or per
    // wrapper object: HIP has four hardware queues per priority and hands them out in order of first use, streams beyond
    // that SHARE queues, and a process that multiplied its tail streams (one pair per ShardedFlatIndex, a stream per index
    // for the library's own pipeline) slowed its later pipelines down by that alone.
    int32_t get_tail(int dev, int which, hipStream_t& out)
    {
        std::lock_guard<std::mutex> g(mu_);
        const long long key = (long long)dev * 2 + (which & 1);
        auto it = tail_.find(key);
        if (it == tail_.end()) {
            int cur = 0;
            hipStream_t st = nullptr;
            HR_CHECK_HIP(hipGetDevice(&cur));
            HR_CHECK_HIP(hipSetDevice(dev));
            const hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            (void)hipSetDevice(cur);
            HR_CHECK_HIP(e);
            it = tail_.emplace(key, st).first;
        }
        out = it->second;
        return HIPRAG_OK;
    }
    void clear()   // hiprag_shutdown, every device synchronised
    {
        std::lock_guard<std::mutex> g(mu_);
        for (auto& kv : hp_)
            if (hipSetDevice(kv.first) == hipSuccess) (void)hipStreamDestroy(kv.second);
        hp_.clear();
        for (auto& kv : tail_)
            if (hipSetDevice((int)(kv.first / 2)) == hipSuccess) (void)hipStreamDestroy(kv.second);
        tail_.clear();
    }

private:
    std::mutex mu_;
    std::unordered_map<int, hipStream_t> hp_;
    std::unordered_map<long long, hipStream_t> tail_;
};
static ScanStreams& scan_streams()
{
    static ScanStreams s;
    return s;
}
constexpr int kHybridSpareCus = 96;   // of 256: dense top-50 alone loses 0-7 % on 160 workgroups; 64-112 measure within the box-to-box spread (config 2)

// The two events of one hybrid call (legs start / dense leg done), pooled per device.
struct LegEvents {
    hipEvent_t in = nullptr, dense_done = nullptr;
    int dev = -1;
};
class LegEventPool {
public:
    int32_t acquire(int dev, LegEvents& out)
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].dev == dev) { out = free_[i]; free_.erase(free_.begin() + (long)i); return HIPRAG_OK; }
        }
        LegEvents e;
        e.dev = dev;
        HR_CHECK_HIP(hipEventCreateWithFlags(&e.in, hipEventDisableTiming));
        hipError_t rc = hipEventCreateWithFlags(&e.dense_done, hipEventDisableTiming);
        if (rc != hipSuccess) { (void)hipEventDestroy(e.in); HR_CHECK_HIP(rc); }
        out = e;
        return HIPRAG_OK;
    }
    void release(const LegEvents& e)
    {
        std::lock_guard<std::mutex> g(mu_);
        free_.push_back(e);
    }
    void clear()
    {
        std::lock_guard<std::mutex> g(mu_);
        for (const LegEvents& e : free_) {
            if (hipSetDevice(e.dev) != hipSuccess) continue;
            (void)hipEventDestroy(e.in);
            (void)hipEventDestroy(e.dense_done);
        }
        free_.clear();
    }

private:
    std::mutex mu_;
    std::vector<LegEvents> free_;
};
static LegEventPool& leg_events()
{
    static LegEventPool p;
    return p;
}
// One lease per hybrid call.  While `busy` the scan stream may still write into buffers the call's frame owns: every way out
// of the frame (error returns included) first waits for it.
struct LegLease {
    LegEvents e;
    hipStream_t hp = nullptr;
    bool held = false, busy = false;
    ~LegLease()
    {
        if (busy && hp) (void)hipStreamSynchronize(hp);
        if (held) leg_events().release(e);
    }
};

// the dense leg of a hybrid call leaves CUs to the BM25 leg; the index's own setting comes back on every way out
struct SpareCusScope {
    uint64_t h = 0;
    int32_t before = -1;
    int32_t enter(uint64_t handle, int32_t spare)
    {
        int32_t rc = hipidx_get_spare_cus(handle, &before);
        if (rc) { before = -1; return rc; }
        h = handle;
        return before >= spare ? HIPRAG_OK : hipidx_set_spare_cus(handle, spare);
    }
    ~SpareCusScope() { if (before >= 0) (void)hipidx_set_spare_cus(h, before); }
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
    leg_events().clear();
    scan_streams().clear();
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

int32_t hiphybrid_search_dev(uint64_t dense_h, uint64_t bm25_h, const float* q_dev, const uint32_t* term_ids_host,
                             const int32_t* q_offsets_host, int32_t nq, int32_t depth, int32_t k, float c, float w_dense,
                             float w_sparse, int64_t* lists_dev, float* out_scores_dev, int64_t* out_ids_dev, void* stream)
{
    HR_REQUIRE(nq >= 0 && depth > 0 && k > 0, "bad hybrid shape");
    if (nq == 0) return HIPRAG_OK;
    HR_REQUIRE(q_dev && q_offsets_host && lists_dev && out_scores_dev && out_ids_dev, "null argument");
    int32_t rc;
    if ((rc = hipidx_reserve_search(dense_h, depth))) return rc;   // also makes the index's device current on this thread
    int cur_dev = 0;
    HR_CHECK_HIP(hipGetDevice(&cur_dev));
    const size_t nd = (size_t)nq * depth;      // lists_dev: int64 [4][nq][depth] = dense score bits | dense ids | BM25 score bits | BM25 ids
    hipStream_t main = (hipStream_t)stream;
    LegLease lease;
    if ((rc = scan_streams().get(cur_dev, lease.hp))) return rc;
    if ((rc = leg_events().acquire(cur_dev, lease.e))) return rc;
    lease.held = true;
    // the dense leg starts where the caller's stream is now (whatever produced q_dev is ahead of it), on the scan stream ...
    HR_CHECK_HIP(hipEventRecord(lease.e.in, main));
    HR_CHECK_HIP(hipStreamWaitEvent(lease.hp, lease.e.in, 0));
    lease.busy = true;
    {
        SpareCusScope spare;
        if ((rc = spare.enter(dense_h, kHybridSpareCus))) return rc;
        if ((rc = hipidx_search_dev(dense_h, q_dev, nq, depth, reinterpret_cast<double*>(lists_dev), nullptr, lists_dev + nd, lease.hp)))
            return rc;
    }
    HR_CHECK_HIP(hipEventRecord(lease.e.dense_done, lease.hp));
    // ... the BM25 leg beside it on the caller's stream, on the CUs the scan leaves (and on all of them once it is done) ...
    if ((rc = hipbm25_search_dev(bm25_h, term_ids_host, q_offsets_host, nq, depth, reinterpret_cast<double*>(lists_dev + 2 * nd), nullptr,
                                 lists_dev + 3 * nd, stream))) return rc;
    // ... and the fusion behind both
    HR_CHECK_HIP(hipStreamWaitEvent(main, lease.e.dense_done, 0));
    lease.busy = false;   // from here the caller's stream orders everything that touches the caller's buffers
    return hiprrf_fuse_dev(lists_dev + nd, lists_dev + 3 * nd, nq, depth, depth, k, c, w_dense, w_sparse, out_scores_dev, out_ids_dev, stream);
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
    DevBuf q, lists, os, oi;
    if ((rc = q.reserve((size_t)nq * d * sizeof(float)))) return rc;
    if ((rc = lists.reserve((size_t)4 * nq * depth * 8))) return rc;
    if ((rc = os.reserve((size_t)nq * k * sizeof(float)))) return rc;
    if ((rc = oi.reserve((size_t)nq * k * 8))) return rc;
    HR_CHECK_HIP(hipMemcpy(q.p, q_host, (size_t)nq * d * sizeof(float), hipMemcpyHostToDevice));
    rc = hiphybrid_search_dev(dense_h, bm25_h, q.as<float>(), term_ids_host, q_offsets_host, nq, depth, k, c, w_dense, w_sparse,
                              lists.as<int64_t>(), os.as<float>(), oi.as<int64_t>(), nullptr);
    // whatever the call enqueued (helper streams included: the null stream waits for them, or the call drained them on its
    // error path) is done before the frame's buffers go
    if (rc) { (void)hipDeviceSynchronize(); return rc; }
    HR_CHECK_HIP(hipMemcpy(out_scores, os.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost));
    HR_CHECK_HIP(hipMemcpy(out_ids, oi.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
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

int32_t hiprag_scan_stream(int32_t device, void** out_stream)
{
    HR_REQUIRE(out_stream, "null out");
    hipStream_t st = nullptr;
    int32_t rc = scan_streams().get(device, st);
    if (rc) return rc;
    *out_stream = (void*)st;
    return HIPRAG_OK;
}

int32_t hiprag_tail_stream(int32_t device, int32_t which, void** out_stream)
{
    HR_REQUIRE(out_stream && (which == 0 || which == 1), "which must be 0 or 1");
    hipStream_t st = nullptr;
    int32_t rc = scan_streams().get_tail(device, which, st);
    if (rc) return rc;
    *out_stream = (void*)st;
    return HIPRAG_OK;
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
