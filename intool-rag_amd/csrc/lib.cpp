// lib.cpp -- library-level entry points of libhiprag.so: error state, device queries, HIP-event timing.
#include "common.h"

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
    if ((rc = hipidx_search_dev(dense_h, q.as<float>(), nq, depth, s64a.as<double>(), nullptr, ida.as<int64_t>(), nullptr))) return rc;
    if ((rc = hipbm25_search_dev(bm25_h, term_ids_host, q_offsets_host, nq, depth, s64b.as<double>(), nullptr,
                                 idb.as<int64_t>(), nullptr))) return rc;
    if ((rc = hiprrf_fuse_dev(ida.as<int64_t>(), idb.as<int64_t>(), nq, depth, depth, k, c, w_dense, w_sparse,
                              os.as<float>(), oi.as<int64_t>(), nullptr))) return rc;
    HR_CHECK_HIP(hipMemcpy(out_scores, os.p, (size_t)nq * k * sizeof(float), hipMemcpyDeviceToHost));
    HR_CHECK_HIP(hipMemcpy(out_ids, oi.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost));
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
