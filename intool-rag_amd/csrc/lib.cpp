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
