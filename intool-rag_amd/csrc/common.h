// common.h -- host-side plumbing shared by the libhiprag translation units (error state, handle registry).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/hiprag.h"

namespace hiprag {

void set_error(const char* fmt, ...);  // thread-local message, printf-style

#define HR_CHECK_HIP(expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            hiprag::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                              __LINE__);                                                       \
            return HIPRAG_E_HIP;                                                               \
        }                                                                                      \
    } while (0)

#define HR_REQUIRE(cond, ...)              \
    do {                                   \
        if (!(cond)) {                     \
            hiprag::set_error(__VA_ARGS__); \
            return HIPRAG_E_INVALID;       \
        }                                  \
    } while (0)

// Handle registry: uint64 -> shared_ptr<T>.  One registry per object kind.
template <typename T>
class Registry {
public:
    uint64_t put(std::shared_ptr<T> p)
    {
        std::lock_guard<std::mutex> g(mu_);
        uint64_t h = next_++;
        map_[h] = std::move(p);
        return h;
    }
    std::shared_ptr<T> get(uint64_t h)
    {
        std::lock_guard<std::mutex> g(mu_);
        auto it = map_.find(h);
        return it == map_.end() ? nullptr : it->second;
    }
    bool erase(uint64_t h)
    {
        std::lock_guard<std::mutex> g(mu_);
        return map_.erase(h) != 0;
    }
    size_t clear()   // drops every handle (hiprag_shutdown); objects die when their last user lets go
    {
        std::lock_guard<std::mutex> g(mu_);
        const size_t n = map_.size();
        map_.clear();
        return n;
    }

private:
    std::mutex mu_;
    std::unordered_map<uint64_t, std::shared_ptr<T>> map_;
    uint64_t next_ = 0x1000;
};

// RAII device buffer (never shrinks implicitly).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int32_t reserve(size_t want)
    {
        if (want <= bytes) return HIPRAG_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        HR_CHECK_HIP(hipMalloc(&p, want));
        bytes = want;
        return HIPRAG_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// Pinned host buffer (device-visible under the same address): staging for asynchronous copies that must not block the host,
// and a place kernels can write small results to directly.
struct PinBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int32_t reserve(size_t want)
    {
        if (want <= bytes) return HIPRAG_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        if (want < 4096) want = 4096;
        HR_CHECK_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        bytes = want;
        return HIPRAG_OK;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// hiprag_shutdown: every translation unit owns the registry of its handle type
size_t clear_dense_registry();
size_t clear_bm25_registry();
size_t clear_encoder_registry();

}  // namespace hiprag
