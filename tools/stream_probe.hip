// build: hipcc -O3 --offload-arch=gfx950 -o stream_probe tools/stream_probe.hip
// Does the streaming rate of a statically partitioned read depend on the allocation?  Pattern 0: wave w reads the
// contiguous range of blocks [w*bpw, (w+1)*bpw) (the dense scan's partition, 1 MiB apart at 1M x 1024 bf16);
// pattern 1: block-cyclic, wave w reads blocks w, w + W, w + 2W, ...  Several allocations, several repeats each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(512) void probe(const f32x4* __restrict__ x, long nblocks, int bpw, int passes, float* out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long W = (long)gridDim.x * 8, gw = (long)blockIdx.x * 8 + wave;
    f32x4 acc = {0, 0, 0, 0};
    for (int pass = 0; pass < passes; ++pass) {
        for (int i = 0; i < bpw; ++i) {
            const long b = PATTERN == 0 ? gw * bpw + i : (long)i * W + gw;
            if (b >= nblocks) break;
            const f32x4* src = x + b * 4096 + lane;   // a block = 64 KiB = 4096 float4
            for (int p = 0; p < 64; p += 16) {
                f32x4 r[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) r[u] = __builtin_nontemporal_load(src + (p + u) * 64);
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += r[u];
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

int main()
{
    const long nblocks = 31250;
    const size_t bytes = (size_t)39550 * 65536;   // the capacity the index ends up with after eight adds
    float* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<void*> keep;
    for (int trial = 0; trial < 8; ++trial) {
        void* junk = nullptr; CK(hipMalloc(&junk, (size_t)(37 + 61 * trial) << 20));   // perturb the placement
        void* x; CK(hipMalloc(&x, bytes)); CK(hipMemset(x, 0, bytes));
        printf("alloc %d at %p:", trial, x);
        for (int pat = 0; pat < 2; ++pat) {
            for (int rep = 0; rep < 3; ++rep) {
                const int bpw = 16, passes = 8;
                for (int it = 0; it < 2; ++it) {
                    if (it == 1) CK(hipEventRecord(e0));
                    for (int k = 0; k < (it == 0 ? 2 : 10); ++k) {
                        if (pat == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, (const f32x4*)x, nblocks, bpw, passes, out);
                        else hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, (const f32x4*)x, nblocks, bpw, passes, out);
                    }
                }
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf(" %s%.0f", rep == 0 ? (pat == 0 ? "contiguous " : "| cyclic ") : "", 10.0 * 8 * nblocks * 65536.0 / ms / 1e6);
            }
        }
        printf(" GB/s\n");
        keep.push_back(x);
        CK(hipFree(junk));
        if (keep.size() > 2) { CK(hipFree(keep.front())); keep.erase(keep.begin()); }
    }
    return 0;
}
