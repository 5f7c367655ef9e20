// Fully divergent 8-byte gathers from an L2-resident table with different cache-policy bits on the load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ uint2 ld(const uint2 *p) {
    uint2 v;
    if (MODE == 0) { v = *p; }
    else if (MODE == 1) { asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
    else if (MODE == 2) { asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
    else if (MODE == 3) { asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
    else if (MODE == 4) { asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
    else { asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); }
    return v;
}

template <int MODE>
__global__ __launch_bounds__(64) void k_gather(const uint2 *__restrict__ tab, uint32_t mask, int iters, uint32_t *out) {
    uint32_t idx = (blockIdx.x * 64u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        uint2 v = ld<MODE>(tab + (idx & mask));
        acc += v.y;
        idx = v.x + acc;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char *name, const uint2 *tab, uint32_t entries, uint32_t *out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int CUS = 256, wpc = 16, iters = 2000;
    hipLaunchKernelGGL(k_gather<MODE>, dim3(CUS * wpc), dim3(64), 0, 0, tab, entries - 1, 10, out);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(CUS * wpc), dim3(64), 0, 0, tab, entries - 1, iters, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("table %6u KB %-8s: %.3f ms, %.1f cycles@2.4GHz per wave-gather per CU\n", entries * 8 / 1024, name, ms, ms * 1e6 / (wpc * iters) * 2.4);
}

int main() {
    uint32_t *out;
    CK(hipMalloc(&out, 256 * 64 * 64 * 4));
    for (uint32_t entries : {1u << 12, 1u << 18}) {
        std::vector<uint2> h(entries);
        uint32_t s = 12345;
        for (auto &e : h) { s = s * 1664525u + 1013904223u; e.x = s >> 3; e.y = s & 7; }
        uint2 *tab;
        CK(hipMalloc(&tab, entries * 8));
        CK(hipMemcpy(tab, h.data(), entries * 8, hipMemcpyHostToDevice));
        run<0>("plain", tab, entries, out);
        run<1>("sc0", tab, entries, out);
        run<2>("sc1", tab, entries, out);
        run<3>("nt", tab, entries, out);
        run<4>("sc0 sc1", tab, entries, out);
        run<5>("sc0 nt", tab, entries, out);
        CK(hipFree(tab));
    }
    return 0;
}
