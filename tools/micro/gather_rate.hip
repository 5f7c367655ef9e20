// Measures how fast one CU serves fully divergent 8-byte gathers (every lane its own cache line) from the
// vector L1 / L2 and from LDS. Build: hipcc -O3 --offload-arch=gfx950 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_gather(const uint2 *__restrict__ tab, uint32_t mask, int iters, uint32_t *out) {
    uint32_t idx = (blockIdx.x * 64u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        uint2 v = tab[idx & mask];
        acc += v.y;
        idx = v.x + acc;          // dependent chain, like a tree descent
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_gather_lds(const uint2 *__restrict__ tab, uint32_t mask, int iters, uint32_t *out) {
    extern __shared__ uint2 s_tab[];
    for (uint32_t i = threadIdx.x; i <= mask; i += WAVES * 64) s_tab[i] = tab[i];
    __syncthreads();
    uint32_t idx = (blockIdx.x * WAVES * 64u + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        uint2 v = s_tab[idx & mask];
        acc += v.y;
        idx = v.x + acc;
    }
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = acc;
}

int main() {
    const int CUS = 256;
    uint32_t *out;
    CK(hipMalloc(&out, CUS * 64 * 64 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (uint32_t entries : {1u << 10, 1u << 12, 1u << 15, 1u << 18, 1u << 21}) {
        std::vector<uint2> h(entries);
        uint32_t s = 12345;
        for (auto &e : h) { s = s * 1664525u + 1013904223u; e.x = s >> 3; e.y = s & 7; }
        uint2 *tab;
        CK(hipMalloc(&tab, entries * 8));
        CK(hipMemcpy(tab, h.data(), entries * 8, hipMemcpyHostToDevice));
        for (int wpc : {4, 8, 16, 24, 32}) {
            const int iters = 2000;
            hipLaunchKernelGGL(k_gather, dim3(CUS * wpc), dim3(64), 0, 0, tab, entries - 1, 10, out);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather, dim3(CUS * wpc), dim3(64), 0, 0, tab, entries - 1, iters, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double instr_per_cu = (double)wpc * iters;
            printf("global table %8u KB waves/CU %2d: %.3f ms, %.1f ns per wave-gather per CU (%.1f cycles @2.4GHz), %.2f Glanes/s\n", entries * 8 / 1024, wpc, ms,
                   ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4, (double)CUS * wpc * 64 * iters / ms / 1e6);
        }
        if (entries * 8 <= 65536) {
            const int iters = 2000;
            hipLaunchKernelGGL(k_gather_lds<16>, dim3(CUS), dim3(1024), entries * 8, 0, tab, entries - 1, iters, out);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather_lds<16>, dim3(CUS), dim3(1024), entries * 8, 0, tab, entries - 1, iters, out);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("LDS    table %8u KB waves/CU 16: %.3f ms, %.1f cycles per wave-gather per CU\n", entries * 8 / 1024, ms, ms * 1e6 / (16.0 * iters) * 2.4);
        }
        CK(hipFree(tab));
    }
    return 0;
}
