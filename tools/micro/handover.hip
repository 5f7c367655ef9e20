// Two numbers behind the device k-d builder's sums (kd_build_gpu.hip: chained_ranks):
//  1. how long a 32-bit word takes from one workgroup to another (relaxed device-scope atomic store, polled with relaxed
//     device-scope atomic loads): two workgroups hand a counter back and forth, on different XCDs and on the same one;
//  2. how fast atomics on ONE address are served: every wave of a grid adds to the same counter, with and without using
//     the returned value.
// Every wait is bounded, so the kernels end whatever happens.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr uint32_t SPIN_LIMIT = 1u << 20;

__device__ __forceinline__ bool wait_for(const uint32_t *word, uint32_t value) {
    for (uint32_t polls = 0; polls < SPIN_LIMIT; polls++) {
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == value) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// workgroup 0 and workgroup `partner` (index mod 8 = XCD) pass the values 1, 2, 3, ... to each other
__global__ __launch_bounds__(64) void ping_pong(uint32_t *word, uint32_t partner, uint32_t rounds, unsigned long long *ticks, uint32_t *failed) {
    if (threadIdx.x != 0 || (blockIdx.x != 0 && blockIdx.x != partner)) return;
    const bool first = blockIdx.x == 0;
    const unsigned long long t0 = wall_clock64();
    for (uint32_t i = 0; i < rounds; i++) {
        if (first) {
            __hip_atomic_store(word, 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!wait_for(word, 2 * i + 2)) { *failed = 1; return; }
        } else {
            if (!wait_for(word, 2 * i + 1)) { *failed = 1; return; }
            __hip_atomic_store(word, 2 * i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (first) *ticks = wall_clock64() - t0;   // 100 MHz
}

template <bool USE>
__global__ __launch_bounds__(64) void one_address(uint32_t *counter, uint32_t repeats, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint32_t r = 0; r < repeats; r++) {
        if (threadIdx.x == 0) {
            if (USE) acc += atomicAdd(counter, 1u);
            else atomicAdd(counter, 1u);
        }
    }
    if (USE && acc == 0xffffffffu) *sink = acc;
}

int main() {
    uint32_t *d;
    unsigned long long *ticks;
    CK(hipMalloc(&d, 4096));
    CK(hipMalloc(&ticks, 8));
    const uint32_t rounds = 2000;
    for (uint32_t partner : {1u, 4u, 8u, 16u}) {
        CK(hipMemset(d, 0, 4096));
        CK(hipMemset(ticks, 0, 8));
        hipLaunchKernelGGL(ping_pong, dim3(17), dim3(64), 0, 0, d, partner, rounds, ticks, d + 64);
        CK(hipDeviceSynchronize());
        unsigned long long t;
        uint32_t failed;
        CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&failed, d + 64, 4, hipMemcpyDeviceToHost));
        printf("workgroups 0 and %2u (XCD 0 and %u): %s%.2f us per hand-over (store -> seen by the other's poll)\n", partner, partner & 7,
               failed ? "GAVE UP; " : "", (double)t * 10.0 / 1000.0 / (2.0 * rounds));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int use = 0; use < 2; use++) {
        for (uint32_t groups : {256u, 2048u, 16384u}) {
            const uint32_t repeats = 16;
            CK(hipMemset(d, 0, 4096));
            CK(hipEventRecord(e0, 0));
            if (use) hipLaunchKernelGGL(one_address<true>, dim3(groups), dim3(64), 0, 0, d, repeats, d + 64);
            else hipLaunchKernelGGL(one_address<false>, dim3(groups), dim3(64), 0, 0, d, repeats, d + 64);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            uint32_t total;
            CK(hipMemcpy(&total, d, 4, hipMemcpyDeviceToHost));
            printf("%5u waves x %u atomic adds on one address, result %s: %.1f ns per atomic (%u counted)\n", groups, repeats,
                   use ? "used  " : "unused", (double)ms * 1e6 / ((double)groups * repeats), total);
        }
    }
    return 0;
}
