// Where do the workgroups of a persistent launch land? One wave per workgroup with the walk's LDS footprint; every wave
// records its XCC id and hardware id (shader engine, CU, SIMD, wave slot).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void probe(uint32_t *out, int spin) {
    __shared__ uint32_t pad[7168 / 4];
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stay resident for a while so that the whole grid is placed like a persistent launch
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + pad[1] - 1; }
}

int main() {
    const int n = 256 * 24;
    uint32_t *d;
    CK(hipMalloc(&d, n * 8));
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, 0, d, 20000);   // 200 us
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(2 * n);
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    std::map<uint32_t, int> per_cu;
    int xcd_match = 0;
    for (int i = 0; i < n; i++) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        if ((uint32_t)(i & 7) == xcc) xcd_match++;
    }
    printf("workgroups whose index %% 8 equals their XCC id: %d of %d; distinct (xcc, se, sh, cu): %zu\n", xcd_match, n, per_cu.size());
    printf("first 96 workgroups: index -> xcc se sh cu simd\n");
    for (int i = 0; i < 96; i++) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        printf("%4d -> %u %u %u %2u %u%s", i, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, (hw >> 4) & 3, (i % 4 == 3) ? "\n" : "   |   ");
    }
    printf("workgroups of XCC 0 (index = 8 j): j -> se cu\n");
    for (int j = 0; j < 96; j++) {
        const uint32_t hw = h[2 * (8 * j)];
        printf("%3d:%u.%-2u%s", j, (hw >> 13) & 7, (hw >> 8) & 0xf, (j % 12 == 11) ? "\n" : "  ");
    }
    printf("waves per CU: ");
    std::map<int, int> hist;
    for (auto &kv : per_cu) hist[kv.second]++;
    for (auto &kv : hist) printf("%d CUs with %d, ", kv.second, kv.first);
    printf("\n(se, cu) keys seen on XCC 0: ");
    for (auto &kv : per_cu) if ((kv.first >> 12) == 0) printf("%u.%u ", (kv.first >> 8) & 0xf, kv.first & 0xf);
    printf("\n");
    return 0;
}
