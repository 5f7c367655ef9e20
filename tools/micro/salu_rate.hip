// Scalar-ALU and VALU issue throughput per CU as a function of resident waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(int iters, uint32_t *out) {
    uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {   // 32 independent-ish scalar adds
#pragma unroll
            for (int k = 0; k < 8; k++)
                asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_add_u32 %3, %3, 7"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MODE == 1) {   // 32 vector adds
#pragma unroll
            for (int k = 0; k < 8; k++)
                asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        } else {   // 16 scalar + 16 vector interleaved
#pragma unroll
            for (int k = 0; k < 4; k++)
                asm volatile("s_add_u32 %0, %0, 1\n v_add_f32 %4, %4, %4\n s_add_u32 %1, %1, 3\n v_add_f32 %5, %5, %5\n"
                             "s_add_u32 %2, %2, 5\n v_add_f32 %6, %6, %6\n s_add_u32 %3, %3, 7\n v_add_f32 %7, %7, %7"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "scc");
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = s0 + s1 + s2 + s3 + (uint32_t)(v0 + v1 + v2 + v3);
}

template <int MODE>
void run(const char *name, uint32_t *out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wpc : {1, 2, 4, 8, 16, 32}) {
        const int iters = 20000;
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wpc), dim3(64), 0, 0, 10, out);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wpc), dim3(64), 0, 0, iters, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double instr_per_cu = (double)wpc * iters * 32;
        printf("%-12s waves/CU %2d: %.3f ms  -> %.2f wave-instructions per cycle per CU (2.4 GHz)\n", name, wpc, ms, instr_per_cu / (ms * 1e-3 * 2.4e9));
    }
}

int main() {
    uint32_t *out;
    CK(hipMalloc(&out, 256 * 32 * 64 * 4));
    run<0>("SALU", out);
    run<1>("VALU", out);
    run<2>("SALU+VALU", out);
    return 0;
}
