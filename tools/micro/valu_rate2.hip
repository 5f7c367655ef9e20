// Issue throughput per CU of single vector instructions (four independent ones per group), 8 and 24 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define REP8(BODY) BODY BODY BODY BODY BODY BODY BODY BODY
template <int MODE>
__global__ __launch_bounds__(64) void k(int iters, uint32_t *out) {
    uint32_t u0 = threadIdx.x * 3 + 1, u1 = threadIdx.x + 7, u2 = 2, u3 = 3, w = 0x3f800001u, m = 0x0ff00ff0u;
    double p0 = threadIdx.x, p1 = 1.0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP8(asm volatile("v_bfi_b32 %0, %5, %0, %4\n v_bfi_b32 %1, %5, %1, %4\n v_bfi_b32 %2, %5, %2, %4\n v_bfi_b32 %3, %5, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 1) { REP8(asm volatile("v_bfe_i32 %0, %0, 28, 1\n v_bfe_i32 %1, %1, 28, 1\n v_bfe_i32 %2, %2, 28, 1\n v_bfe_i32 %3, %3, 28, 1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 2) { REP8(asm volatile("v_bfe_u32 %0, %0, 3, 5\n v_bfe_u32 %1, %1, 3, 5\n v_bfe_u32 %2, %2, 3, 5\n v_bfe_u32 %3, %3, 3, 5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 3) { REP8(asm volatile("v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 4) { REP8(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 5) { REP8(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 6) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 7) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 8) { REP8(asm volatile("v_and_or_b32 %0, %0, %5, %4\n v_and_or_b32 %1, %1, %5, %4\n v_and_or_b32 %2, %2, %5, %4\n v_and_or_b32 %3, %3, %5, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 9) { REP8(asm volatile("v_lshl_or_b32 %0, %0, 3, %4\n v_lshl_or_b32 %1, %1, 3, %4\n v_lshl_or_b32 %2, %2, 3, %4\n v_lshl_or_b32 %3, %3, 3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 10) { REP8(asm volatile("v_add3_u32 %0, %0, %5, %4\n v_add3_u32 %1, %1, %5, %4\n v_add3_u32 %2, %2, %5, %4\n v_add3_u32 %3, %3, %5, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 11) { REP8(asm volatile("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 12) { REP8(asm volatile("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 13) { REP8(asm volatile("v_med3_f32 %0, %0, %5, %4\n v_med3_f32 %1, %1, %5, %4\n v_med3_f32 %2, %2, %5, %4\n v_med3_f32 %3, %3, %5, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 14) { REP8(asm volatile("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 15) { REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 16) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 17) { REP8(asm volatile("v_cmp_class_f32 vcc, %0, %4\n v_cmp_class_f32 vcc, %1, %4\n v_cmp_class_f32 vcc, %2, %4\n v_cmp_class_f32 vcc, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 18) { REP8(asm volatile("v_cmp_lt_i32 vcc, %0, %4\n v_cmp_lt_i32 vcc, %1, %4\n v_cmp_lt_i32 vcc, %2, %4\n v_cmp_lt_i32 vcc, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 19) { REP8(asm volatile("v_perm_b32 %0, %0, %5, %4\n v_perm_b32 %1, %1, %5, %4\n v_perm_b32 %2, %2, %5, %4\n v_perm_b32 %3, %3, %5, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 20) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %5, %4\n v_mad_u32_u24 %1, %1, %5, %4\n v_mad_u32_u24 %2, %2, %5, %4\n v_mad_u32_u24 %3, %3, %5, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 21) { REP8(asm volatile("v_sub_f32 %0, s20, %0\n v_sub_f32 %1, s20, %1\n v_sub_f32 %2, s20, %2\n v_sub_f32 %3, s20, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 22) { REP8(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
        if (MODE == 23) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1" : "+v"(p0), "+v"(p1));) }
        if (MODE == 24) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1" : "+v"(p0), "+v"(p1));) }
        if (MODE == 25) { REP8(asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s20, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s20, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(w), "v"(m) : "vcc", "s20");) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = u0 + u1 + u2 + u3 + (uint32_t)p0 + (uint32_t)p1;
}
template <int MODE>
void run(const char *name, uint32_t *out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-30s", name);
    for (int wpc : {8, 24}) {
        const int iters = 4000;
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wpc), dim3(64), 0, 0, 10, out);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wpc), dim3(64), 0, 0, iters, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %2d waves/CU: %.2f per cycle per CU", wpc, (double)wpc * iters * 32 / (ms * 1e-3 * 2.4e9));
    }
    printf("\n");
}
int main() {
    uint32_t *out;
    CK(hipMalloc(&out, 256 * 32 * 64 * 4));
    run<0>("v_bfi_b32", out);
    run<1>("v_bfe_i32 (1 bit)", out);
    run<2>("v_bfe_u32", out);
    run<3>("v_ashrrev_i32", out);
    run<4>("v_and_b32", out);
    run<5>("v_xor_b32", out);
    run<6>("v_add_u32", out);
    run<7>("v_lshl_add_u32", out);
    run<8>("v_and_or_b32", out);
    run<9>("v_lshl_or_b32", out);
    run<10>("v_add3_u32", out);
    run<11>("v_min_f32", out);
    run<12>("v_max_i32", out);
    run<13>("v_med3_f32", out);
    run<14>("v_sub_f32", out);
    run<15>("v_mul_f32", out);
    run<16>("v_cndmask_b32_e32 (vcc)", out);
    run<17>("v_cmp_class_f32 -> vcc", out);
    run<18>("v_cmp_lt_i32 -> vcc", out);
    run<19>("v_perm_b32", out);
    run<20>("v_mad_u32_u24", out);
    run<21>("v_sub_f32 with sgpr operand", out);
    run<22>("v_add_u32 with literal", out);
    run<23>("v_pk_mul_f32", out);
    run<24>("v_pk_fma_f32", out);
    run<25>("v_readfirstlane_b32", out);
    return 0;
}
