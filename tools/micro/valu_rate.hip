// Issue throughput per CU of the vector instructions the k-d walk is made of, at 24 resident waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// 4 independent instructions per block, 8 blocks per loop trip
#define REP8(BODY) BODY BODY BODY BODY BODY BODY BODY BODY
template <int MODE>
__global__ __launch_bounds__(64) void k(int iters, uint32_t *out) {
    float v0 = threadIdx.x + 1.0f, v1 = 1.5f, v2 = 2.5f, v3 = 3.5f, w = 1.000001f;
    uint32_t u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3;
    unsigned long long m = 0x5555555555555555ull, q0 = threadIdx.x, q1 = 5;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(w));) }
        if (MODE == 1) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(w), "s"(m));) }
        if (MODE == 2) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" : : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "vcc");) }
        if (MODE == 3) { REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n v_cmp_lt_f32 s[24:25], %2, %3\n v_cmp_lt_f32 s[26:27], %3, %0" : : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (MODE == 4) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));) }
        if (MODE == 5) { REP8(asm volatile("v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_scale_f32 %1, vcc, %1, %4, %1\n v_div_scale_f32 %2, vcc, %2, %4, %2\n v_div_scale_f32 %3, vcc, %3, %4, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(w) : "vcc");) }
        if (MODE == 6) { REP8(asm volatile("v_div_fmas_f32 %0, %0, %4, %0\n v_div_fmas_f32 %1, %1, %4, %1\n v_div_fmas_f32 %2, %2, %4, %2\n v_div_fmas_f32 %3, %3, %4, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(w) : "vcc");) }
        if (MODE == 7) { REP8(asm volatile("v_div_fixup_f32 %0, %0, %4, %0\n v_div_fixup_f32 %1, %1, %4, %1\n v_div_fixup_f32 %2, %2, %4, %2\n v_div_fixup_f32 %3, %3, %4, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(w));) }
        if (MODE == 8) { REP8(asm volatile("v_and_b32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_lshlrev_b32 %2, 1, %2\n v_bfe_u32 %3, %3, 3, 5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0));) }
        if (MODE == 9) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 3, %1\n v_lshl_add_u64 %1, %1, 3, %0\n v_lshl_add_u64 %0, %0, 3, %1\n v_lshl_add_u64 %1, %1, 3, %0" : "+v"(q0), "+v"(q1));) }
        if (MODE == 10) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));) }
        if (MODE == 11) { REP8(asm volatile("v_mbcnt_lo_u32_b32 %0, %4, 0\n v_mbcnt_hi_u32_b32 %1, %4, %0\n v_mbcnt_lo_u32_b32 %2, %4, 0\n v_mbcnt_hi_u32_b32 %3, %4, %2" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"((uint32_t)m));) }
        if (MODE == 12) { REP8(asm volatile("v_max_i32 %0, %0, %4\n v_sub_f32 %1, %1, %5\n v_mul_f32 %2, %2, %5\n v_subbrev_co_u32 %3, vcc, 0, %3, vcc" : "+v"(u0), "+v"(v1), "+v"(v2), "+v"(u3) : "v"(u1), "v"(w) : "vcc");) }
    }
    out[blockIdx.x * 64 + threadIdx.x] = u0 + u1 + u2 + u3 + (uint32_t)(v0 + v1 + v2 + v3) + (uint32_t)q0 + (uint32_t)q1;
}

template <int MODE>
void run(const char *name, uint32_t *out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wpc : {4, 8, 24}) {
        const int iters = 5000;
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wpc), dim3(64), 0, 0, 10, out);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wpc), dim3(64), 0, 0, iters, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double instr_per_cu = (double)wpc * iters * 32;
        printf("%-28s waves/CU %2d: %7.3f ms -> %.2f wave-instructions per cycle per CU at 2.4 GHz\n", name, wpc, ms, instr_per_cu / (ms * 1e-3 * 2.4e9));
    }
}

int main() {
    uint32_t *out;
    CK(hipMalloc(&out, 256 * 32 * 64 * 4));
    run<0>("v_fma_f32", out);
    run<1>("v_cndmask_b32 (sgpr mask)", out);
    run<2>("v_cmp_lt_f32 -> vcc", out);
    run<3>("v_cmp_lt_f32 -> sgpr pair", out);
    run<4>("v_rcp_f32", out);
    run<5>("v_div_scale_f32", out);
    run<6>("v_div_fmas_f32", out);
    run<7>("v_div_fixup_f32", out);
    run<8>("and/add/lshl/bfe u32", out);
    run<9>("v_lshl_add_u64", out);
    run<10>("v_mov_b32", out);
    run<11>("v_mbcnt", out);
    run<12>("max_i32/sub/mul/subbrev", out);
    return 0;
}
