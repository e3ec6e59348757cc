// Micro-benchmark 2: per-instruction issue cost in SHADER CYCLES (s_memtime) for the VALU ops the
// solve kernels are made of.  Each test issues 8 independent instructions x 8 x iters per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 1e-3f + i;
    const float b = 1.0001f, c = 1e-4f;
    unsigned long long m = 0x5555555555555555ull;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#define OP1(name) asm volatile(name " %0, %0\n" name " %1, %1\n" name " %2, %2\n" name " %3, %3\n" name " %4, %4\n" name " %5, %5\n" name " %6, %6\n" name " %7, %7\n" \
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))
#define OP2(name) asm volatile(name " %0, %0, %8\n" name " %1, %1, %8\n" name " %2, %2, %8\n" name " %3, %3, %8\n" name " %4, %4, %8\n" name " %5, %5, %8\n" name " %6, %6, %8\n" name " %7, %7, %8\n" \
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b))
#define OP3(name) asm volatile(name " %0, %0, %8, %9\n" name " %1, %1, %8, %9\n" name " %2, %2, %8, %9\n" name " %3, %3, %8, %9\n" name " %4, %4, %8, %9\n" name " %5, %5, %8, %9\n" name " %6, %6, %8, %9\n" name " %7, %7, %8, %9\n" \
                               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c))
            if (MODE == 0) OP3("v_fma_f32");
            else if (MODE == 1) OP2("v_mul_f32");
            else if (MODE == 2) OP2("v_max_f32");
            else if (MODE == 3) OP2("v_add_u32");
            else if (MODE == 4) OP1("v_exp_f32");
            else if (MODE == 5) OP1("v_rcp_f32");
            else if (MODE == 6) OP1("v_sqrt_f32");
            else if (MODE == 7) OP1("v_log_f32");
            else if (MODE == 8) OP1("v_cvt_f32_i32");
            else if (MODE == 9) OP1("v_mov_b32");
            else if (MODE == 10)  // select on an SGPR-pair mask (VOP3)
                asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                             "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "s"(m));
            else if (MODE == 11)  // compare into an SGPR pair then select: the usual pair
                asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_gt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                             "v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_gt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "vcc");
            else if (MODE == 12)  // compares only (VOPC -> vcc)
                asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
                             "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b) : "vcc");
            else if (MODE == 13)  // DPP move (the reductions)
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                             "v_add_f32_dpp %6, %6, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
            else if (MODE == 14) OP2("v_ldexp_f32");
            else if (MODE == 15) OP3("v_mad_u32_u24");
            else if (MODE == 16) OP2("v_fmac_f32");
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char *name) {
    float *out; unsigned long long *cyc; hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    const int iters = 2048;
    printf("%-26s", name);
    for (int w : {1, 2, 4, 8}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, out, cyc, 16);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)iters * 64;  // instructions per wave
        printf("  w=%d: %6.2f cyc/inst/SIMD (wall %6.3f ms, clk %.2f GHz)", w, (double)c / n / w, ms, c / (ms * 1e6));
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32"); run<16>("v_fmac_f32 (VOP2)"); run<1>("v_mul_f32"); run<2>("v_max_f32"); run<3>("v_add_u32");
    run<15>("v_mad_u32_u24"); run<9>("v_mov_b32"); run<8>("v_cvt_f32_i32"); run<14>("v_ldexp_f32");
    run<4>("v_exp_f32"); run<5>("v_rcp_f32"); run<6>("v_sqrt_f32"); run<7>("v_log_f32");
    run<10>("v_cndmask_b32 (sgpr mask)"); run<12>("v_cmp_gt_f32 -> vcc"); run<11>("v_cmp + v_cndmask (x4)");
    run<13>("v_add_f32_dpp");
    return 0;
}
