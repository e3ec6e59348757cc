// Micro-benchmark 3: a wave's progress on DEPENDENT VALU chains (what the solve kernels mostly are)
// versus independent ones, for 1..8 waves per SIMD.  ns per instruction per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 1e-3f + i;
    const float b = 1.0001f, c = 1e-4f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64 / CHAINS; u++) {
#pragma unroll
            for (int j = 0; j < CHAINS; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    if (s == 12345.678f) out[0] = s;
}

template <int CHAINS>
void run() {
    float *out; hipMalloc(&out, 4);
    const int iters = 2048;
    printf("%d independent chain(s) per wave:", CHAINS);
    for (int w : {1, 2, 4, 8}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<CHAINS>, dim3(256 * w), dim3(256), 0, 0, out, 16);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CHAINS>, dim3(256 * w), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 64;
        printf("  w=%d: %5.2f ns/inst/wave, %5.2f ns/inst/SIMD", w, ms * 1e6 / n, ms * 1e6 / n / w);
    }
    printf("\n");
    hipFree(out);
}

int main() { run<1>(); run<2>(); run<4>(); run<8>(); return 0; }
