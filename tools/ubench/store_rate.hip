// Micro-benchmark (runs on the GPU box): cost of the sweep-scratch access patterns.  Every workgroup owns a slab (as the
// column kernels do) and writes then re-reads `levels` rows of 256 lanes, 3 values per (level, lane):
//   rows   : 3 dword stores / loads per level          ([level][value][lane], what the kernels do)
//   quads  : 3 dwordx4 stores / loads per 4 levels     ([level / 4][value][lane][4])
//   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>  // 0 rows, 1 quads
__global__ void __launch_bounds__(256) sweep(float *slab, int levels, int reps, float *out) {
    float *base = slab + (size_t)blockIdx.x * levels * 3 * 256;
    const int t = threadIdx.x;
    float acc = 0.f, a = t * 1e-3f, b = 1.f, c = 0.5f;
    for (int r = 0; r < reps; r++) {
        if (MODE == 0) {
            for (int k = 0; k < levels; k++) {
                a = a * 0.999f + 0.001f; b = b * 0.998f + a; c = c * 0.997f + b;
                base[(k * 3 + 0) * 256 + t] = a; base[(k * 3 + 1) * 256 + t] = b; base[(k * 3 + 2) * 256 + t] = c;
            }
            __syncthreads();
            for (int k0 = levels - 16; k0 >= 0; k0 -= 16) {
                float v[48];
#pragma unroll
                for (int j = 0; j < 16; j++)
#pragma unroll
                    for (int i = 0; i < 3; i++) v[j * 3 + i] = base[((k0 + j) * 3 + i) * 256 + t];
#pragma unroll
                for (int j = 0; j < 48; j++) acc = acc * 0.99f + v[j];
            }
        } else {
            for (int k = 0; k < levels; k += 4) {
                float4 A, B, C;
                a = a * 0.999f + 0.001f; b = b * 0.998f + a; c = c * 0.997f + b; A.x = a; B.x = b; C.x = c;
                a = a * 0.999f + 0.001f; b = b * 0.998f + a; c = c * 0.997f + b; A.y = a; B.y = b; C.y = c;
                a = a * 0.999f + 0.001f; b = b * 0.998f + a; c = c * 0.997f + b; A.z = a; B.z = b; C.z = c;
                a = a * 0.999f + 0.001f; b = b * 0.998f + a; c = c * 0.997f + b; A.w = a; B.w = b; C.w = c;
                float4 *q = reinterpret_cast<float4 *>(base + (size_t)(k / 4) * 3 * 1024);
                q[0 * 256 + t] = A; q[1 * 256 + t] = B; q[2 * 256 + t] = C;
            }
            __syncthreads();
            for (int k0 = levels - 16; k0 >= 0; k0 -= 16) {
                float4 v[12];
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 3; i++) v[j * 3 + i] = reinterpret_cast<const float4 *>(base + (size_t)(k0 / 4 + j) * 3 * 1024)[i * 256 + t];
#pragma unroll
                for (int j = 0; j < 12; j++) { acc = acc * 0.99f + v[j].x; acc = acc * 0.99f + v[j].y; acc = acc * 0.99f + v[j].z; acc = acc * 0.99f + v[j].w; }
            }
        }
        __syncthreads();
    }
    if (acc == 1234.5f) out[0] = acc;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, grid = ncu * 4, levels = 64, reps = 40;
    float *slab, *out;
    CK(hipMalloc(&slab, (size_t)grid * levels * 3 * 256 * 4)); CK(hipMalloc(&out, 4));
    printf("%d workgroups x %d levels x 3 values x 256 lanes = %.0f MB of scratch\n", grid, levels, grid * levels * 3 * 1024.0 / 1e6);
    for (int mode = 0; mode < 2; mode++) {
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(sweep<0>, dim3(grid), dim3(256), 0, 0, slab, levels, reps, out);
            else hipLaunchKernelGGL(sweep<1>, dim3(grid), dim3(256), 0, 0, slab, levels, reps, out);
        };
        launch(); CK(hipDeviceSynchronize());
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a, 0)); for (int r = 0; r < 3; r++) launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
        const double wave_levels_per_cu = 4.0 * 4 * levels * reps;   // 4 workgroups x 4 waves
        printf("  %-6s %8.3f ms   %6.1f clk per wave-level per CU (store + load of 3 values) @2.4 GHz\n", mode ? "quads" : "rows", ms,
               ms * 1e-3 * 2.4e9 / wave_levels_per_cu);
    }
    return 0;
}
