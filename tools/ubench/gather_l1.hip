// Micro-benchmark (runs on the GPU box): what a wave-wide gather costs the CU's vector L1 by bytes per lane, lane stride and
// alignment.  Every wave reads rows of a table (16 KB: L1-resident, 8 MB: L2-resident) at pseudo-random row indices; per
// variant it prints the wave-instructions per microsecond per CU and the implied cycles per instruction at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 -o gather_l1 gather_l1.hip && ./gather_l1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int W> struct Vec;
template <> struct Vec<4> { float v[1]; };
template <> struct Vec<8> { float v[2]; };
template <> struct Vec<12> { float v[3]; };
template <> struct Vec<16> { float v[4]; };

// lane byte offset inside a row = band * band_bytes + lane_in_band * stride + misalign ; rows are row_bytes apart
template <int W>
__global__ void __launch_bounds__(256) gather(const char *table, unsigned nrows, unsigned row_bytes, unsigned stride, unsigned band_bytes,
                                              unsigned misalign, int iters, float *out, bool per_band_rows, unsigned active) {
    const unsigned tid = threadIdx.x, lane = tid & 63, band = lane >> 4, lib = lane & 15;
    const unsigned lane_off = band * band_bytes + lib * stride + misalign;
    unsigned seed = blockIdx.x * 9781u + (tid >> 6) * 7919u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            seed = seed * 1664525u + 1013904223u;
            unsigned r = (seed >> 8) + (per_band_rows ? band * 0x9e37u : 0u);
            r &= nrows - 1;
            if (lane < active) {   // inactive lanes issue nothing (exec mask)
                const Vec<W> x = *reinterpret_cast<const Vec<W> *>(table + (size_t)r * row_bytes + lane_off);
#pragma unroll
                for (int i = 0; i < W / 4; i++) part += x.v[i];
            }
        }
        acc += part;
    }
    if (acc == 12345.678f) out[0] = acc;
}

struct Case { const char *name; int W; unsigned stride, band_bytes, misalign; bool per_band; unsigned active = 64; };

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const double clk_ghz = prop.clockRate * 1e-6;
    printf("%s: %d CUs, clockRate %.2f GHz\n", prop.name, ncu, clk_ghz);
    const size_t big = 8u << 20;
    char *table; float *out;
    CK(hipMalloc(&table, big + 4096)); CK(hipMemset(table, 0, big + 4096)); CK(hipMalloc(&out, 4));
    const Case cases[] = {
        {"dword   dense (64 lanes x 4 B = 256 B, aligned)", 4, 4, 64, 0, false},
        {"dwordx2 dense (512 B, aligned)", 8, 8, 128, 0, false},
        {"dwordx4 dense (1 KB, aligned)", 16, 16, 256, 0, false},
        {"dwordx4 dense, base + 4 B (misaligned)", 16, 16, 256, 4, false},
        {"dwordx4 lane stride 12 B (3 of 4 useful, overlapping)", 16, 12, 192, 0, false},
        {"dwordx4 lane stride 8 B (overlapping)", 16, 8, 128, 0, false},
        {"dwordx4 lane stride 4 B (overlapping)", 16, 4, 64, 0, false},
        {"dwordx2 lane stride 4 B", 8, 4, 64, 0, false},
        {"dword   4 bands in 4 different rows (4 x 64 B)", 4, 4, 64, 0, true},
        {"dwordx2 4 bands in 4 rows (4 x 128 B)", 8, 8, 128, 0, true},
        {"dwordx4 4 bands in 4 rows (4 x 256 B)", 16, 16, 256, 0, true},
        {"dwordx4 4 bands in 4 rows, stride 12 B (4 x 192 B)", 16, 12, 192, 0, true},
        {"dwordx4 4 bands in 4 rows, stride 8 B (4 x 128 B)", 16, 8, 128, 0, true},
        {"dwordx4 4 bands in 4 rows, stride 4 B (4 x 64 B)", 16, 4, 64, 0, true},
        {"dwordx3 dense (768 B)", 12, 12, 192, 0, false},
        {"dwordx3 lane stride 16 B", 12, 16, 256, 0, false},
        {"dwordx4 dense, 32 of 64 lanes active", 16, 16, 256, 0, false, 32},
        {"dwordx4 dense, 16 of 64 lanes active", 16, 16, 256, 0, false, 16},
        {"dwordx2 dense, 16 of 64 lanes active", 8, 8, 128, 0, false, 16},
        {"dword   dense, 16 of 64 lanes active", 4, 4, 64, 0, false, 16},
    };
    const int iters = 400, grid = ncu * 4;
    for (size_t tbytes : {(size_t)16384, big}) {
        printf("\ntable %zu KB (%s)\n", tbytes >> 10, tbytes <= 32768 ? "fits the 32 KB vector L1" : "L2-resident");
        for (const Case &c : cases) {
            const unsigned row_bytes = 1024, nrows = (unsigned)(tbytes / row_bytes);
            auto launch = [&]() {
                switch (c.W) {
                    case 4: hipLaunchKernelGGL(gather<4>, dim3(grid), dim3(256), 0, 0, table, nrows, row_bytes, c.stride, c.band_bytes, c.misalign, iters, out, c.per_band, c.active); break;
                    case 12: hipLaunchKernelGGL(gather<12>, dim3(grid), dim3(256), 0, 0, table, nrows, row_bytes, c.stride, c.band_bytes, c.misalign, iters, out, c.per_band, c.active); break;
                    case 8: hipLaunchKernelGGL(gather<8>, dim3(grid), dim3(256), 0, 0, table, nrows, row_bytes, c.stride, c.band_bytes, c.misalign, iters, out, c.per_band, c.active); break;
                    default: hipLaunchKernelGGL(gather<16>, dim3(grid), dim3(256), 0, 0, table, nrows, row_bytes, c.stride, c.band_bytes, c.misalign, iters, out, c.per_band, c.active); break;
                }
            };
            launch(); CK(hipDeviceSynchronize());
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, 0)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
            const double instr_per_cu = (double)grid / ncu * 4 /*waves*/ * iters * 8;
            const double ns_per_instr = ms * 1e6 / instr_per_cu;
            printf("  %-58s %7.3f ms  %6.2f ns/wave-instr/CU = %5.1f clk @%.1f GHz  useful %6.1f B/clk/CU\n", c.name, ms, ns_per_instr,
                   ns_per_instr * clk_ghz, clk_ghz, 64.0 * c.stride / (ns_per_instr * clk_ghz));
        }
    }
    return 0;
}
