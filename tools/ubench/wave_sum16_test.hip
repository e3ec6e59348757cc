// Unit test of wave_sum16 (device.h) on the GPU: hipcc -O3 --offload-arch=gfx950 -I../../rrtmgp.jl_amd/csrc -I../../include
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "common.h"
#include "device.h"
using namespace rrtmgp;

template <typename FT>
__global__ void k(const FT *in, FT *out) {
    FT v[16], w[4];
    for (int j = 0; j < 16; j++) v[j] = in[j * 64 + threadIdx.x] * FT(1.0001);
    wave_sum16(v, w);
    for (int i = 0; i < 4; i++) out[i * 64 + threadIdx.x] = w[i];
}

template <typename FT>
int run(const char *name) {
    std::vector<FT> h(16 * 64), o(4 * 64);
    for (int j = 0; j < 16; j++)
        for (int l = 0; l < 64; l++) h[j * 64 + l] = (FT)(std::sin(0.37 * j + 0.11 * l) + 0.01 * j);
    FT *d, *r;
    hipMalloc(&d, h.size() * sizeof(FT)); hipMalloc(&r, o.size() * sizeof(FT));
    hipMemcpy(d, h.data(), h.size() * sizeof(FT), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<FT>, dim3(1), dim3(64), 0, 0, d, r);
    hipMemcpy(o.data(), r, o.size() * sizeof(FT), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int j = 0; j < 16; j++) {
        double s = 0;
        for (int l = 0; l < 64; l++) s += (double)(h[j * 64 + l] * (FT)1.0001);
        const int i = j % 4, row = j / 4;
        for (int l = 16 * row; l < 16 * row + 16; l++) worst = std::fmax(worst, std::fabs((double)o[i * 64 + l] - s));
    }
    printf("%s: max |wave_sum16 - serial| = %.3e\n", name, worst);
    return worst < (sizeof(FT) == 4 ? 1e-4 : 1e-12) ? 0 : 1;
}

int main() { return run<float>("float") + run<double>("double"); }
