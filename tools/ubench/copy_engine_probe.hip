// Which engine does hipMemcpyAsync use?  (tools/experiments/README.md, round 4.)  Each variant copies a distinct size so that
// the rocprofv3 traces tell them apart: SDMA copies appear in the memory-copy trace, shader copies as __amd_rocclr_copyBuffer
// kernels.  Usage: copy_engine_probe   (run under rocprofv3 --kernel-trace --memory-copy-trace)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void busy(float *p, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = p[i % n];
    for (int k = 0; k < iters; k++) v = v * 1.0001f + 0.5f;
    p[i % n] = v;
}
int main() {
    const size_t MB = 1 << 20, N = 64 * MB;
    char *dev; CK(hipMalloc(&dev, N));
    char *reg = (char *)aligned_alloc(4096, N); for (size_t i = 0; i < N; i += 4096) reg[i] = 1;
    CK(hipHostRegister(reg, N, hipHostRegisterDefault));
    char *pin; CK(hipHostMalloc(&pin, N, hipHostMallocDefault));
    char *pag = (char *)aligned_alloc(4096, N); for (size_t i = 0; i < N; i += 4096) pag[i] = 1;
    hipStream_t K, C, D; CK(hipStreamCreateWithFlags(&K, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&C, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&D, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto kern = [&](hipStream_t s) { hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, s, (float *)dev, 1 << 20, 20000); };
    size_t sz = 2 * MB;
    auto next = [&](const char *what) { sz += 4096; printf("%-70s %zu bytes\n", what, sz); return sz; };
    // 1-3: plain copies on an idle stream
    CK(hipMemcpyAsync(reg, dev, next("D2H registered, idle stream"), hipMemcpyDeviceToHost, C)); CK(hipStreamSynchronize(C));
    CK(hipMemcpyAsync(pin, dev, next("D2H hipHostMalloc, idle stream"), hipMemcpyDeviceToHost, C)); CK(hipStreamSynchronize(C));
    CK(hipMemcpyAsync(pag, dev, next("D2H pageable, idle stream"), hipMemcpyDeviceToHost, C)); CK(hipStreamSynchronize(C));
    CK(hipMemcpyAsync(dev, reg, next("H2D registered, idle stream"), hipMemcpyHostToDevice, C)); CK(hipStreamSynchronize(C));
    // 4: same stream as a kernel
    kern(K); CK(hipMemcpyAsync(reg, dev, next("D2H registered, behind a kernel on the same stream"), hipMemcpyDeviceToHost, K)); CK(hipStreamSynchronize(K));
    kern(K); CK(hipMemcpyAsync(pin, dev, next("D2H hipHostMalloc, behind a kernel on the same stream"), hipMemcpyDeviceToHost, K)); CK(hipStreamSynchronize(K));
    // 5: other stream, waiting for the kernel's event
    kern(K); CK(hipEventRecord(ev, K)); CK(hipStreamWaitEvent(C, ev, 0));
    CK(hipMemcpyAsync(reg, dev, next("D2H registered, other stream behind hipStreamWaitEvent(kernel)"), hipMemcpyDeviceToHost, C)); CK(hipDeviceSynchronize());
    kern(K); CK(hipEventRecord(ev, K)); CK(hipStreamWaitEvent(C, ev, 0));
    CK(hipMemcpyAsync(pin, dev, next("D2H hipHostMalloc, other stream behind hipStreamWaitEvent(kernel)"), hipMemcpyDeviceToHost, C)); CK(hipDeviceSynchronize());
    kern(K); CK(hipEventRecord(ev, K)); CK(hipStreamWaitEvent(C, ev, 0));
    CK(hipMemcpyAsync(dev + 32 * MB, reg, next("H2D registered, other stream behind hipStreamWaitEvent(kernel)"), hipMemcpyHostToDevice, C)); CK(hipDeviceSynchronize());
    // 6: other stream, host waits for the event, then copies
    kern(K); CK(hipEventRecord(ev, K)); CK(hipEventSynchronize(ev));
    CK(hipMemcpyAsync(reg, dev, next("D2H registered, other stream after hipEventSynchronize"), hipMemcpyDeviceToHost, C)); CK(hipDeviceSynchronize());
    // 7: copy while another kernel is RUNNING on K (no dependency)
    kern(K); CK(hipMemcpyAsync(reg, dev + 32 * MB, next("D2H registered, idle copy stream while a kernel runs elsewhere"), hipMemcpyDeviceToHost, C)); CK(hipDeviceSynchronize());
    kern(K); CK(hipMemcpyAsync(dev + 32 * MB, reg, next("H2D registered, idle copy stream while a kernel runs elsewhere"), hipMemcpyHostToDevice, C)); CK(hipDeviceSynchronize());
    // 8: alternating directions on one stream
    for (int i = 0; i < 3; i++) {
        CK(hipMemcpyAsync(dev + 32 * MB, reg, next("H2D registered, alternating with D2H on one stream"), hipMemcpyHostToDevice, C));
        CK(hipMemcpyAsync(reg + 32 * MB, dev, next("D2H registered, alternating with H2D on one stream"), hipMemcpyDeviceToHost, C));
    }
    CK(hipDeviceSynchronize());
    // 9: D2H and H2D on two streams at once, D2H behind a kernel event
    kern(K); CK(hipEventRecord(ev, K)); CK(hipStreamWaitEvent(D, ev, 0));
    CK(hipMemcpyAsync(dev + 32 * MB, reg, next("H2D registered on C, while D2H on D waits for a kernel"), hipMemcpyHostToDevice, C));
    CK(hipMemcpyAsync(reg + 32 * MB, dev, next("D2H registered on D behind hipStreamWaitEvent(kernel), H2D in flight on C"), hipMemcpyDeviceToHost, D));
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
