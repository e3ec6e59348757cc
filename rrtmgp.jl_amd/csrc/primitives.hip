// primitives.hip — the device math forms of the column kernels, evaluated element-wise: the accuracy contract of the
// Float32 builds made measurable (include/rrtmgp_hip.h, rrtmgp_hip_eval_primitive; tests/test_primitives.py compares with
// Float64 and states the maximal error in ulp per primitive for the shipped and the IEEE-Float32 library).
// The reference computes with Julia's exp (< 1 ulp) and IEEE `/` and sqrt (docs/src/precision.md of the reference).
#include "device.h"

namespace rrtmgp {

template <typename FT>
__global__ void eval_primitive_kernel(int op, const FT *x, const FT *y, FT *out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FT a = x[i], b = y ? y[i] : FT(1);
    FT r = FT(0), e1, om1;
    switch (op) {
        case RRTMGP_PRIM_EXP_NEG: r = m_exp_neg(a); break;                 // e^-a, a >= 0
        case RRTMGP_PRIM_EXP_PAIR_E1: exp_pair(a, e1, om1); r = e1; break;  // e^-a of the two-stream pair
        case RRTMGP_PRIM_EXP_PAIR_OM1: exp_pair(a, e1, om1); r = om1; break; // 1 - e^-a
        case RRTMGP_PRIM_RCP: r = m_rcp(a); break;
        case RRTMGP_PRIM_DIV: r = m_div(a, b); break;
        case RRTMGP_PRIM_SQRT_POS: r = m_sqrt_pos(a); break;
        case RRTMGP_PRIM_IEEE_DIV: r = ieee_div(a, b); break;               // the quotients of increment_2stream
        default: break;
    }
    out[i] = r;
}

// (device buffers freed on every path: a failing copy or launch returns through the guard)
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)rr_free(p); }
};

template <typename FT>
static int eval_primitive(int device, int op, const FT *x, const FT *y, FT *out, int64_t n) {
    RR_HIP(hipSetDevice(device));
    DevBuf dx, dy, dout;
    const size_t bytes = (size_t)n * sizeof(FT);
    RR_HIP(rr_malloc(&dx.p, bytes));
    RR_HIP(rr_malloc(&dout.p, bytes));
    if (y) RR_HIP(rr_malloc(&dy.p, bytes));
    RR_HIP(hipMemcpy(dx.p, x, bytes, hipMemcpyHostToDevice));
    if (y) RR_HIP(hipMemcpy(dy.p, y, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(eval_primitive_kernel<FT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, (const FT *)dx.p,
                       (const FT *)dy.p, (FT *)dout.p, n);
    RR_HIP(hipGetLastError());
    RR_HIP(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return RRTMGP_OK;
}

}  // namespace rrtmgp

using namespace rrtmgp;

extern "C" int rrtmgp_hip_eval_primitive(int device, int32_t op, int32_t ftype, const void *x, const void *y, void *out, int64_t n) {
    RR_CHECK(x && out && n > 0 && n < (1LL << 31), "bad argument");
    RR_CHECK(op >= 0 && op <= RRTMGP_PRIM_IEEE_DIV, "unknown primitive");
    RR_CHECK(ftype == RRTMGP_F32 || ftype == RRTMGP_F64, "ftype must be 4 or 8");
    RR_CHECK(y || (op != RRTMGP_PRIM_DIV && op != RRTMGP_PRIM_IEEE_DIV), "the quotients take two operands");
    return ftype == RRTMGP_F32 ? eval_primitive<float>(device, op, (const float *)x, (const float *)y, (float *)out, n)
                               : eval_primitive<double>(device, op, (const double *)x, (const double *)y, (double *)out, n);
}
