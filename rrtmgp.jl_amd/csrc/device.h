// device.h — device-side building blocks shared by the LW and SW column kernels.
//
// Execution model (gfx950): one workgroup per column, one lane per g-point
// (a wavefront owns 64 g-points = a few whole bands; with whole 16-g-point bands build_gas deals the bands to the
// wavefronts by minor-gas slot count, DevGas::lane_gpt; otherwise wavefront w owns g-points 64w..64w+63).
//
//  * prepare_column (lane = layer): everything that depends on the layer only —
//    gas table of volume mixing ratios, T / ln p interpolation indices and
//    fractions, Planck-table positions, cloud size-table positions, RH position.
//  * prepare_chunk (lane = (layer, band) pair, CH layers at a time): everything
//    that depends on (layer, band) but not on the g-point — the binary-species
//    parameter eta (indices, fractions, column mixing), minor-gas scalings, Planck
//    band sources, combined cloud and aerosol optical properties.  The reference
//    recomputes all of this for every g-point (src/optics/gas_optics.jl:129-170,
//    344-412; cloud_optics.jl:70-244; aerosol_optics.jl:141-431); hoisting it
//    evaluates the SAME expressions once per band, so results are unchanged.
//  * the g-point lanes then only gather their 8(+8) major coefficients and the
//    minor coefficients (coalesced: the tables are re-laid-out g-point-innermost),
//    combine them with the band records read from LDS, and run the vertical
//    sweeps with 3 values per level in the sweep scratch.
#pragma once

#include "common.h"
#include "variants.h"

namespace rrtmgp {

constexpr int CH = 16;  // layers per preparation chunk
// ... of the kernel variant CA (clouds | aerosols << 1, -1 = run-time flags): the aerosol records would push the LDS
// of a workgroup past a quarter of the CU's 160 KB (3 resident workgroups instead of 4), so those variants prepare
// 8 layers at a time
__host__ __device__ constexpr int chunk_layers(int ca, bool diag = false) {
    return ca >= 2 || (diag && RR_DIAG_MIN_WAVES >= 4) ? CH / 2 : CH;
}
// ... of the HALF instances of the main two-stream kernels (shorter chunks where that admits one more workgroup per CU)
template <typename FT>
__host__ __device__ constexpr int half_chunk_layers() { return sizeof(FT) == 8 ? RR_F64_HALF_CHUNK : CH / 2; }


// ---- numerics (src/Numerics.jl:24-63), all of the working precision --------------
template <typename FT> struct Num;
template <> struct Num<float> {
    static __device__ __forceinline__ float eps() { return 1.1920928955078125e-07f; }
    static __device__ __forceinline__ float pi() { return 3.14159265358979323846f; }
};
template <> struct Num<double> {
    static __device__ __forceinline__ double eps() { return 2.220446049250313e-16; }
    static __device__ __forceinline__ double pi() { return 3.14159265358979323846; }
};

// Float32 device math.  The library computes with IEEE-accurate forms by default (hard part 5: the
// reference runs Julia's exp (< 1 ulp) and IEEE `/` and sqrt, docs/src/precision.md): correctly rounded quotients,
// reciprocals and square roots, e^-x to 1.2 ulp, all written out below around v_rcp_f32 / v_rsq_f32 / v_exp_f32 with FMA
// residual steps (no libm call and no compiler-expanded division in the g-point loops); tests/test_primitives.py measures
// every form on the GPU against Float64.  -DRR_FAST_F32 (`make fast` -> libhip_rrtmgp_fast.so, with
// -fno-hip-fp32-correctly-rounded-divide-sqrt) builds the raw-instruction forms instead (__expf, v_rcp_f32, v_sqrt_f32:
// 1-14 ulp), an opt-in that bench.py reports as variants.fast_f32.
// Float64 uses hand-written forms of <= 2 ulp (exp_pair, m_rcp, m_div below) in both builds; -DRR_LIBM_F64 (an
// experiment) swaps libm and the compiler's division back in.
#ifndef RR_FAST_F32
__device__ __forceinline__ float m_exp(float x) { return expf(x); }
#else
__device__ __forceinline__ float m_exp(float x) { return __expf(x); }
#endif
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
__device__ __forceinline__ float m_expm1(float x) { return expm1f(x); }
__device__ __forceinline__ double m_expm1(double x) { return expm1(x); }
__device__ __forceinline__ float m_log(float x) { return logf(x); }
__device__ __forceinline__ double m_log(double x) { return log(x); }
__device__ __forceinline__ float m_pow(float x, float y) { return powf(x, y); }
__device__ __forceinline__ double m_pow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_sin(float x) { return sinf(x); }
__device__ __forceinline__ double m_sin(double x) { return sin(x); }
__device__ __forceinline__ float m_cos(float x) { return cosf(x); }
__device__ __forceinline__ double m_cos(double x) { return cos(x); }
// Division and reciprocal of well-scaled Float32 quantities in the per-g-point loops (optical
// depths, albedos, two-stream denominators: never denormal or near overflow): v_rcp_f32 (1 ulp)
// times the numerator (the -DRR_FAST_F32 forms; the default forms add one FMA residual step, below).
// Float64: v_rcp_f64 refined by two Newton steps (5 instructions), and for a quotient one residual correction on top
// (8 instructions; the compiler's IEEE expansion with its scaling and fix-up steps is 13): at most 1 ulp for the well-scaled
// operands of the g-point loops, far inside the Float64 parity budget (1e-11 relative).  A zero divisor gives NaN / inf
// exactly where `1.0 / x` gives inf, and every such call site selects its result away (tau <= 0).
#ifdef RR_LIBM_F64
__device__ __forceinline__ double m_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ double m_div(double a, double b) { return a / b; }
#else
__device__ __forceinline__ double m_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double m_div(double a, double b) {
    const double r = m_rcp(b), q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
#endif
#ifndef RR_FAST_F32
// IEEE-accurate Float32 forms of the per-g-point loops, written out (round 5) instead of `/`, sqrtf and libm: the compiler's
// correctly rounded division is 10 instructions between two s_setreg (denormal mode on and off again), sqrtf likewise, and
// libm's expf / expm1f are ~15 / ~45.  The operands here are well scaled (optical depths, albedos, two-stream denominators:
// never denormal, never near overflow), which is all these forms need:
//  * quotient / reciprocal: v_rcp_f32 (1 ulp) + ONE residual correction with FMAs — the rounded result of q + (a - b q) rc,
//    where the residual is exact: correctly rounded except for ties closer than 2^-23 ulp (the form `ieee_div` has always used);
//  * square root: v_rsq_f32 + one Heron step on the exact residual x - s^2, same argument: correctly rounded;
//  * e^-y: see exp_neg_acc below (<= 1 ulp).
// tests/test_primitives.py measures all of them on the GPU against Float64 (rrtmgp_hip_eval_primitive).
__device__ __forceinline__ float m_rcp(float x) {
    const float rc = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, rc, 1.0f), rc, rc);
}
__device__ __forceinline__ float m_div(float a, float b) {
    const float rc = __builtin_amdgcn_rcpf(b), q = a * rc;
    return fmaf(fmaf(-b, q, a), rc, q);
}
__device__ __forceinline__ float m_sqrt_pos(float x) {   // x in [k_min, O(10)]: positive, normal
    const float r = __builtin_amdgcn_rsqf(x), s = x * r;
    return fmaf(fmaf(-s, s, x), 0.5f * r, s);
}
#else
__device__ __forceinline__ float m_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float m_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float m_sqrt_pos(float x) { return __builtin_amdgcn_sqrtf(x); }  // x in [k_min, O(10)]
#endif
__device__ __forceinline__ double m_sqrt_pos(double x) { return sqrt(x); }
// correctly rounded division whatever the translation unit's division flag (increment_2stream)
// (Float32: v_rcp_f32 + one FMA residual correction: within 0.5 ulp + 2^-40 for normal operands.  `/` and __fdiv_rn are the
// 2.5-ulp form in a translation unit built with -fno-hip-fp32-correctly-rounded-divide-sqrt.)
__device__ __forceinline__ float ieee_div(float a, float b) {
    const float rc = __builtin_amdgcn_rcpf(b), q = a * rc;
    return fmaf(fmaf(-b, q, a), rc, q);
}
// Float64: NOT the IEEE quotient — v_rcp_f64 + two Newton steps + one residual correction, <= 1 ulp for normal, well-scaled
// operands (wrong for denormal or near-overflow divisors, which increment_2stream's max(eps, .) denominators never are)
__device__ __forceinline__ double ieee_div(double a, double b) { return m_div(a, b); }
template <typename FT> __device__ __forceinline__ FT m_max(FT a, FT b) { return a > b ? a : b; }
template <typename FT> __device__ __forceinline__ FT m_min(FT a, FT b) { return a < b ? a : b; }
template <typename FT> __device__ __forceinline__ FT m_abs(FT a) { return a < FT(0) ? -a : a; }

// e1 = exp(-x) and om1 = 1 - exp(-x) for x >= 0: the pair the two-stream coefficients need
// (e1 = exp(-tau k), om1 = -expm1(-tau k); longwave_2stream.jl:167-168, shortwave_2stream.jl:204-209).
// Float64: exp(-x) and 1 - exp(-x) from ONE argument reduction (libm's exp + expm1 are two, ~75 FP64 instructions
// together; this is ~25).  x = n ln2 + r with |r| <= ln2 / 2, p = expm1(-r) by a degree-13 polynomial (next term
// 0.347^14 / 14! = 4e-18), e^-x = 2^-n (1 + p); 1 - e^-x is -p itself when n = 0 (no cancellation for small x, which is
// what expm1 is for) and 1 - e^-x otherwise (e^-x <= 0.71 there).  Error <= 2 ulp on both results for x >= 0; x beyond
// ~745 gives exactly (0, 1) like libm.  -DRR_LIBM_F64 builds keep libm.
// (1 - e^-r) for |r| <= ln2 / 2, and the reduction x = n ln2 + r of a non-negative x
__device__ __forceinline__ double exp_reduce(double x, int &ni) {
    // an overflowed optical depth (+inf, or beyond 2^52 ln 2 where the reduction loses r) must saturate to an opaque layer,
    // (0, 1), as libm does, not turn into NaN: everything from 1500 on gives exactly that (2^-2000 underflows to 0)
    x = x > 1500.0 ? 1500.0 : x;   // (compare + select, not fmin: a NaN optical depth stays NaN and surfaces in the fluxes)
    const double n = __builtin_rint(x * 1.4426950408889634074);   // x / ln 2
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);  // ln2_hi (the low 21 bits are zero: n * ln2_hi is exact)
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);         // ln2_lo
    // q = (1 - e^-r) / r = 1 - r/2 + r^2/6 - ... , Horner
    double q = 1.0 / 87178291200.0;                                // 1/14!
    q = __builtin_fma(q, -r, 1.0 / 6227020800.0);
    q = __builtin_fma(q, -r, 1.0 / 479001600.0);
    q = __builtin_fma(q, -r, 1.0 / 39916800.0);
    q = __builtin_fma(q, -r, 1.0 / 3628800.0);
    q = __builtin_fma(q, -r, 1.0 / 362880.0);
    q = __builtin_fma(q, -r, 1.0 / 40320.0);
    q = __builtin_fma(q, -r, 1.0 / 5040.0);
    q = __builtin_fma(q, -r, 1.0 / 720.0);
    q = __builtin_fma(q, -r, 1.0 / 120.0);
    q = __builtin_fma(q, -r, 1.0 / 24.0);
    q = __builtin_fma(q, -r, 1.0 / 6.0);
    q = __builtin_fma(q, -r, 0.5);
    q = __builtin_fma(q, -r, 1.0);
    ni = (int)__builtin_fmin(n, 2000.0);
    return r * q;                                                  // (may be slightly negative: r in [-ln2/2, ln2/2])
}
__device__ __forceinline__ void exp_pair(double x, double &e1, double &om1) {
#ifdef RR_LIBM_F64
    e1 = exp(-x);
    om1 = -expm1(-x);
#else
    int ni;
    const double m = exp_reduce(x, ni);
    e1 = __builtin_ldexp(1.0 - m, -ni);                            // 2^-n e^-r
    om1 = ni == 0 ? m : 1.0 - e1;
#endif
}
// e^-y for y >= 0 (transmissivities, the direct beam): the same reduction in Float64, the fast / libm exp in Float32
__device__ __forceinline__ double m_exp_neg(double y) {
#ifdef RR_LIBM_F64
    return exp(-y);
#else
    int ni;
    const double m = exp_reduce(y, ni);
    return __builtin_ldexp(1.0 - m, -ni);
#endif
}
// e^-y for y >= 0 to <= 1 ulp (Julia's exp(::Float32): < 1 ulp) around v_exp_f32, which is 2^p to 1 ulp for any p — what
// __expf loses is the ARGUMENT: p = -y log2(e) rounded to Float32 is off by up to half an ulp of p, i.e. |p| 2^-24 ln 2 in the
// result (10 ulp at y = 20).  Here the product is kept as p_hi + p_lo (one FMA recovers the rounding error of the product, a
// second adds y times the low word of log2 e) and e^-y = 2^p_hi (1 + p_lo ln 2): 6 instructions.  y beyond 200 saturates to
// 0 like expf (2^-288 underflows); the clamp is a compare + select so that a NaN optical depth stays NaN.
__device__ __forceinline__ float exp_neg_acc(float y) {
    y = y > 200.0f ? 200.0f : y;
    const float L2E_HI = 1.4426950216293335f, L2E_LO = 1.925963033500011e-08f;   // log2(e) = hi + lo
    const float p = -y * L2E_HI;
    float pl = fmaf(-y, L2E_HI, -p);
    pl = fmaf(-y, L2E_LO, pl);
    const float e = __builtin_amdgcn_exp2f(p);
    return fmaf(e, pl * 0.6931471824645996f, e);
}
#ifndef RR_FAST_F32
__device__ __forceinline__ float m_exp_neg(float y) { return exp_neg_acc(y); }
#else
__device__ __forceinline__ float m_exp_neg(float y) { return m_exp(-y); }
#endif
__device__ __forceinline__ void exp_pair(float x, float &e1, float &om1) {
#ifndef RR_FAST_F32
    e1 = exp_neg_acc(x);
#else
    e1 = __expf(-x);
#endif
    // x <= 1/2: x * sum_{n=0..7} (-x)^n / (n+1)!  (truncation < 2^-27 relative); above: 1 - e1 with e1 < 0.61
    float p = -1.0f / 40320.0f;
    p = fmaf(p, x, 1.0f / 5040.0f);
    p = fmaf(p, x, -1.0f / 720.0f);
    p = fmaf(p, x, 1.0f / 120.0f);
    p = fmaf(p, x, -1.0f / 24.0f);
    p = fmaf(p, x, 1.0f / 6.0f);
    p = fmaf(p, x, -0.5f);
    p = fmaf(p, x, 1.0f);
    om1 = x > 0.5f ? 1.0f - e1 : x * p;
}

template <typename FT> __device__ __forceinline__ FT k_min() { return m_sqrt(Num<FT>::eps()); }
template <typename FT> __device__ __forceinline__ FT tau_thresh() { return m_sqrt(m_sqrt(Num<FT>::eps())); }
template <typename FT> __device__ __forceinline__ FT resonance_window() { return m_sqrt(Num<FT>::eps()); }
template <typename FT> __device__ __forceinline__ FT mu0_min() { return Num<FT>::eps(); }

// ---- McICA counter-based stream (spec: include/rrtmgp_hip.h) -----------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t mcica_key(uint64_t seed, int64_t gcol, int64_t igpt, int is_sw) {
    const uint64_t G = 0x9e3779b97f4a7c15ULL;
    uint64_t k = mix64(seed + G * (uint64_t)gcol);
    return mix64(k ^ ((uint64_t)igpt | ((uint64_t)(is_sw ? 1 : 0) << 32)));
}
__host__ __device__ __forceinline__ double mcica_draw(uint64_t key, int draw) {
    const uint64_t G = 0x9e3779b97f4a7c15ULL;
    uint64_t k = mix64(key + G * ((uint64_t)draw + 1));
    return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

// ---- wavefront (64-lane) sum over g-points -------------------------------------------
// DPP butterfly inside each 16-lane row, then row_bcast15 / row_bcast31 across rows;
// the total lands in lane 63.  Fixed order => bit-reproducible broadband fluxes.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    // rows disabled by ROW_MASK read 0 (the additive identity), so `v += dpp_mov(v)` folds to one v_add_f32_dpp
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF,
                                                                 ROW_MASK == 0xF));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <typename FT>
__device__ __forceinline__ FT wave_sum_to_lane63(FT v) {
    v += dpp_mov<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_mov<0x140, 0xF>(v);  // row_mirror      -> every lane holds its row's sum
    v += dpp_mov<0x142, 0xA>(v);  // row_bcast15 into rows 1, 3
    v += dpp_mov<0x143, 0xC>(v);  // row_bcast31 into rows 2, 3 -> lane 63 holds the wave sum
    return v;
}

// Sum over each DPP row (16 lanes = one band of rrtmgp-data v1.9); every lane of the row gets it.
template <typename FT>
__device__ __forceinline__ FT row_sum(FT v) {
    v += dpp_mov<0xB1, 0xF>(v);
    v += dpp_mov<0x4E, 0xF>(v);
    v += dpp_mov<0x141, 0xF>(v);
    v += dpp_mov<0x140, 0xF>(v);
    return v;
}
// g-point sum of the layer loop into a per-wave LDS accumulator that was zeroed at the start of the column: the four
// in-row DPP steps, then lanes 15 / 31 / 47 / 63 add their row's sum with ONE ds_add (the LDS serialises the four lanes
// in a fixed order: reproducible).  Saves the two cross-row DPP steps (v_mov_dpp + v_add + hazard nops each) of
// wave_sum_to_lane63: ~8 issue slots per sum in a loop that is VALU-issue bound.
template <typename FT>
__device__ __forceinline__ void wave_add_to(FT *slot, FT v) {
    const FT rs = row_sum(v);
    if ((threadIdx.x & 15) == 15) (void)__hip_atomic_fetch_add(slot, rs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The flux accumulators are kept per "segment": a whole wave (written by lane 63), or a
// 16-lane row (written by its lane 15) when per-band fluxes are requested.
template <bool BAND, typename FT>
__device__ __forceinline__ FT seg_sum(FT v) {
    return BAND ? row_sum(v) : wave_sum_to_lane63(v);
}

// ---- 16 wavefront sums at once ----------------------------------------------------------
// Summing ONE value over the 64 lanes costs 6 dependent DPP adds (plus their hazard nops).  16
// values are cheaper together: v_permlane32_swap / v_permlane16_swap exchange half-wavefronts and
// odd/even rows of a register PAIR, so one swap + one add halves two values at a time
// (16 -> 8 -> 4 registers, each row then holding a different value), and only 4 registers need
// the 4 in-row DPP steps.  40 instructions instead of 96, and independent chains (no nops).
// (Inline asm: with ROCm 7.2 the __builtin_amdgcn_permlane{16,32}_swap intrinsics lose their second
// result for 32-bit values — `v_add v1, v1, v1` after the swap; tools/ubench/wave_sum16_test.hip.
// The s_nop's: the operands may have been written by the VALU instruction just before and are read
// by the one just after; without them the sums are wrong — the compiler's hazard recogniser does not
// see through inline asm.  The swaps of one stage share one pair of nops.)
// x[i] <-> y[i] across the two half-wavefronts: x[i] = [x_lo, y_lo], y[i] = [x_hi, y_hi]
__device__ __forceinline__ void lane_swap32x4(unsigned (&x)[4], unsigned (&y)[4]) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\t"
                 "v_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\ts_nop 1"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
}
// odd rows of x[i] <-> even rows of y[i]: x[i] = [x_r0, y_r0, x_r2, y_r2], y[i] = [x_r1, y_r1, x_r3, y_r3]
__device__ __forceinline__ void lane_swap16x4(unsigned (&x)[4], unsigned (&y)[4]) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                 "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
}
template <bool HALVES>
__device__ __forceinline__ void lane_swap_x4(float (&x)[4], float (&y)[4]) {
    unsigned ux[4], uy[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { ux[i] = __builtin_bit_cast(unsigned, x[i]); uy[i] = __builtin_bit_cast(unsigned, y[i]); }
    if (HALVES) lane_swap32x4(ux, uy); else lane_swap16x4(ux, uy);
#pragma unroll
    for (int i = 0; i < 4; i++) { x[i] = __builtin_bit_cast(float, ux[i]); y[i] = __builtin_bit_cast(float, uy[i]); }
}
template <bool HALVES>
__device__ __forceinline__ void lane_swap_x4(double (&x)[4], double (&y)[4]) {
    unsigned xl[4], xh[4], yl[4], yh[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned long long a = __builtin_bit_cast(unsigned long long, x[i]), b = __builtin_bit_cast(unsigned long long, y[i]);
        xl[i] = (unsigned)a; xh[i] = (unsigned)(a >> 32); yl[i] = (unsigned)b; yh[i] = (unsigned)(b >> 32);
    }
    if (HALVES) { lane_swap32x4(xl, yl); lane_swap32x4(xh, yh); } else { lane_swap16x4(xl, yl); lane_swap16x4(xh, yh); }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        x[i] = __builtin_bit_cast(double, ((unsigned long long)xh[i] << 32) | xl[i]);
        y[i] = __builtin_bit_cast(double, ((unsigned long long)yh[i] << 32) | yl[i]);
    }
}
// On return every lane of DPP row r (lanes 16r .. 16r+15) holds in w[i] the wavefront sum of v[i + 4 r].
template <typename FT>
__device__ __forceinline__ void wave_sum16(const FT (&v)[16], FT (&w)[4]) {
    FT a[4], b[4], u[8];
#pragma unroll
    for (int h = 0; h < 2; h++) {  // pairs (v[i], v[i + 8]), four at a time
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] = v[4 * h + i]; b[i] = v[4 * h + i + 8]; }
        lane_swap_x4<true>(a, b);
#pragma unroll
        for (int i = 0; i < 4; i++) u[4 * h + i] = a[i] + b[i];  // lanes 0-31: v[i] over both halves; 32-63: v[i + 8]
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = u[i]; b[i] = u[i + 4]; }
    lane_swap_x4<false>(a, b);
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = a[i] + b[i];  // rows 0..3: v[i], v[i + 4], v[i + 8], v[i + 12] over the four rows
#pragma unroll
    for (int i = 0; i < 4; i++) w[i] = row_sum(w[i]);
}

// ---- dimensions + LDS carve ----------------------------------------------------------
struct ColDims {
    int nlay, nlev, ngas1 /* rows of the gas table */, nwaves, nbnd;
    int lw, twostream, has_cld, has_aero, n_acc /* accumulated flux components per level */;
    int max_int; /* minor-gas scaling slots per layer row (the larger region's; a multiple of MINOR_PAIR) */
    int nseg;    /* flux accumulator segments per block: nwaves, or 4 per wave with per-band fluxes */
    int diag;    /* clear-sky fluxes are carried next to the all-sky ones (n_acc doubles) */
    /* sizes of the small lookup tables mirrored in LDS (TabCache) */
    int n_t_ref, n_p_ref, n_t_plnk, n_gases_ref, nint0, nint1, nslot0, nslot1;
};

// 4 values read with one ds_read_b128 (Float32) / two (Float64)
template <typename FT>
struct alignas(4 * sizeof(FT)) V4 {
    FT x, y, z, w;
};

constexpr int NBMAX = 16;  // bands per lookup the chunk records are laid out for (v1.9: 16 LW / 14 SW)

// Everything the kernels keep per layer, one record per layer (array of structs: a single LDS
// base address, constant field offsets).
template <typename FT>
struct alignas(32) LayerRec {
    FT fP, ray_fac;            // read together with idx by the g-point lanes; ray_fac = (h2o + 1) col_dry (SW Rayleigh)
    int idx;                   // jT | jP << 8 | tropo << 16 (0-based lower T index, lower p plane, 0 = lower atm.)
    FT fT;
    FT col_dry, h2o;
    int aero_mask;
    int liq_loc, ice_loc;
    FT dens_fact, dry_fact, cld_frac, path_liq;
    FT path_ice, liq_fac, ice_fac, rh_f;
    int rh_loc, pl_lay_loc;
    FT pl_lay_f, aod_t, aod_ts;
    FT pad_;  // aod_*: per-layer (tau, tau*ssa) of the 550 nm band (SW with aerosols)
};
template <typename FT>
struct LevelRec {
    FT f;     // Planck-table position of t_lev (LW)
    int loc;
};

// Band-level records of one chunk of CH layers, fixed strides (record r = kk * NBMAX + band), so every
// array sits at a compile-time LDS offset.
template <typename FT, int CHK = CH>
struct alignas(32) ChunkFixed {
    V4<FT> wgt[CHK * NBMAX];  // (1-fe1)(1-fT), fe1 (1-fT), (1-fe2) fT, fe2 fT: the (eta, T) weights of interp2d / interp3d
    V4<FT> amp[CHK * NBMAX];  // col_dry * (cm1 (1-fP), cm1 fP, cm2 (1-fP), cm2 fP): column amounts x pressure weights
    V4<FT> cld[CHK * NBMAX];  // cloud (tau, tau*ssa, tau*ssa*g, -) or (absorption tau, -, -, -)
    FT Blev[(CHK + 1) * NBMAX];
    int je[CHK * NBMAX];      // je1 | je2 << 8
    FT pad[NBMAX];           // keeps the tail 32-byte aligned
    // the tail is only allocated as far as it is used (carve_shared):
    FT Blay[CHK * NBMAX];     // layer Planck sources: no-scattering LW only
    V4<FT> aer[CHK * NBMAX];  // aerosol (tau, tau*ssa, tau*ssa*g, -): only with an aerosol lookup
};

template <typename FT, int CHK = CH>
struct ColShared {
    ChunkFixed<FT, CHK> *ch;  // LDS offset 0
    LayerRec<FT> *lay;   // constant offset
    LevelRec<FT> *lev;   // [nlev]
    FT *vmr;             // [ngas1][nlay]; row 0 is 1 (get_vmr ig == 0, VolumeMixingRatios.jl:97-99)
    int mscale_row;      // = max_int
    FT *mscale;          // [CH][max_int] minor-gas scalings of the current chunk, slot-contiguous per layer (pairs of slots)
    FT *acc;             // [nseg][nlev][n_acc]
    int *misc;           // [0..nwaves): cloudy g-points per wave; [nwaves]: pl_sfc_loc; [+1]: cld start; [+2]: cld finish; [+3]: next column;
                         // [+4]: highest layer with a scattering particle, cloud or aerosol (-1: none)
    FT *miscf;           // [0]: pl_sfc_f
    uint64_t *mask;      // McICA masks of columns with more than 128 layers and clouds: [word][256 lanes], else unused
    // TabCache: the small lookup tables the preparation steps index with data-dependent positions, copied once per
    // workgroup (not per column) so that those dependent reads are LDS round trips instead of L2 ones
    FT *tab_t_ref, *tab_ln_p_ref, *tab_t_planck, *tab_eta_half;
    FT *tab_vmr_gm;  // VmrGM: the well-mixed vector, [ngas1] with entry 0 = 1 (it does not depend on the column)
    int *tab_key_species, *tab_gasdata[2], *tab_slot_int[2];
};

template <typename T>
__host__ __device__ inline T *carve(char *&p, size_t n) {
    T *r = reinterpret_cast<T *>(p);
    p += (n * sizeof(T) + 31) & ~size_t(31);
    return r;
}

template <typename FT, int CHK>
__host__ __device__ inline size_t carve_shared(ColShared<FT, CHK> &s, char *base, const ColDims &d) {
    char *p = base;
    s.ch = carve<ChunkFixed<FT, CHK>>(p, 1);
    // drop the unused tail of the chunk record: aer, and Blay when neither is needed (both multiples of 32 bytes)
    if (!d.has_aero) {
        p -= sizeof(V4<FT>) * CHK * NBMAX;
        if (!(d.lw && !d.twostream)) p -= sizeof(FT) * CHK * NBMAX;
    }
    s.lay = carve<LayerRec<FT>>(p, d.nlay);
    s.lev = carve<LevelRec<FT>>(p, d.nlev);
    s.vmr = carve<FT>(p, (size_t)d.ngas1 * d.nlay);
    s.mscale_row = d.max_int;
    s.mscale = carve<FT>(p, (size_t)(d.max_int > 0 ? d.max_int : MINOR_PAIR) * CHK);
    s.acc = carve<FT>(p, (size_t)d.nseg * d.nlev * d.n_acc);
    s.misc = carve<int>(p, d.nwaves + 5);
    s.miscf = carve<FT>(p, 4);
    s.mask = carve<uint64_t>(p, d.has_cld && d.nlay > 128 ? (size_t)((d.nlay + 63) / 64 + 1) * 256 : 0);
    s.tab_t_ref = carve<FT>(p, d.n_t_ref);
    s.tab_ln_p_ref = carve<FT>(p, d.n_p_ref);
    s.tab_t_planck = carve<FT>(p, d.lw ? d.n_t_plnk : 0);
    s.tab_eta_half = carve<FT>(p, (size_t)2 * d.nbnd * d.n_t_ref);
    s.tab_vmr_gm = carve<FT>(p, d.ngas1);
    s.tab_key_species = carve<int>(p, 4 * d.nbnd);
    s.tab_gasdata[0] = carve<int>(p, 4 * (d.nint0 > 0 ? d.nint0 : 1));
    s.tab_gasdata[1] = carve<int>(p, 4 * (d.nint1 > 0 ? d.nint1 : 1));
    s.tab_slot_int[0] = carve<int>(p, d.nslot0 > 0 ? d.nslot0 : 1);
    s.tab_slot_int[1] = carve<int>(p, d.nslot1 > 0 ? d.nslot1 : 1);
    return (size_t)(p - base);
}

// Fill the TabCache (once per workgroup); cold_gas_view() builds the lookup view whose small-table pointers are these copies.
template <typename FT, int CHK>
__device__ inline void cache_small_tables(const ColShared<FT, CHK> &sh, const ColDims &d, const DevGas<FT> &lk,
                                                const DevState<FT> &as) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if (as.vmr_kind == RRTMGP_VMR_GM)
        for (int ig = tid; ig < d.ngas1; ig += nt) sh.tab_vmr_gm[ig] = ig == 0 ? FT(1) : ig <= as.ngas ? as.vmr[ig - 1] : FT(0);
    for (int i = tid; i < d.n_t_ref; i += nt) sh.tab_t_ref[i] = lk.t_ref[i];
    for (int i = tid; i < d.n_p_ref; i += nt) sh.tab_ln_p_ref[i] = lk.ln_p_ref[i];
    if (d.lw) for (int i = tid; i < d.n_t_plnk; i += nt) sh.tab_t_planck[i] = lk.t_planck[i];
    for (int i = tid; i < 2 * d.nbnd * d.n_t_ref; i += nt) sh.tab_eta_half[i] = lk.eta_half[i];
    for (int i = tid; i < 4 * d.nbnd; i += nt) sh.tab_key_species[i] = lk.key_species[i];
    for (int i = tid; i < 4 * d.nint0; i += nt) sh.tab_gasdata[0][i] = lk.m_gasdata[0][i];
    for (int i = tid; i < 4 * d.nint1; i += nt) sh.tab_gasdata[1][i] = lk.m_gasdata[1][i];
    for (int i = tid; i < d.nslot0; i += nt) sh.tab_slot_int[0][i] = lk.m_slot_int[0][i];
    for (int i = tid; i < d.nslot1; i += nt) sh.tab_slot_int[1][i] = lk.m_slot_int[1][i];
    __syncthreads();
}

// ---- cold kernel arguments ------------------------------------------------------------------------------------------
// The column kernels take ONE by-value argument struct (LwArgs / SwArgs: ~100 pointers and sizes).  Read as `a.member` the
// compiler loads every member it ever needs at kernel entry and keeps it in SGPRs for the whole kernel: ~100 live scalar
// registers, 260-340 of them spilled to VGPR lanes, and the preparation phases (which use most of the pointers: state,
// cloud, aerosol, flux arrays) spent 30 % of their instructions on v_readlane_b32 + hazard nops getting them back
// (round-5 review).  Only the layer loops' own operands (table arena, strides, sweep scratch) are worth a register for the
// length of a column.  Everything else is read WHERE IT IS USED, straight from the kernarg segment with s_load (constant
// address space, scalar cache), at an offset made opaque so that the load is neither hoisted out of the column loop nor
// kept alive across the layer loops:
//   const DevState<FT> as = cold_state<FT>(offsetof(LwArgs<FT>, as));     // inside the column loop, next to its uses
// A pointer that was loaded from memory has no known address space (accesses through it would be flat_load / flat_store
// with 64-bit VGPR addresses): global_ptr() re-types it as a global pointer, which is what the by-value path gets from
// clang's kernel-argument coercion.
template <typename T>
__device__ __forceinline__ T kernarg_load(unsigned offset) {
    static_assert(sizeof(T) % 4 == 0, "whole dwords");
    asm volatile("" : "+s"(offset));   // opaque: the loads below stay behind this point of the program
    typedef __attribute__((address_space(4))) const unsigned cu32;
    cu32 *p = (cu32 *)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() + offset);
    unsigned w[sizeof(T) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = p[i];
    T v;
    __builtin_memcpy(&v, w, sizeof(T));
    return v;
}
template <typename T>
__device__ __forceinline__ T *global_ptr(T *p) {
    return (T *)(__attribute__((address_space(1))) T *)(unsigned long long)p;
}
template <typename T>
__device__ __forceinline__ T *kernarg_ptr(unsigned offset) { return global_ptr(kernarg_load<T *>(offset)); }

template <typename FT>
__device__ __forceinline__ DevState<FT> cold_state(unsigned offset) {
    DevState<FT> s = kernarg_load<DevState<FT>>(offset);
    s.layerdata = global_ptr(s.layerdata); s.t_lev = global_ptr(s.t_lev); s.t_sfc = global_ptr(s.t_sfc);
    s.vmr_h2o = global_ptr(s.vmr_h2o); s.vmr_o3 = global_ptr(s.vmr_o3); s.vmr = global_ptr(s.vmr);
    s.cld_r_eff_liq = global_ptr(s.cld_r_eff_liq); s.cld_r_eff_ice = global_ptr(s.cld_r_eff_ice);
    s.cld_path_liq = global_ptr(s.cld_path_liq); s.cld_path_ice = global_ptr(s.cld_path_ice);
    s.cld_frac = global_ptr(s.cld_frac); s.cld_cover = global_ptr(s.cld_cover);
    s.aero_size = global_ptr(s.aero_size); s.aero_mass = global_ptr(s.aero_mass);
    s.aod_sw_ext = global_ptr(s.aod_sw_ext); s.aod_sw_sca = global_ptr(s.aod_sw_sca);
    return s;
}
template <typename FT>
__device__ __forceinline__ DevCld<FT> cold_cld(unsigned offset) {
    DevCld<FT> c = kernarg_load<DevCld<FT>>(offset);
    c.liqdata = global_ptr(c.liqdata); c.icedata = global_ptr(c.icedata);
    return c;
}
template <typename FT>
__device__ __forceinline__ DevAero<FT> cold_aero(unsigned offset) {
    DevAero<FT> a = kernarg_load<DevAero<FT>>(offset);
    a.size_bin_limits = global_ptr(a.size_bin_limits); a.rh_levels = global_ptr(a.rh_levels); a.dust = global_ptr(a.dust);
    a.sea_salt = global_ptr(a.sea_salt); a.sulfate = global_ptr(a.sulfate); a.black_carbon_rh = global_ptr(a.black_carbon_rh);
    a.black_carbon = global_ptr(a.black_carbon); a.organic_carbon_rh = global_ptr(a.organic_carbon_rh);
    a.organic_carbon = global_ptr(a.organic_carbon);
    return a;
}
template <typename FT>
__device__ __forceinline__ DevFlux<FT> cold_flux(unsigned offset) {
    DevFlux<FT> f = kernarg_load<DevFlux<FT>>(offset);
    f.up = global_ptr(f.up); f.dn = global_ptr(f.dn); f.net = global_ptr(f.net); f.dir = global_ptr(f.dir);
    f.metric = global_ptr(f.metric);
    f.band_up = global_ptr(f.band_up); f.band_dn = global_ptr(f.band_dn); f.band_net = global_ptr(f.band_net);
    f.clear_up = global_ptr(f.clear_up); f.clear_dn = global_ptr(f.clear_dn); f.clear_net = global_ptr(f.clear_net);
    f.clear_dir = global_ptr(f.clear_dir);
    return f;
}
// The lookup as the preparation steps see it: sizes and the Planck table from the kernarg segment, the small tables they
// index with data-dependent positions from their LDS copies (TabCache, filled once per workgroup by cache_small_tables).
template <typename FT, int CHK>
__device__ __forceinline__ DevGas<FT> gas_view(DevGas<FT> v, const ColShared<FT, CHK> &sh, const ColDims &d) {
    v.t_ref = sh.tab_t_ref; v.ln_p_ref = sh.tab_ln_p_ref; v.eta_half = sh.tab_eta_half;
    if (d.lw) v.t_planck = sh.tab_t_planck;
    v.key_species = sh.tab_key_species;
    v.m_gasdata[0] = sh.tab_gasdata[0]; v.m_gasdata[1] = sh.tab_gasdata[1];
    v.m_slot_int[0] = sh.tab_slot_int[0]; v.m_slot_int[1] = sh.tab_slot_int[1];
    return v;
}
template <typename FT, int CHK>
__device__ __forceinline__ DevGas<FT> cold_gas_view(unsigned offset, const ColShared<FT, CHK> &sh, const ColDims &d) {
    DevGas<FT> v = kernarg_load<DevGas<FT>>(offset);
    v.tot_planck = global_ptr(v.tot_planck); v.band_row_lo = global_ptr(v.band_row_lo);
    return gas_view(v, sh, d);
}

// ---- g-point independent column preparation (lane = layer) --------------------------
// loc_lower for uniform grids, src/optics/optics_utils.jl:7-14 (returns 0-based index)
template <typename FT>
__device__ __forceinline__ int loc_lower_eq0(FT xi, FT dx, int n, const FT *x) {
    if (xi <= x[0]) return 0;
    if (xi >= x[n - 1]) return n - 2;
    int j = (int)((xi - x[0]) / dx);
    return j < n - 2 ? j : n - 2;
}

// interp1d_equispaced position (optics_utils.jl:34-44) as (loc0, factor); the
// out-of-range returns y[1] / y[end] are encoded as (0, 0) and (n-2, 1).
template <typename FT>
__device__ __forceinline__ void planck_pos(FT xi, const FT *x, int n, int &loc, FT &f) {
    if (xi < x[0]) { loc = 0; f = FT(0); return; }
    if (xi > x[n - 1]) { loc = n - 2; f = FT(1); return; }
    FT dx = x[1] - x[0];
    loc = loc_lower_eq0(xi, dx, n, x);
    f = (xi - x[loc]) / dx;
}

// interp1d_loc_factor for non-uniform grids (optics_utils.jl:21-27,51-62), 0-based
template <typename FT>
__device__ __forceinline__ void loc_factor_gen(FT xi, const FT *x, int n, int &loc, FT &f) {
    if (xi < x[0]) { loc = 0; f = FT(0); return; }
    if (xi > x[n - 1]) { loc = n - 2; f = FT(1); return; }
    int l = n - 2;
    if (xi <= x[0]) l = 0;
    else
        for (int i = 0; i < n; i++)
            if (xi < x[i]) { l = i - 1; break; }
    loc = l;
    f = (xi - x[l]) / (x[l + 1] - x[l]);
}

// cloud size-table position, src/optics/cloud_optics.jl:170-182 (0-based loc)
template <typename FT>
__device__ __forceinline__ void cld_pos(FT re, FT r_lwr, FT r_upr, int nsize, int &loc, FT &fac) {
    FT dr = (r_upr - r_lwr) / FT(nsize - 1);
    re = m_max(m_min(re, r_upr), r_lwr);
    int l = (int)((re - r_lwr) / dr) + 1;
    l = l > nsize - 1 ? nsize - 1 : l;
    l = l < 1 ? 1 : l;
    fac = (re - r_lwr - FT(l - 1) * dr) / dr;
    loc = l - 1;
}

template <typename FT, int CHK>
__device__ inline void prepare_column(const ColShared<FT, CHK> &sh, const ColDims &d, const DevGas<FT> &lk,
                                      const DevCld<FT> *cld, const DevAero<FT> *aero, const DevState<FT> &as, int col) {
    const int nlay = d.nlay, nlev = d.nlev, tid = threadIdx.x, nt = blockDim.x;
    const FT *ld = as.layerdata + (size_t)4 * nlay * col;
    // thread index rotated by one wavefront: the loops after the layer records start on the wavefronts that have none
    const int tid2 = nt > 64 ? (tid >= 64 ? tid - 64 : tid + nt - 64) : tid;
    for (int k = tid; k < nlay; k += nt) {
        const FT col_dry = ld[4 * k + 0], p = ld[4 * k + 1], t = ld[4 * k + 2];
        LayerRec<FT> rec;
        rec.col_dry = col_dry;
        const int tropo = p > lk.p_ref_tropo ? 0 : 1;  // gas_optics.jl:188 (0 = lower)
        // compute_interp_frac_temp, gas_optics.jl:87-93
        const FT dT = lk.t_ref[1] - lk.t_ref[0];
        const int jT = loc_lower_eq0(t, dT, lk.n_t_ref, lk.t_ref);
        rec.fT = (t - lk.t_ref[jT]) / dT;
        // compute_interp_frac_press, gas_optics.jl:100-117
        const FT dlp = lk.ln_p_ref[0] - lk.ln_p_ref[1];
        const FT logp = m_log(p);
        int j = (int)((lk.ln_p_ref[0] - logp) / dlp) + 1;
        const int n_p_ref = lk.n_pp - 1;
        j = j < 1 ? 1 : j;
        j = j > n_p_ref - 1 ? n_p_ref - 1 : j;
        j += 1;                                          // 1-based jpress
        rec.fP = (lk.ln_p_ref[j - 2] - logp) / dlp;
        const int jP = (j + tropo) - 2;                  // (jpress + tropo1 - 1) - 1 -> 0-based lower plane
        rec.idx = jT | (jP << 8) | (tropo << 16);
        {   // vmr_h2o of this layer (get_vmr, VolumeMixingRatios.jl:91-129)
            const int ig = lk.idx_h2o;
            if (as.vmr_kind == RRTMGP_VMR_GM)
                rec.h2o = ig == 1 ? as.vmr_h2o[(size_t)nlay * col + k] : ig == 3 ? as.vmr_o3[(size_t)nlay * col + k] : as.vmr[ig - 1];
            else
                rec.h2o = as.vmr[(size_t)(ig - 1) + (size_t)as.ngas * ((size_t)k + (size_t)nlay * col)];
        }
        rec.dens_fact = FT(0.01) * p / t;                // gas_optics.jl:368-370
        rec.dry_fact = FT(1) / (FT(1) + rec.h2o);        // gas_optics.jl:367
        rec.ray_fac = (rec.h2o + FT(1)) * col_dry;       // compute_tau_rayleigh, gas_optics.jl:430-444
        rec.pad_ = FT(0);
        rec.pl_lay_loc = 0; rec.pl_lay_f = FT(0);
        if (d.lw) planck_pos(t, lk.t_planck, lk.n_t_plnk, rec.pl_lay_loc, rec.pl_lay_f);
        rec.cld_frac = rec.path_liq = rec.path_ice = rec.liq_fac = rec.ice_fac = FT(0);
        rec.liq_loc = rec.ice_loc = 0;
        if (d.has_cld) {
            const size_t o = (size_t)nlay * col + k;
            rec.cld_frac = as.cld_frac[o];
            rec.path_liq = as.cld_path_liq[o];
            rec.path_ice = as.cld_path_ice[o];
            cld_pos(as.cld_r_eff_liq[o], cld->radliq_lwr, cld->radliq_upr, cld->nsize_liq, rec.liq_loc, rec.liq_fac);
            cld_pos(as.cld_r_eff_ice[o], cld->radice_lwr, cld->radice_upr, cld->nsize_ice, rec.ice_loc, rec.ice_fac);
        }
        rec.aero_mask = 0; rec.rh_loc = 0; rec.rh_f = FT(0); rec.aod_t = rec.aod_ts = FT(0);
        if (d.has_aero) {
            const FT *mass = as.aero_mass + (size_t)RRTMGP_N_AEROSOLS * ((size_t)nlay * col + k);
            int any = 0;
            for (int ia = 0; ia < RRTMGP_N_AEROSOLS; ia++) any |= (mass[ia] > FT(0));  // aerosol_optics.jl:464-483
            rec.aero_mask = any;
            loc_factor_gen(ld[4 * k + 3], aero->rh_levels, aero->nrh, rec.rh_loc, rec.rh_f);
        }
        sh.lay[k] = rec;
    }
    // gas table: row ig (1-based gas index), row 0 = 1 (get_vmr, VolumeMixingRatios.jl:91-129)
    if (as.vmr_kind == RRTMGP_VMR_GM) {
        // only rows 1 (h2o) and 3 (o3) depend on the layer; the others repeat the well-mixed vector (LDS copy)
        // (the first wavefront is busy with the layer records above: the other wavefronts take these two loops first)
        for (int i = tid2; i < d.ngas1 * nlay; i += nt) {
            const int ig = i / nlay;
            if (ig != 1 && ig != 3) sh.vmr[i] = sh.tab_vmr_gm[ig];
        }
        for (int k = tid2; k < 2 * nlay; k += nt) {  // both profile rows in one batch of loads
            const bool o3 = k >= nlay;
            const int kl = o3 ? k - nlay : k;
            if ((o3 ? 3 : 1) < d.ngas1) sh.vmr[(o3 ? 3 : 1) * nlay + kl] = (o3 ? as.vmr_o3 : as.vmr_h2o)[(size_t)nlay * col + kl];
        }
    } else {
        // per-gas profiles: 8 independent loads per thread are in flight before the first store
        const int n = d.ngas1 * nlay;
        for (int i0 = tid; i0 < n; i0 += 8 * nt) {
            FT v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int i = i0 + j * nt, ig = i / nlay, k = i - ig * nlay;
                v[j] = (i < n && ig >= 1 && ig <= as.ngas)
                           ? as.vmr[(size_t)(ig - 1) + (size_t)as.ngas * ((size_t)k + (size_t)nlay * col)]
                           : (ig == 0 ? FT(1) : FT(0));
            }
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (i0 + j * nt < n) sh.vmr[i0 + j * nt] = v[j];
        }
    }
    if (d.lw)
        for (int k = tid2; k < nlev; k += nt) {
            LevelRec<FT> lr;
            planck_pos(as.t_lev[(size_t)nlev * col + k], lk.t_planck, lk.n_t_plnk, lr.loc, lr.f);
            sh.lev[k] = lr;
        }
    if (tid == 0) {
        if (d.lw) planck_pos(as.t_sfc[col], lk.t_planck, lk.n_t_plnk, sh.misc[d.nwaves], sh.miscf[0]);
        for (int w = 0; w < d.nwaves; w++) sh.misc[w] = 0;
        sh.misc[d.nwaves + 1] = nlay;  // first cloudy layer (min over layers below)
        sh.misc[d.nwaves + 2] = -1;    // last cloudy layer (max)
        sh.misc[d.nwaves + 4] = -1;    // last layer with a cloud or an aerosol: everything above it only absorbs (longwave)
    }
    __syncthreads();
    if (d.has_cld || d.has_aero) {
        // _get_start / _get_finish, cloud_optics.jl:310-322 (0-based, -1 when clear), one layer per thread
        for (int k = tid; k < nlay; k += nt) {
            const bool cl = d.has_cld && sh.lay[k].cld_frac > FT(0);
            if (cl) {
                atomicMin(&sh.misc[d.nwaves + 1], k);
                atomicMax(&sh.misc[d.nwaves + 2], k);
            }
            if (cl || sh.lay[k].aero_mask) atomicMax(&sh.misc[d.nwaves + 4], k);
        }
        __syncthreads();
        if (tid == 0 && sh.misc[d.nwaves + 1] == nlay) sh.misc[d.nwaves + 1] = -1;
    }
    __syncthreads();
}

// ---- optics_utils.jl:189-223 ---------------------------------------------------------------
// The increment arrives as (t2, t2*s2, (t2*s2)*g2), formed once per (layer, band) by prepare_chunk.
template <typename FT>
__device__ __forceinline__ void increment_2stream(FT &t1, FT &s1, FT &g1, FT t2, FT t2s2, FT t2s2g2) {
    const FT tau = t1 + t2;
    FT ssa = t1 * s1 + t2s2;
    // These two quotients are correctly rounded in Float32 too (ieee_div: the fast reciprocal + one FMA correction, 3
    // instructions more).  ssa of a near-conservative mixture feeds (1 - ssa) in the two-stream coefficients, which amplifies
    // its rounding error by 1 / (1 - ssa): with the 1.5-ulp `x * rcp(y)` form the worst of 1500 random cases (seed 633,
    // aerosol-laden) was 9.4e-2 W/m2 from the Float64 result, with this form 4.2e-2 — what the Float32 CPU oracle gives.
    const FT ssag = ieee_div(t1 * s1 * g1 + t2s2g2, m_max(Num<FT>::eps(), ssa));
    ssa = ieee_div(ssa, m_max(Num<FT>::eps(), tau));
    t1 = tau; s1 = ssa; g1 = ssag;
}

template <typename FT>
__device__ __forceinline__ void delta_scale(FT &tau, FT &ssa, FT &g) {
    const FT ssa_one_minus_g2 = ssa * (FT(1) - g) * (FT(1) + g);
    const FT one_minus_wf = (FT(1) - ssa) + ssa_one_minus_g2;
    const FT tau_s = one_minus_wf * tau;
    const FT ssa_s = ssa_one_minus_g2 / m_max(Num<FT>::eps(), one_minus_wf);
    const FT g_s = g / m_max(Num<FT>::eps(), FT(1) + g);
    tau = tau_s; ssa = ssa_s; g = g_s;
}

// ---- cloud optics of one (layer, band): cloud_optics.jl:154-244 -----------------------------
template <typename FT, int CHK>
__device__ __forceinline__ void cloud_props(const DevCld<FT> &lc, const ColShared<FT, CHK> &sh, int ibnd, int ice_rgh, int k,
                                            FT &tl, FT &tls, FT &tlsg, FT &ti, FT &tis, FT &tisg) {
    tl = tls = tlsg = ti = tis = tisg = FT(0);
    const LayerRec<FT> &L = sh.lay[k];
    const FT pl = L.path_liq, pi = L.path_ice;
    if (pl > Num<FT>::eps()) {
        const int nl = lc.nsize_liq;
        const FT *t = lc.liqdata + (size_t)(3 * nl) * ibnd;
        const int loc = L.liq_loc;
        const FT fac = L.liq_fac, fc1 = FT(1) - fac;
        tl = m_max((fc1 * t[loc] + fac * t[loc + 1]) * pl, FT(0));
        tls = (fc1 * t[nl + loc] + fac * t[nl + loc + 1]) * tl;
        tlsg = (fc1 * t[2 * nl + loc] + fac * t[2 * nl + loc + 1]) * tls;
    }
    if (pi > Num<FT>::eps()) {
        const int ni = lc.nsize_ice;
        const FT *t = lc.icedata + (size_t)(3 * ni) * ((size_t)ibnd + (size_t)lc.nband * (ice_rgh - 1));
        const int loc = L.ice_loc;
        const FT fac = L.ice_fac, fc1 = FT(1) - fac;
        ti = m_max((fc1 * t[loc] + fac * t[loc + 1]) * pi, FT(0));
        tis = (fc1 * t[ni + loc] + fac * t[ni + loc + 1]) * ti;
        tisg = (fc1 * t[2 * ni + loc] + fac * t[2 * ni + loc + 1]) * tis;
    }
}

// ---- aerosol optics of one (layer, band): aerosol_optics.jl:141-431 --------------------------
template <typename FT, int CHK>
__device__ inline void lookup_aerosol(const DevAero<FT> &la, const ColShared<FT, CHK> &sh, const FT *mass, const FT *size,
                                      int ibnd, int k, FT &tc, FT &tsc, FT &tsgc) {
    const int nrh = la.nrh, nbin = la.nbin;
    const int loc = sh.lay[k].rh_loc;
    const FT f = sh.lay[k].rh_f, omf = FT(1) - f;
    FT t_cum = FT(0), ts_cum = FT(0), tsg_cum = FT(0);
    auto size_bin = [&](FT sz) {  // locate_merra_size_bin, aerosol_optics.jl:438-451
        int bin = 0;
        for (int ib = 0; ib < nbin; ib++) {
            if (la.size_bin_limits[2 * ib] <= sz && sz <= la.size_bin_limits[2 * ib + 1]) { bin = ib; break; }
            bin = nbin - 1;
        }
        return bin;
    };
    constexpr int dust_ids[5] = {0, 7, 8, 9, 10}, salt_ids[5] = {1, 11, 12, 13, 14};
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const FT m = mass[dust_ids[s]];
        if (m > FT(0)) {
            const int bin = size_bin(size[dust_ids[s]]);
            const FT *tb = la.dust + 3 * ((size_t)bin + (size_t)nbin * ibnd);
            const FT t = m * tb[0], ts = t * tb[1], tsg = ts * tb[2];
            t_cum += t; ts_cum += ts; tsg_cum += tsg;
        }
    }
#pragma unroll
    for (int s = 0; s < 5; s++) {
        const FT m = mass[salt_ids[s]];
        if (m > FT(0)) {
            const int bin = size_bin(size[salt_ids[s]]);
            const FT *tb = la.sea_salt + 3 * ((size_t)loc + (size_t)nrh * ((size_t)bin + (size_t)nbin * ibnd));
            const FT t = m * (tb[0] * omf + tb[3] * f);
            const FT ts = t * (tb[1] * omf + tb[4] * f);
            const FT tsg = ts * (tb[2] * omf + tb[5] * f);
            t_cum += t; ts_cum += ts; tsg_cum += tsg;
        }
    }
    auto rh_species = [&](const FT *tab, FT m) {
        const FT *tb = tab + 3 * ((size_t)loc + (size_t)nrh * ibnd);
        const FT t = m * (tb[0] * omf + tb[3] * f);
        const FT ts = t * (tb[1] * omf + tb[4] * f);
        const FT tsg = ts * (tb[2] * omf + tb[5] * f);
        t_cum += t; ts_cum += ts; tsg_cum += tsg;
    };
    auto dry_species = [&](const FT *tab, FT m) {
        const FT *tb = tab + 3 * (size_t)ibnd;
        const FT t = m * tb[0], ts = t * tb[1], tsg = ts * tb[2];
        t_cum += t; ts_cum += ts; tsg_cum += tsg;
    };
    if (mass[2] > FT(0)) rh_species(la.sulfate, mass[2]);
    if (mass[3] > FT(0)) rh_species(la.black_carbon_rh, mass[3]);
    if (mass[4] > FT(0)) dry_species(la.black_carbon, mass[4]);
    if (mass[5] > FT(0)) rh_species(la.organic_carbon_rh, mass[5]);
    if (mass[6] > FT(0)) dry_species(la.organic_carbon, mass[6]);
    tc = t_cum; tsc = ts_cum; tsgc = tsg_cum;
}

// ---- (layer, band) records for layers [k0, k0 + kn) -------------------------------------------
// Callers synchronise the workgroup before (previous chunk fully consumed) and after.
template <typename FT, int CHK>
__device__ inline void prepare_chunk(const ColShared<FT, CHK> &sh, const ColDims &d, const DevGas<FT> &lk,
                                     const DevCld<FT> *cld, const DevAero<FT> *aero, const DevState<FT> &as, int col,
                                     int k0, int kn, bool delta) {
    const int nlay = d.nlay, nb = d.nbnd, tid = threadIdx.x, nt = blockDim.x;
    const int NE = lk.n_eta;
    for (int u = tid; u < CHK * nb; u += nt) {
        // consecutive lanes take consecutive bands of one layer: the record stores below are contiguous in LDS
        // (16-byte stride between lanes instead of 256) and the sh.lay[k] / sh.lev[k] reads are broadcasts
        const int kk = u / nb, b = u - kk * nb, k = k0 + kk;
        if (kk >= kn) continue;
        const int t = kk * NBMAX + b;
        // Planck band sources at levels k0 .. k0 + kn (interp1d_equispaced, compute_optical_props.jl:180-186): the
        // table reads are issued first so that their latency is covered by the rest of the task; the task of the
        // chunk's last layer also does the level above it
        FT pl0 = FT(0), pl1 = FT(0), pt0 = FT(0), pt1 = FT(0);
        const bool top_too = d.lw && kk == kn - 1;
        if (d.lw) {
            const FT *tp = lk.tot_planck + (size_t)lk.n_t_plnk * b + sh.lev[k].loc;
            pl0 = tp[0]; pl1 = tp[1];
            if (top_too) {
                const FT *tq = lk.tot_planck + (size_t)lk.n_t_plnk * b + sh.lev[k + 1].loc;
                pt0 = tq[0]; pt1 = tq[1];
            }
        }
        const int li = sh.lay[k].idx;
        const int jT = li & 0xff, tropo = li >> 16;
        const int ig0 = lk.key_species[0 + 2 * (tropo + 2 * b)], ig1 = lk.key_species[1 + 2 * (tropo + 2 * b)];
        const FT vmr1 = sh.vmr[ig0 * nlay + k], vmr2 = sh.vmr[ig1 * nlay + k];
        // compute_interp_frac_eta, gas_optics.jl:129-170
        int je[2];
        FT fe[2], cm[2];
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const FT eta_half = lk.eta_half[(tropo * nb + b) * lk.n_t_ref + jT + it];  // vmr_ref ratio of the key species
            const FT col_mix = vmr1 + eta_half * vmr2;
            FT eta = vmr1 * (FT(1) / col_mix);
            if (col_mix <= FT(0)) eta = FT(0.5);
            const FT loc_eta = eta * FT(NE - 1);
            int j = (int)loc_eta;
            j = j > NE - 2 ? NE - 2 : j;
            je[it] = j;
            fe[it] = loc_eta - FT(j);
            cm[it] = col_mix;
        }
        sh.ch->je[t] = je[0] | (je[1] << 8);
        {   // the weights every g-point of the band would form from (fT, fP, fe, col_mix): formed once here
            const FT fT = sh.lay[k].fT, fP = sh.lay[k].fP, cd = sh.lay[k].col_dry;
            const FT omfT = FT(1) - fT, omfP = FT(1) - fP;
            sh.ch->wgt[t] = V4<FT>{(FT(1) - fe[0]) * omfT, fe[0] * omfT, (FT(1) - fe[1]) * fT, fe[1] * fT};
            sh.ch->amp[t] = V4<FT>{(cm[0] * omfP) * cd, (cm[0] * fP) * cd, (cm[1] * omfP) * cd, (cm[1] * fP) * cd};
        }
        if (d.lw) {
            sh.ch->Blev[t] = pl0 * (FT(1) - sh.lev[k].f) + pl1 * sh.lev[k].f;
            if (top_too) sh.ch->Blev[(kk + 1) * NBMAX + b] = pt0 * (FT(1) - sh.lev[k + 1].f) + pt1 * sh.lev[k + 1].f;
        }
        if (d.lw && !d.twostream) {  // layer sources: no-scattering solver only (Blay is not allocated otherwise)
            const FT *tp = lk.tot_planck + (size_t)lk.n_t_plnk * b + sh.lay[k].pl_lay_loc;
            sh.ch->Blay[t] = tp[0] * (FT(1) - sh.lay[k].pl_lay_f) + tp[1] * sh.lay[k].pl_lay_f;
        }
        if (d.has_cld) {
            FT c0 = FT(0), c1 = FT(0), c2 = FT(0);
            if (sh.lay[k].cld_frac > FT(0)) {
                FT tl, tls, tlsg, ti, tis, tisg;
                cloud_props(*cld, sh, b, as.ice_rgh, k, tl, tls, tlsg, ti, tis, tisg);
                if (d.twostream) {  // add_cloud_optics_2stream!, cloud_optics.jl:120-130
                    FT tau_cl = tl + ti;
                    FT ssa_cl = tls + tis;
                    FT g_cl = (tlsg + tisg) / m_max(Num<FT>::eps(), ssa_cl);
                    ssa_cl /= m_max(Num<FT>::eps(), tau_cl);
                    if (delta) delta_scale(tau_cl, ssa_cl, g_cl);
                    // stored as (tau, tau*ssa, tau*ssa*g): the products every g-point of the band would form in
                    // increment_2stream, same operations in the same order
                    c0 = tau_cl; c1 = tau_cl * ssa_cl; c2 = c1 * g_cl;
                } else {
                    c0 = (tl - tls) + (ti - tis);  // cloud_optics.jl:45
                }
            }
            sh.ch->cld[t] = V4<FT>{c0, c1, c2, FT(0)};
        }
        if (d.has_aero) {
            FT a0 = FT(0), a1 = FT(0), a2 = FT(0);
            if (sh.lay[k].aero_mask) {
                const size_t o = (size_t)RRTMGP_N_AEROSOLS * ((size_t)nlay * col + k);
                FT ta, tsa, tsga;
                lookup_aerosol(*aero, sh, as.aero_mass + o, as.aero_size + o, b, k, ta, tsa, tsga);
                if (!d.lw && b == aero->iband_550nm - 1) { sh.lay[k].aod_t = ta; sh.lay[k].aod_ts = tsa; }
                if (d.twostream) {  // aerosol_optics.jl:113-122
                    FT g_aero = tsga / m_max(Num<FT>::eps(), tsa);
                    FT ssa_aero = tsa / m_max(Num<FT>::eps(), ta);
                    if (delta) delta_scale(ta, ssa_aero, g_aero);
                    a0 = ta; a1 = ta * ssa_aero; a2 = a1 * g_aero;
                } else {
                    a0 = ta - tsa;  // aerosol_optics.jl:45
                }
            }
            sh.ch->aer[t] = V4<FT>{a0, a1, a2, FT(0)};
        }
    }
    // minor-gas scalings, compute_tau_minor gas_optics.jl:364-396; 0 where the gas is absent (vmr <= 0) and in the
    // padding slots.  Row kk holds the slots of layer k0 + kk: consecutive lanes write consecutive words.
    const int S = d.max_int;
    for (int t = tid; t < S * CHK; t += nt) {
        const int kk = t / S, slot = t - kk * S, k = k0 + kk;
        if (kk >= kn) continue;
        const int tropo = sh.lay[k].idx >> 16;
        FT scaling = FT(0);
        const int i = slot < (tropo ? lk.m_nslot[1] : lk.m_nslot[0]) ? (tropo ? lk.m_slot_int[1] : lk.m_slot_int[0])[slot] : -1;
        if (i == RAYLEIGH_SLOT) scaling = sh.lay[k].ray_fac;   // compute_tau_rayleigh, gas_optics.jl:430-444
        if (i >= 0) {
            const int *gd = (tropo ? lk.m_gasdata[1] : lk.m_gasdata[0]) + 4 * i;
            const FT vmr_imnr = sh.vmr[gd[0] * nlay + k];
            if (vmr_imnr > FT(0)) {
                scaling = vmr_imnr * sh.lay[k].col_dry;
                if (gd[2] == 1) {
                    scaling *= sh.lay[k].dens_fact;
                    if (gd[1] > 0) {
                        const FT vs = sh.vmr[gd[1] * nlay + k];
                        if (gd[3] == 1) scaling *= (FT(1) - vs * sh.lay[k].dry_fact);
                        else scaling *= vs * sh.lay[k].dry_fact;
                    }
                }
            }
        }
        sh.mscale[t] = scaling;
    }
}

// ---- per-lane band constants ---------------------------------------------------------
// (explicit lower/upper members: arrays indexed by the run-time region would live in scratch memory)
struct LaneBand {
    int g, ibnd, ngb;
    int m_pack;          // p0 | n0 << 8 | p1 << 16 | n1 << 24   (first scaling PAIR of the band / slot count, per region)
    int npw0, npw1;      // wave-uniform: slot pairs of the wavefront's largest band, per region (>= 1)
    unsigned gk;         // arena byte offset of this lane's g-point in the kmajor table
    unsigned gm0, gm1;   // ... of this lane's first slot pair in kminor lower / upper
    unsigned gE;         // g * sizeof(FT)
    __device__ __forceinline__ int m_st(unsigned tropo) const { return (m_pack >> (tropo ? 16 : 0)) & 0xff; }  // first pair
    __device__ __forceinline__ int m_n(unsigned tropo) const { return (m_pack >> (tropo ? 24 : 8)) & 0xff; }
    __device__ __forceinline__ unsigned gm(unsigned tropo) const { return tropo ? gm1 : gm0; }
};

template <typename FT>
__device__ __forceinline__ LaneBand lane_band(const DevGas<FT> &lk, int g) {
    LaneBand lb;
    lb.g = g;
    lb.ibnd = lk.gpt2bnd[g];
    const int gi = g - lk.bnd_lo[lb.ibnd];
    lb.ngb = lk.bnd_ng[lb.ibnd];
    // slots of the band per region: its minor contributors, and in front of them the Rayleigh slot of a SW lookup
    const int lead = lk.is_sw ? 1 : 0;
    const int n0 = lk.m_bnd_st[0][lb.ibnd + 1] - lk.m_bnd_st[0][lb.ibnd] + lead, n1 = lk.m_bnd_st[1][lb.ibnd + 1] - lk.m_bnd_st[1][lb.ibnd] + lead;
    const int p0 = lk.m_st2[0][lb.ibnd] / MINOR_PAIR, p1 = lk.m_st2[1][lb.ibnd] / MINOR_PAIR;
    lb.m_pack = p0 | (n0 << 8) | (p1 << 16) | (n1 << 24);
    // the gathers of a layer are issued for the wavefront's largest band (a gather costs the same with 16 lanes active)
    int w0 = 1, w1 = 1;
    while (__any(n0 > w0 * MINOR_PAIR)) w0++;
    while (__any(n1 > w1 * MINOR_PAIR)) w1++;
    lb.npw0 = __builtin_amdgcn_readfirstlane(w0);
    lb.npw1 = __builtin_amdgcn_readfirstlane(w1);
    constexpr unsigned E = sizeof(FT);
    lb.gE = g * E;
    lb.gk = lk.off_kmajor + g * KMAJOR_ENTRY_BYTES;
    lb.gm0 = lk.off_kminor[0] + (lk.m_koff[0][lb.ibnd] + gi * MINOR_ENTRY) * E;
    lb.gm1 = lk.off_kminor[1] + (lk.m_koff[1][lb.ibnd] + gi * MINOR_ENTRY) * E;
    return lb;
}

// ---- gas optics of one (layer, g-point): src/optics/gas_optics.jl:176-320 -----------------
// kk = layer index inside the current chunk.  Table reads use 32-bit byte offsets on a
// wave-uniform base (global_load ... saddr form); every offset fits 32 bits by construction.
template <typename T>
__device__ __forceinline__ T ldg(const void *base, unsigned byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <typename FT>
struct alignas(2 * sizeof(FT)) V2 {
    FT x, y;
};

// The gas optics of a layer come in two halves, so that a layer loop can put arithmetic that does not depend on
// this layer's tables (the two-stream coefficients and the adding step of an EARLIER layer) between them:
//   gas_issue  : LDS records -> gather addresses -> every gather of the layer in flight; nothing is consumed;
//   gas_finish : ONE wait, then the interpolations.
template <typename FT>
struct SlotPair {       // one kminor entry per T plane: {c_2p(e), c_2p(e+1), c_2p+1(e), c_2p+1(e+1)} at (jT, jeta1) and (jT + 1, jeta2)
    V4<FT> a, b;
};
constexpr int MINOR_PAIRS_AHEAD = 4;   // slot pairs gathered ahead of the layer's single wait; bands with more than 8 slots loop
template <typename FT, bool SW>
struct GasLoads {
    FT k000, k100, k010, k110, q000, q100, q010, q110;   // kmajor corners (T plane 1: k, T plane 2: q)
    FT p000, p100, p010, p110, r000, r100, r010, r110;   // planck_fraction corners (LW)
    SlotPair<FT> g0, g1, g2, g3;                         // minor-gas slot pairs (g1..g3 only when the wavefront has them)
    V2<FT> s0, s1, s2, s3;                               // their scalings
    V4<FT> wr, ar;                                       // (eta, T) weights; amount x pressure weights
    FT fP;
    unsigned a1, a2, gstep;
    const FT *ms;
    int n, npw;
};

template <typename FT, bool SW, int CHK>
__device__ __forceinline__ GasLoads<FT, SW> gas_issue(const DevGas<FT> &lk, const ColShared<FT, CHK> &sh, const LaneBand &lb,
                                                      int k, int kk) {
    constexpr unsigned E = sizeof(FT), EK = KMAJOR_ENTRY_BYTES;  // one 16-byte entry per (t, p, eta, g-point): common.h
    GasLoads<FT, SW> G;
    const LayerRec<FT> &L = sh.lay[k];
    const int li = L.idx;
    const unsigned jT = li & 0xff, jP = (li >> 8) & 0xff, tropo = li >> 16;
    G.fP = L.fP;
    const int r = kk * NBMAX + lb.ibnd;
    const unsigned jep = sh.ch->je[r];
    const unsigned je1 = jep & 0xff, je2 = jep >> 8;
    // band-level weights prepared once per (layer, band) by prepare_chunk
    G.wr = sh.ch->wgt[r];
    G.ar = sh.ch->amp[r];
    const unsigned NE = lk.n_eta, NG = lk.n_gpt;
    // interp3d, optics_utils.jl:136-181, on the [t][p][eta][gpt] layout
    const unsigned sE = NG * EK, sP = NE * sE;
    const unsigned row = (jT * lk.n_pp + jP) * NE;  // (t, p) row, in eta units
    const unsigned o1 = __umul24(row + je1, sE) + lb.gk;
    const unsigned o2 = __umul24(row + lk.n_pp * NE + je2, sE) + lb.gk;
    // the corner strides are wave-uniform: they go into the scalar base of the load, so that one VGPR
    // offset serves four loads
    const char *b0 = lk.arena, *b1 = lk.arena + sE, *b2 = lk.arena + sP, *b3 = lk.arena + sP + sE;
    // The vector memory pipeline prices a gather by the instruction, not by the bytes: 8 cycles of the CU's texture
    // addresser for a 4-byte one, 16 for 8 AND for 16 bytes per lane (tools/ubench/gather_l1.hip), and these kernels are
    // bound by exactly that.  Every kmajor gather is therefore a 16-byte one: the table entry of (t, p, eta, g) also holds
    // the neighbours along eta (and along p, SW Float32) that the interpolation reads with it (build_gas, common.h).
    G.p000 = G.p100 = G.p010 = G.p110 = G.r000 = G.r100 = G.r010 = G.r110 = FT(0);
    if constexpr (SW && sizeof(FT) == 4) {          // {k(e, p), k(e+1, p), k(e, p+1), k(e+1, p+1)}: 2 gathers instead of 8
        const V4<FT> a = ldg<V4<FT>>(b0, o1), e = ldg<V4<FT>>(b0, o2);
        G.k000 = a.x; G.k100 = a.y; G.k010 = a.z; G.k110 = a.w;
        G.q000 = e.x; G.q100 = e.y; G.q010 = e.z; G.q110 = e.w;
    } else if constexpr (SW) {                      // {k(e), k(e+1)}: 4 instead of 8
        const V2<FT> a = ldg<V2<FT>>(b0, o1), c = ldg<V2<FT>>(b2, o1), e = ldg<V2<FT>>(b0, o2), g = ldg<V2<FT>>(b2, o2);
        G.k000 = a.x; G.k100 = a.y; G.k010 = c.x; G.k110 = c.y;
        G.q000 = e.x; G.q100 = e.y; G.q010 = g.x; G.q110 = g.y;
    } else if constexpr (sizeof(FT) == 4) {         // {k(e), pf(e), k(e+1), pf(e+1)}: 4 instead of 8
        const V4<FT> a = ldg<V4<FT>>(b0, o1), c = ldg<V4<FT>>(b2, o1), e = ldg<V4<FT>>(b0, o2), g = ldg<V4<FT>>(b2, o2);
        G.k000 = a.x; G.p000 = a.y; G.k100 = a.z; G.p100 = a.w;
        G.k010 = c.x; G.p010 = c.y; G.k110 = c.z; G.p110 = c.w;
        G.q000 = e.x; G.r000 = e.y; G.q100 = e.z; G.r100 = e.w;
        G.q010 = g.x; G.r010 = g.y; G.q110 = g.z; G.r110 = g.w;
    } else {                                        // {k, pf}: 8 gathers of 16 bytes
        const V2<FT> a = ldg<V2<FT>>(b0, o1), b = ldg<V2<FT>>(b1, o1);
        const V2<FT> c = ldg<V2<FT>>(b2, o1), d = ldg<V2<FT>>(b3, o1);
        const V2<FT> e = ldg<V2<FT>>(b0, o2), f = ldg<V2<FT>>(b1, o2);
        const V2<FT> g = ldg<V2<FT>>(b2, o2), h = ldg<V2<FT>>(b3, o2);
        G.k000 = a.x; G.k100 = b.x; G.k010 = c.x; G.k110 = d.x; G.q000 = e.x; G.q100 = f.x; G.q010 = g.x; G.q110 = h.x;
        G.p000 = a.y; G.p100 = b.y; G.p010 = c.y; G.p110 = d.y; G.r000 = e.y; G.r100 = f.y; G.r010 = g.y; G.r110 = h.y;
    }
    // (compute_tau_rayleigh, gas_optics.jl:430-444: the four (eta, T) corners of krayl arrive in slot 0 of the first
    // slot pair below, and (h2o + 1) col_dry as that slot's scaling)
    // compute_tau_minor, gas_optics.jl:344-412 (scalings hoisted to prepare_chunk).
    // The slots of a g-point sit in pairs, and a kminor entry holds a pair at eta and eta + 1 (build_gas): one 16-byte
    // gather per T plane and one 8-byte LDS read of the 2 scalings serve two contributors.  The first pair is loaded
    // UNCONDITIONALLY, right behind the kmajor corners: no exec masking.  Padding entries (a band without minor gases owns
    // one all-padding pair) are 0 in the table and carry a zero scaling, which leaves the in-order sum unchanged.  Further
    // pairs are issued as far as the LARGEST band of this wavefront has them in this region (wave-uniform branches; lanes
    // of smaller bands re-read their first pair with zero scalings).
    G.n = lb.m_n(tropo);
    G.npw = tropo ? lb.npw1 : lb.npw0;
    const char *kmn = lk.arena;
    const unsigned NCb = (tropo ? lk.m_ncontrib[1] : lk.m_ncontrib[0]) * E;
    G.a1 = __umul24(jT * NE + je1, NCb) + lb.gm(tropo);
    G.a2 = __umul24((jT + 1) * NE + je2, NCb) + lb.gm(tropo);
    G.gstep = lb.ngb * (MINOR_ENTRY * E);  // byte distance between the pairs of one g-point
    G.ms = sh.mscale + kk * sh.mscale_row + lb.m_st(tropo) * MINOR_PAIR;
    const V2<FT> z2{FT(0), FT(0)};
    // g1..g3 / s1..s3 stay unset unless loaded (gas_finish reads them under the same wave-uniform conditions)
    G.g0 = SlotPair<FT>{ldg<V4<FT>>(kmn, G.a1), ldg<V4<FT>>(kmn, G.a2)};
    G.s0 = *reinterpret_cast<const V2<FT> *>(G.ms);
    auto further = [&](int p, SlotPair<FT> &gp, V2<FT> &sp) {
        const bool mine = G.n > p * MINOR_PAIR;
        const unsigned c = mine ? p * G.gstep : 0u;
        gp = SlotPair<FT>{ldg<V4<FT>>(kmn, G.a1 + c), ldg<V4<FT>>(kmn, G.a2 + c)};
        sp = *reinterpret_cast<const V2<FT> *>(G.ms + (mine ? p * MINOR_PAIR : 0));
        if (!mine) sp = z2;
    };
    if (G.npw > 1) {   // nested: a wavefront whose bands have one pair evaluates one branch, not three
        further(1, G.g1, G.s1);
        if (G.npw > 2) {
            further(2, G.g2, G.s2);
            if (G.npw > 3) further(3, G.g3, G.s3);
        }
    }
    return G;
}

// `stores_after`: vector-memory stores issued BEHIND the gathers (the sweep records of an earlier layer).  vmcnt
// retires in order, so waiting for "at most that many outstanding" covers every gather and leaves the stores in flight.
template <typename FT, bool SW>
__device__ __forceinline__ void gas_finish(const DevGas<FT> &lk, const GasLoads<FT, SW> &G, FT &tau, FT &ssa, FT &pfrac,
                                           int stores_after = 0) {
    constexpr unsigned E = sizeof(FT);
    // every gather of this layer has been issued (some arithmetic ago): one wait instead of one per operand
    if (stores_after == 3) __builtin_amdgcn_s_waitcnt(0x0F73);       // vmcnt(3) expcnt(7) lgkmcnt(15)
    else if (stores_after == 6) __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6)
    else __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
    const FT w11 = G.wr.x, w21 = G.wr.y, w12 = G.wr.z, w22 = G.wr.w;
    FT tau_minor = FT(0);
    FT tau_ray = FT(0);
    auto consume = [&](const SlotPair<FT> &c, const V2<FT> &sc, bool first) {
        // interp2d, optics_utils.jl:85-98, slot by slot in the reference's order; the leading slot of a SW lookup is the
        // Rayleigh coefficient, and that product is tau_rayleigh (gas_optics.jl:430-444)
        const FT t0 = (w11 * c.a.x + w21 * c.a.y + w12 * c.b.x + w22 * c.b.y) * sc.x;
        if (SW && first) tau_ray = t0;
        else tau_minor += t0;
        tau_minor += (w11 * c.a.z + w21 * c.a.w + w12 * c.b.z + w22 * c.b.w) * sc.y;
    };
    consume(G.g0, G.s0, true);
    if (G.npw > 1) {
        consume(G.g1, G.s1, false);
        if (G.npw > 2) {
            consume(G.g2, G.s2, false);
            if (G.npw > 3) {
                consume(G.g3, G.s3, false);
                if (G.npw > MINOR_PAIRS_AHEAD) {   // bands with more than 8 slots (rare; exposed)
                    const char *kmn = lk.arena;
                    for (int p = MINOR_PAIRS_AHEAD; p * MINOR_PAIR < G.n; p++) {
                        const unsigned c = __umul24((unsigned)p, G.gstep);
                        const SlotPair<FT> e{ldg<V4<FT>>(kmn, G.a1 + c), ldg<V4<FT>>(kmn, G.a2 + c)};
                        consume(e, *reinterpret_cast<const V2<FT> *>(G.ms + p * MINOR_PAIR), false);
                    }
                }
            }
        }
    }
    // interp3d (optics_utils.jl:136-181) with the (eta, T) products and the column-amount x pressure products hoisted:
    // cm (1-fP) (1-fT) ((1-fe) k000 + fe k100) + ... regrouped as amp * (w11 k000 + w21 k100) + ...
    const FT tau_major = G.ar.x * (w11 * G.k000 + w21 * G.k100) + G.ar.y * (w11 * G.k010 + w21 * G.k110) +
                         G.ar.z * (w12 * G.q000 + w22 * G.q100) + G.ar.w * (w12 * G.q010 + w22 * G.q110);
    if (!SW) {
        const FT fP = G.fP, omfP = FT(1) - fP;
        pfrac = omfP * ((w11 * G.p000 + w21 * G.p100) + (w12 * G.r000 + w22 * G.r100)) +
                fP * ((w11 * G.p010 + w21 * G.p110) + (w12 * G.r010 + w22 * G.r110));
        tau = m_max(tau_major + tau_minor, FT(0));
        ssa = FT(0);
    } else {
        tau = m_max(tau_major + tau_minor + tau_ray, FT(0));
        ssa = tau_ray * m_rcp(tau);
        if (tau <= FT(0)) ssa = FT(0);
        pfrac = FT(0);
    }
    (void)E;
}

template <typename FT, bool SW, int CHK>
__device__ __forceinline__ void gas_optics(const DevGas<FT> &lk, const ColShared<FT, CHK> &sh, const LaneBand &lb, int k, int kk,
                                           int nb, FT &tau, FT &ssa, FT &pfrac) {
    const GasLoads<FT, SW> G = gas_issue<FT, SW>(lk, sh, lb, k, kk);
    gas_finish<FT, SW>(lk, G, tau, ssa, pfrac);
}

// ---- McICA mask for this lane's g-point: cloud_optics.jl:264-334 ----------------------------
// Bits of (m0, m1) are layers 0..63 / 64..127.  Columns with more than 128 layers keep one 64-bit word per 64 layers
// in LDS instead (sh.mask[word * 256 + lane]: only this lane reads them back, no barrier).  Returns any(mask).
template <typename FT, int CHK>
__device__ inline bool build_cloud_mask(const ColShared<FT, CHK> &sh, const ColDims &d, uint64_t key, uint64_t &m0,
                                        uint64_t &m1) {
    m0 = m1 = 0;
    const int start = sh.misc[d.nwaves + 1], finish = sh.misc[d.nwaves + 2];
    const bool big = d.nlay > 128;
    if (big)
        for (int w = 0; w < (d.nlay + 63) / 64; w++) sh.mask[w * 256 + threadIdx.x] = 0;
    if (start < 0) return false;
    if (big) {
        int draw = 0, widx = finish >> 6;
        uint64_t wcur = 0;
        bool any = false;
        FT cf_above = sh.lay[finish].cld_frac;
        double r_above = mcica_draw(key, draw++);
        bool mask_above = r_above >= (double)(FT(1) - cf_above);
        if (mask_above) { wcur |= 1ULL << (finish & 63); any = true; }
        for (int k = finish - 1; k >= start; k--) {
            if ((k >> 6) != widx) { sh.mask[widx * 256 + threadIdx.x] = wcur; wcur = 0; widx = k >> 6; }
            const FT cf = sh.lay[k].cld_frac;
            bool mk = false;
            if (cf > FT(0)) {
                const double r = mask_above ? r_above : mcica_draw(key, draw++) * (double)(FT(1) - cf_above);
                mk = r >= (double)(FT(1) - cf);
                r_above = r;
            }
            if (mk) { wcur |= 1ULL << (k & 63); any = true; }
            cf_above = cf;
            mask_above = mk;
        }
        sh.mask[widx * 256 + threadIdx.x] = wcur;
        return any;
    }
    int draw = 0;
    FT cf_above = sh.lay[finish].cld_frac;
    double r_above = mcica_draw(key, draw++);
    bool mask_above = r_above >= (double)(FT(1) - cf_above);
    auto setbit = [&](int k) { if (k < 64) m0 |= (1ULL << k); else m1 |= (1ULL << (k - 64)); };
    if (mask_above) setbit(finish);
    for (int k = finish - 1; k >= start; k--) {
        const FT cf = sh.lay[k].cld_frac;
        bool mk;
        if (cf > FT(0)) {
            const double r = mask_above ? r_above : mcica_draw(key, draw++) * (double)(FT(1) - cf_above);
            mk = r >= (double)(FT(1) - cf);
            r_above = r;
        } else {
            mk = false;
        }
        if (mk) setbit(k);
        cf_above = cf;
        mask_above = mk;
    }
    return (m0 | m1) != 0;
}
__device__ __forceinline__ bool mask_bit(uint64_t m0, uint64_t m1, int k) {
    return k < 64 ? ((m0 >> k) & 1ULL) : ((m1 >> (k - 64)) & 1ULL);
}
// The layer loops visit the layers in order, so the mask is walked instead of indexed: one shift and one test per
// layer (the 64-bit variable shift + select of mask_bit costs six VALU instructions).  UP: layers 0, 1, 2, ... (bit 0
// is the current layer, shift right); otherwise nlay-1, nlay-2, ... (bit 63 is the current layer, shift left).
// The word switch happens once per 64 layers: as a real (wave-uniform) branch it costs the scalar compare that is there
// anyway; if-converted it is two v_cndmask_b32 in every layer.  The empty asm keeps the compiler from if-converting.
#define RR_REFILL_BRANCH() asm volatile("" ::: "memory")
template <bool UP>
struct MaskWalk {
    uint64_t cur, other;
    const uint64_t *base;  // wave-uniform; != nullptr: more than 128 layers, lane t's word w is base[w * 256 + t] (LDS)
    __device__ __forceinline__ const uint64_t *word(int w) const { return base + w * 256 + threadIdx.x; }
    __device__ __forceinline__ MaskWalk(uint64_t m0, uint64_t m1, int nlay, const uint64_t *lds_words = nullptr) {
        base = nlay > 128 ? lds_words : nullptr;
        if (base) {
            const int top = (nlay - 1) >> 6;
            cur = UP ? *word(0) : *word(top) << (63 - ((nlay - 1) & 63));
            other = UP ? *word(1) : *word(top - 1);
        } else if (UP) { cur = m0; other = m1; }
        else {
            if (nlay > 64) { cur = m1 << (128 - nlay); other = m0; }
            else { cur = m0 << (64 - nlay); other = 0; }
        }
    }
    // Deep columns: `other` = the word after the current one, fetched from LDS once per 64 layers.  Called at the start
    // of every chunk with the first layer of the chunk in walking order (chunk edges fall on word edges), so the
    // per-layer step below never touches memory.
    __device__ __forceinline__ void refill(int k_edge) {
        if (!base) return;
        if (UP) { if ((k_edge & 63) == 0) other = *word((k_edge >> 6) + 1); }   // (one word past the last is allocated)
        else if ((k_edge & 63) == 63 && k_edge >= 64) other = *word((k_edge >> 6) - 1);
    }
    // the bit of layer `kq`, which must be the next one in walking order
    __device__ __forceinline__ bool next(int kq) {
        bool b;
        if (UP) {
            b = (unsigned)cur & 1u;
            cur >>= 1;
            if ((kq & 63) == 63) { RR_REFILL_BRANCH(); cur = other; }
        } else {
            b = (long long)cur < 0;
            cur <<= 1;
            if ((kq & 63) == 0) { RR_REFILL_BRANCH(); cur = other; }
        }
        return b;
    }
};

// ---- column queue -------------------------------------------------------------------------------------
// The persistent grid takes its first column statically (blockIdx.x) and every further one from a device-wide
// counter: columns differ in cost (cloudy layers, minor-gas counts across the tropopause), and with a static
// stride the launch ends when the unluckiest workgroup does.  Which workgroup solves a column does not enter the
// result (per-workgroup scratch, McICA keyed by the global column).
template <typename FT, int CHK>
__device__ __forceinline__ int next_column(const ColShared<FT, CHK> &sh, const ColDims &d, int *queue) {
    __syncthreads();
    if (threadIdx.x == 0) sh.misc[d.nwaves + 3] = (int)gridDim.x + atomicAdd(queue, 1);
    __syncthreads();
    return sh.misc[d.nwaves + 3];
}

// Called by every workgroup when it has left the column loop: the LAST one out zeroes {next column, workgroups done}, so the
// counters are ready for the next launch on this lane and no memset has to be queued in front of it.  (Every workgroup has made
// its final atomicAdd on queue[0] before it counts itself done, so the reset cannot race with a taker.)
// NO fence: both counters are only ever touched by device-scope atomics (performed at the L2) and by the two stores of the
// last workgroup, which the end of the kernel publishes; the same thread's last atomicAdd on queue[0] has RETURNED (its value
// was consumed by next_column) before this one is issued.  A __threadfence() here is an L1 invalidate on gfx950: with one
// column per workgroup (short steps) every workgroup that finished emptied the vector L1 of its CU under the three other
// resident workgroups — +12 % on a 1 024-column step (profiles/r05_small_step_ab.txt).
__device__ __forceinline__ void queue_release(int *queue) {
    if (threadIdx.x == 0) {
        const int done = __hip_atomic_fetch_add(&queue[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            __hip_atomic_store(&queue[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&queue[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- sweep scratch: NV values per (level, lane), lane-contiguous (3; 6 when the clear-sky
// recurrences are carried next to the all-sky ones) ----------------------------------------
constexpr int SWEEP_LANES = 256;  // lanes per scratch row, whatever the workgroup size (<= 256 g-points per lookup)
typedef unsigned sweep_u2 __attribute__((ext_vector_type(2)));
template <typename FT, int NV = 3>
struct Sweep {
    // The slab of this workgroup as a BUFFER resource (4 SGPRs).  A buffer access forms its address from the resource base,
    // one 32-bit VGPR offset (the lane: it never changes), one 32-bit SGPR offset (the row: wave-uniform, scalar arithmetic)
    // and an immediate - no vector instruction computes an address.  As global_load / global_store with one 32-bit offset
    // sum (rounds 2-5) every access of the second sweep cost a v_add_u32 (48 of the ~240 VALU instructions of a 16-level
    // batch): the compiler cannot fold an unsigned 32-bit term into the 64-bit address, and written as 64-bit pointer
    // arithmetic it re-associates the lane into the base and adds the level with v_lshl_add_u64 per level.
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned lane;  // threadIdx.x * sizeof(FT)
    static constexpr unsigned row = SWEEP_LANES * sizeof(FT);  // a compile-time stride
    static constexpr int AUX_LD = RR_SWEEP_NT >= 1 ? 2 : 0, AUX_ST = RR_SWEEP_NT >= 2 ? 2 : 0;   // nt bit (experiments)
    __device__ __forceinline__ Sweep(FT *slab, size_t bytes, unsigned tid)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes), 0x00020000)),
          lane(tid * (unsigned)sizeof(FT)) {}
    __device__ __forceinline__ void put(int lev, int a, FT v) const {
        const int so = (lev * NV + a) * (int)row;
        if constexpr (sizeof(FT) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, lane, so, AUX_ST);
        else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sweep_u2, v), rsrc, lane, so, AUX_ST);
    }
    __device__ __forceinline__ FT get(int lev, int a) const {
        const int so = (lev * NV + a) * (int)row;
        if constexpr (sizeof(FT) == 4) return __builtin_bit_cast(FT, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane, so, AUX_LD));
        else return __builtin_bit_cast(FT, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane, so, AUX_LD));
    }
    // Three values of one level at once (a = 0, or 3 for the clear-sky twin): three rows, three 4-byte accesses per lane.
    // ([level][lane] records, one access per lane and level, were measured twice: 12-byte records in round 3, and 16-byte
    // slots read with one 12-byte load in round 4 — written as 12 bytes or as the whole slot — 3.5 vs 4.1 M columns/s.
    // Fewer memory instructions, yet slower: tools/experiments/README.md.)
    __device__ __forceinline__ void put3(int lev, int a, FT x, FT y, FT z) const {
        put(lev, a, x); put(lev, a + 1, y); put(lev, a + 2, z);
    }
    __device__ __forceinline__ void get3(int lev, int a, FT &x, FT &y, FT &z) const {
        x = get(lev, a); y = get(lev, a + 1); z = get(lev, a + 2);
    }
};

// ---- write one column's broadband fluxes ---------------------------------------------------
// compute_net_flux! (Fluxes.jl:225-237) then apply_metric_scaling! (:295-304): net = up - dn first,
// then up, dn, net (and dir) are each multiplied by the (nlev, ncol) factor.
template <typename FT, int CHK>
__device__ inline void store_column(const DevFlux<FT> &fl, const ColShared<FT, CHK> &sh, const ColDims &d, int col, int ncol,
                                    bool zero, const DevGas<FT> &lk) {
    const int nlev = d.nlev;
    // accumulator components per level: all-sky (up, dn[, dir]) and, with the clear-sky diagnostic, the same again
    const int nset = d.diag ? 2 : 1, ncomp = d.n_acc / nset;
    for (int i = threadIdx.x; i < nlev * nset; i += blockDim.x) {
        const int set = i / nlev, lev = i - set * nlev;
        FT c[3] = {FT(0), FT(0), FT(0)};
        if (!zero) {
            for (int w = 0; w < d.nseg; w++)
                for (int a = 0; a < ncomp; a++) c[a] += sh.acc[((size_t)w * nlev + lev) * d.n_acc + set * ncomp + a];
        }
        FT up = c[0], dn = c[1], dir = c[2];
        FT net = up - dn;
        if (fl.metric) {
            const FT m = fl.metric[(size_t)nlev * col + lev];
            up *= m; dn *= m; net *= m; dir *= m;
        }
        const size_t o = fl.layout == RRTMGP_LAYOUT_NCOL_NLEV ? (size_t)col + (size_t)fl.ld * lev
                                                               : (size_t)lev + (size_t)nlev * col;
        if (set == 0) {
            fl.up[o] = up; fl.dn[o] = dn; fl.net[o] = net;
            if (ncomp == 3 && fl.dir) fl.dir[o] = dir;
        } else {
            fl.clear_up[o] = up; fl.clear_dn[o] = dn; fl.clear_net[o] = net;
            if (ncomp == 3 && fl.clear_dir) fl.clear_dir[o] = dir;
        }
    }
    // FluxBand (Fluxes.jl:170-215): band b owns the rows [band_row_lo[b], band_row_lo[b + 1]); scaled like
    // the broadband fluxes (Fluxes.jl:448-454), net from the scaled values (update_fluxes.jl:198-201)
    if (fl.band_up) {
        for (int i = threadIdx.x; i < d.nbnd * nlev; i += blockDim.x) {
            const int b = i / nlev, lev = i - b * nlev;
            FT up = FT(0), dn = FT(0);
            if (!zero) {
                const int r0 = lk.band_row_lo[b], r1 = lk.band_row_lo[b + 1];
                for (int r = r0; r < r1; r++) {
                    up += sh.acc[((size_t)r * nlev + lev) * d.n_acc];
                    dn += sh.acc[((size_t)r * nlev + lev) * d.n_acc + 1];
                }
            }
            if (fl.metric) {
                const FT m = fl.metric[(size_t)nlev * col + lev];
                up *= m; dn *= m;
            }
            const size_t o = ((size_t)b * fl.band_ncol + col) * nlev + lev;
            fl.band_up[o] = up; fl.band_dn[o] = dn;
            if (fl.band_net) fl.band_net[o] = up - dn;
        }
    }
}

}  // namespace rrtmgp
