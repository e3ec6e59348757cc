// variants.h — every compile-time switch of the kernels, in one place.
//
// The shipped library is built with none of them (Makefile: `make`): IEEE-accurate Float32 forms, device.h.  The
// raw-instruction Float32 build takes RR_FAST_F32 alone (`make fast` -> libhip_rrtmgp_fast.so); RR_LIBM_F64 (experiment)
// puts libm and the compiler's division back under the Float64 forms.  What is left besides that are TUNABLES with a shipped default.  A build with another value is an
// experiment: it compiles only with -DRR_EXPERIMENTS (`make variant` adds it) and the library says what it was built with
// (rrtmgp_hip_build_flags, also appended to rrtmgp_hip_version), so an experimental .so cannot pass for the shipped one:
// tests/test_abi.py asserts that the shipped library reports no flags.
// The timing-only switches of rounds 1-3 (RR_EXP_*: kernels that give wrong results by construction, to bound what a
// restructuring could gain) are no longer in the sources: their results are in tools/experiments/README.md and the code is
// tools/experiments/timing_switches_r01_r03.patch.
#pragma once

#ifndef RR_MIN_WAVES
#define RR_MIN_WAVES 4  // waves per SIMD the Float32 column kernels are register-allocated for
#endif
#ifndef RR_DIAG_MIN_WAVES  // ... the Float32 clear-sky-diagnostic variants
#define RR_DIAG_MIN_WAVES 3
#endif
#ifndef RR_F64_HALF_WAVES  // ... the Float64 instances with 8-layer chunks (2 = as the 16-layer ones: 256 VGPRs; 3 = 168 VGPRs)
#define RR_F64_HALF_WAVES 2
#endif
#ifndef RR_F64_HALF_CHUNK  // ... and their layers per chunk of LDS records
#define RR_F64_HALF_CHUNK 8
#endif
#ifndef RR_SW_HALF_WAVES  // ... the Float32 shortwave main instances with 8-layer chunks (5 = 96 VGPRs: a fifth workgroup per CU; experiment)
#define RR_SW_HALF_WAVES RR_MIN_WAVES
#endif
#ifndef RR_ACC_ATOMIC  // SW layer loop: g-point sums by 4 DPP steps + one LDS add (1) or 6 DPP steps + store (0)
#define RR_ACC_ATOMIC 1
#endif

#ifndef RR_SWEEP_NT  // sweep scratch cache policy: 0 = default; 1 = non-temporal LOADS in the second sweep; 2 = non-temporal loads AND stores
#define RR_SWEEP_NT 0
#endif

#define RR_STR2(x) #x
#define RR_STR(x) RR_STR2(x)
#ifdef RR_PRECISE_F32
#error "RR_PRECISE_F32 is gone: the IEEE-accurate Float32 forms are the default build since round 6 (the opt-in is RR_FAST_F32)"
#endif
#ifdef RR_FAST_F32
#define RR_BUILD_FLAGS_PRECISE " RR_FAST_F32"
#else
#define RR_BUILD_FLAGS_PRECISE ""
#endif
#ifdef RR_LIBM_F64
#define RR_HAS_RR_LIBM_F64 " RR_LIBM_F64"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_LIBM_F64 ""
#endif
#if RR_MIN_WAVES != 4
#define RR_HAS_RR_MIN_WAVES " RR_MIN_WAVES=" RR_STR(RR_MIN_WAVES)
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_MIN_WAVES ""
#endif
#if RR_DIAG_MIN_WAVES != 3
#define RR_HAS_RR_DIAG_MIN_WAVES " RR_DIAG_MIN_WAVES=" RR_STR(RR_DIAG_MIN_WAVES)
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_DIAG_MIN_WAVES ""
#endif
#if RR_ACC_ATOMIC != 1
#define RR_HAS_RR_ACC_ATOMIC " RR_ACC_ATOMIC=" RR_STR(RR_ACC_ATOMIC)
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_ACC_ATOMIC ""
#endif
#if RR_F64_HALF_WAVES != 2
#define RR_HAS_RR_F64_HALF_WAVES " RR_F64_HALF_WAVES=" RR_STR(RR_F64_HALF_WAVES)
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_F64_HALF_WAVES ""
#endif
#if RR_F64_HALF_CHUNK != 8
#define RR_HAS_RR_F64_HALF_CHUNK " RR_F64_HALF_CHUNK=" RR_STR(RR_F64_HALF_CHUNK)
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_F64_HALF_CHUNK ""
#endif
#if RR_SW_HALF_WAVES != RR_MIN_WAVES
#define RR_HAS_RR_SW_HALF_WAVES " RR_SW_HALF_WAVES=" RR_STR(RR_SW_HALF_WAVES)
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_SW_HALF_WAVES ""
#endif
#if RR_SWEEP_NT != 0
#define RR_HAS_RR_SWEEP_NT " RR_SWEEP_NT=" RR_STR(RR_SWEEP_NT)
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_SWEEP_NT ""
#endif
#define RR_BUILD_FLAGS (RR_HAS_RR_SWEEP_NT RR_BUILD_FLAGS_PRECISE RR_HAS_RR_MIN_WAVES RR_HAS_RR_DIAG_MIN_WAVES RR_HAS_RR_ACC_ATOMIC RR_HAS_RR_F64_HALF_WAVES RR_HAS_RR_F64_HALF_CHUNK RR_HAS_RR_LIBM_F64 RR_HAS_RR_SW_HALF_WAVES)
#if defined(RR_ANY_EXPERIMENT) && !defined(RR_EXPERIMENTS)
#error "non-default tuning values are experiments: build them with `make variant NAME=... EXTRA=...` (adds -DRR_EXPERIMENTS), never into the shipped library"
#endif
