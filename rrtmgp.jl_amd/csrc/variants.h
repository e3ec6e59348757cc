// variants.h — every compile-time switch of the kernels, in one place.
//
// The shipped library is built with none of them (Makefile: `make`), the IEEE-Float32 build with RR_PRECISE_F32 alone
// (`make precise`).  Everything else is an EXPERIMENT: A/B alternatives, and timing-only switches that give WRONG RESULTS
// by construction (they exist to bound what a restructuring could gain; tools/experiments/README.md).  An experiment
// compiles only with -DRR_EXPERIMENTS (`make variant` adds it) and the library says what it was built with
// (rrtmgp_hip_build_flags, also appended to rrtmgp_hip_version), so an experimental .so cannot pass for the shipped one:
// tests/test_abi.py asserts that the shipped library reports no flags.  A new switch must be listed here
// (tests/test_abi.py greps the sources for RR_EXP_* / RR_SCRATCH_* / RR_PREP_* names).
#pragma once

// tunables with a shipped default
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
#ifndef RR_ACC_ATOMIC  // SW layer loop: g-point sums by 4 DPP steps + one LDS add (1) or 6 DPP steps + store (0)
#define RR_ACC_ATOMIC 1
#endif

#define RR_STR2(x) #x
#define RR_STR(x) RR_STR2(x)
#define RR_BUILD_FLAGS_PRECISE ""
#ifdef RR_PRECISE_F32
#undef RR_BUILD_FLAGS_PRECISE
#define RR_BUILD_FLAGS_PRECISE " RR_PRECISE_F32"
#endif
#ifdef RR_EXP_NO_CHUNK_BARRIER
#define RR_HAS_RR_EXP_NO_CHUNK_BARRIER " RR_EXP_NO_CHUNK_BARRIER"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_NO_CHUNK_BARRIER ""
#endif
#ifdef RR_EXP_NO_LAYER_SUMS
#define RR_HAS_RR_EXP_NO_LAYER_SUMS " RR_EXP_NO_LAYER_SUMS"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_NO_LAYER_SUMS ""
#endif
#ifdef RR_EXP_PREP_ONCE
#define RR_HAS_RR_EXP_PREP_ONCE " RR_EXP_PREP_ONCE"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_PREP_ONCE ""
#endif
#ifdef RR_EXP_PREP_SAME_LANES
#define RR_HAS_RR_EXP_PREP_SAME_LANES " RR_EXP_PREP_SAME_LANES"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_PREP_SAME_LANES ""
#endif
#ifdef RR_PREP_KK_MINOR
#define RR_HAS_RR_PREP_KK_MINOR " RR_PREP_KK_MINOR"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_PREP_KK_MINOR ""
#endif
#ifdef RR_EXP_NO_MINOR
#define RR_HAS_RR_EXP_NO_MINOR " RR_EXP_NO_MINOR"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_NO_MINOR ""
#endif
#ifdef RR_EXP_ZERO_G1
#define RR_HAS_RR_EXP_ZERO_G1 " RR_EXP_ZERO_G1"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_ZERO_G1 ""
#endif
#ifdef RR_EXP_MINOR_ONE_GROUP
#define RR_HAS_RR_EXP_MINOR_ONE_GROUP " RR_EXP_MINOR_ONE_GROUP"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_MINOR_ONE_GROUP ""
#endif
#ifdef RR_EXP_REFILL_SELECT
#define RR_HAS_RR_EXP_REFILL_SELECT " RR_EXP_REFILL_SELECT"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_REFILL_SELECT ""
#endif
#ifdef RR_EXP_MASK_128_ONLY
#define RR_HAS_RR_EXP_MASK_128_ONLY " RR_EXP_MASK_128_ONLY"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_MASK_128_ONLY ""
#endif
#ifdef RR_EXP_SCRATCH_ROW0
#define RR_HAS_RR_EXP_SCRATCH_ROW0 " RR_EXP_SCRATCH_ROW0"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_SCRATCH_ROW0 ""
#endif
#ifdef RR_EXP_SCRATCH_ROW0_STORES
#define RR_HAS_RR_EXP_SCRATCH_ROW0_STORES " RR_EXP_SCRATCH_ROW0_STORES"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_SCRATCH_ROW0_STORES ""
#endif
#ifdef RR_EXP_SCRATCH_ROW0_LOADS
#define RR_HAS_RR_EXP_SCRATCH_ROW0_LOADS " RR_EXP_SCRATCH_ROW0_LOADS"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_SCRATCH_ROW0_LOADS ""
#endif
#ifdef RR_SCRATCH_NT_STORE
#define RR_HAS_RR_SCRATCH_NT_STORE " RR_SCRATCH_NT_STORE"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_SCRATCH_NT_STORE ""
#endif
#ifdef RR_SCRATCH_NT_LOAD
#define RR_HAS_RR_SCRATCH_NT_LOAD " RR_SCRATCH_NT_LOAD"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_SCRATCH_NT_LOAD ""
#endif
#ifdef RR_EXP_SW_256
#define RR_HAS_RR_EXP_SW_256 " RR_EXP_SW_256"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_SW_256 ""
#endif
#ifdef RR_EXP_SCRATCH_X3
#define RR_HAS_RR_EXP_SCRATCH_X3 " RR_EXP_SCRATCH_X3"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_SCRATCH_X3 ""
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
#ifdef RR_EXP_LW_K_ONLY
#define RR_HAS_RR_EXP_LW_K_ONLY " RR_EXP_LW_K_ONLY"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_LW_K_ONLY ""
#endif
#ifdef RR_EXP_NO_MINOR_TAIL
#define RR_HAS_RR_EXP_NO_MINOR_TAIL " RR_EXP_NO_MINOR_TAIL"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_NO_MINOR_TAIL ""
#endif
#ifdef RR_EXP_FAKE_ROWS
#define RR_HAS_RR_EXP_FAKE_ROWS " RR_EXP_FAKE_ROWS"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_FAKE_ROWS ""
#endif
#ifdef RR_EXP_AERO_CONST
#define RR_HAS_RR_EXP_AERO_CONST " RR_EXP_AERO_CONST"
#define RR_ANY_EXPERIMENT 1
#else
#define RR_HAS_RR_EXP_AERO_CONST ""
#endif
#define RR_BUILD_FLAGS (RR_HAS_RR_F64_HALF_CHUNK RR_HAS_RR_EXP_AERO_CONST RR_HAS_RR_EXP_FAKE_ROWS RR_HAS_RR_EXP_NO_MINOR_TAIL RR_HAS_RR_EXP_LW_K_ONLY RR_BUILD_FLAGS_PRECISE RR_HAS_RR_EXP_NO_CHUNK_BARRIER RR_HAS_RR_EXP_NO_LAYER_SUMS RR_HAS_RR_EXP_PREP_ONCE RR_HAS_RR_EXP_PREP_SAME_LANES RR_HAS_RR_PREP_KK_MINOR RR_HAS_RR_EXP_NO_MINOR RR_HAS_RR_EXP_ZERO_G1 RR_HAS_RR_EXP_MINOR_ONE_GROUP RR_HAS_RR_EXP_REFILL_SELECT RR_HAS_RR_EXP_MASK_128_ONLY RR_HAS_RR_EXP_SCRATCH_ROW0 RR_HAS_RR_EXP_SCRATCH_ROW0_STORES RR_HAS_RR_EXP_SCRATCH_ROW0_LOADS RR_HAS_RR_SCRATCH_NT_STORE RR_HAS_RR_SCRATCH_NT_LOAD RR_HAS_RR_EXP_SW_256 RR_HAS_RR_EXP_SCRATCH_X3 RR_HAS_RR_MIN_WAVES RR_HAS_RR_DIAG_MIN_WAVES RR_HAS_RR_ACC_ATOMIC RR_HAS_RR_F64_HALF_WAVES)
#if defined(RR_ANY_EXPERIMENT) && !defined(RR_EXPERIMENTS)
#error "RR_EXP_* / tuning switches are experiments: build them with `make variant NAME=... EXTRA=...` (adds -DRR_EXPERIMENTS), never into the shipped library"
#endif
