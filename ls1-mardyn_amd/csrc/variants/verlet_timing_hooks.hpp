// TIMING VARIANTS of the list force pass (kernels_force_verlet.hip) — NEVER part of the shipped library.
//
// Included only under -DLS1_BUILD_VARIANT, which tools/ab_variant.sh passes (and nothing else does: the regular Makefile has no
// such flag, a stray variant switch without it is a compile error, and a library that contains a variant object reports
// ls1hip_get_option("build_variant") == 1 and "+variant" in ls1hip_version — tests/test_abi_cpu.py checks the shipped one).
// Every switch below removes a PHASE of the workgroup's life to time the others (profiles/r3_ab_force_pass_decomposition.txt,
// profiles/r4_force_pass_phases.txt); forces are WRONG BY CONSTRUCTION and are scaled by a run-time zero so that the molecules of
// a bench run keep moving ballistically instead of blowing up.
//   -DLS1_NOLOOP_MOCK   everything but the pair loop (staging, list head, fused epilogue, reductions)
//   -DLS1_NOEPI_MOCK    no fused epilogue (no velocity loads, no position / velocity stores)
//   -DLS1_NOSTAGE_MOCK  no staging of the region into LDS (the pair loop reads whatever the LDS holds)
// -DLS1_POS_AOS (x y z of a staged molecule side by side) is a layout variant with correct results, handled in the kernel file.
#pragma once

#if defined(LS1_NOEPI_MOCK)
#define LS1_HOOK_EPILOGUE(P) ((P).which != 0)  // run-time false in the bench's single-pass traversal
#else
#define LS1_HOOK_EPILOGUE(P) true
#endif

#if defined(LS1_NOSTAGE_MOCK)
#define LS1_HOOK_STAGING(P) ((P).which != 0)   // run-time false
#else
#define LS1_HOOK_STAGING(P) true
#endif

#if defined(LS1_NOLOOP_MOCK)
#define LS1_HOOK_LAST_ROW(nw) (3u)
#define LS1_HOOK_ROWS(nw) (0u)
#else
#define LS1_HOOK_LAST_ROW(nw) ((nw) - 1u)
#define LS1_HOOK_ROWS(nw) (nw)
#endif

#if defined(LS1_NOLOOP_MOCK) || defined(LS1_NOSTAGE_MOCK)
#define LS1_HOOK_FORCE(f, P) ((f) * (double)(P).which)  // 0 in the single-pass traversal the bench runs
#else
#define LS1_HOOK_FORCE(f, P) (f)
#endif
