// TIMING VARIANTS of the list force pass (kernels_force_verlet.hip) — NEVER part of the shipped library.
//
// Included only under -DLS1_BUILD_VARIANT, which tools/ab_variant.sh passes (and nothing else does: the regular Makefile has no
// such flag, a stray variant switch without it is a compile error, and a library that contains a variant object reports
// ls1hip_get_option("build_variant") == 1 and "+variant" in ls1hip_version — tests/test_abi_cpu.py checks the shipped one).
// Every switch below removes a PHASE of the workgroup's life to time the others (profiles/r3_ab_force_pass_decomposition.txt,
// profiles/r4_force_pass_phases.txt); forces are WRONG BY CONSTRUCTION and are replaced by a run-time zero so that the molecules of
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
#define LS1_HOOK_FORCE(f, P) ((P).which != 0 ? (f) : 0.0)  // exactly 0 in the single-pass traversal the bench runs (a select: a NaN from unstaged LDS cannot get through)
#else
#define LS1_HOOK_FORCE(f, P) (f)
#endif

// ---- round 4: what would a shorter memory chain be worth? ------------------------------------------------------------------------
//   -DLS1_X_LATE_V       the epilogue's velocity loads are issued behind the pair loop instead of ahead of it (correct results)
//   -DLS1_X_NT_STORE     the fused epilogue's r', v' stores are non-temporal (correct results)
//   -DLS1_WARM_MOCK      every brick stages the region of one of 256 bricks (record + positions then come from L2 / MALL): the
//                        upper bound of what prefetching the next brick's record and region into the cache could gain (WRONG forces)
#if defined(LS1_X_LATE_V)
#define LS1_HOOK_EARLY_V false
#else
#define LS1_HOOK_EARLY_V true
#endif
#if defined(LS1_X_NT_STORE)
#define LS1_HOOK_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define LS1_HOOK_STORE(ptr, val) (*(ptr) = (val))
#endif
#if defined(LS1_WARM_MOCK)
#define LS1_HOOK_RECORD_OF(did) ((did) & 255u)
#define LS1_HOOK_OWN_FROM_LDS false  // (the LDS holds another brick's region: the molecule's own position comes from global memory)
#undef LS1_HOOK_FORCE
#define LS1_HOOK_FORCE(f, P) ((P).which != 0 ? (f) : 0.0)
#else
#define LS1_HOOK_RECORD_OF(did) (did)
#define LS1_HOOK_OWN_FROM_LDS true
#endif

//   -DLS1_X_REGSTAGE     the region of a regular brick is staged through registers (global_load + ds_write, the round-3 form) instead
//                        of by LDS-DMA (correct results; profiles/r4_ab_lds_dma_staging.txt)
#if defined(LS1_X_REGSTAGE)
#define LS1_HOOK_DMA_STAGING false
#else
#define LS1_HOOK_DMA_STAGING true
#endif

//   -DLS1_X_STAGGER      the workgroups of the FIRST round (the first 512: two per CU) start with a pseudo-random delay of 0-24 us (one
//                        workgroup lifetime): do the two workgroups of a CU run their memory and pair-loop phases in lockstep?  (correct results)
#if defined(LS1_X_STAGGER)
#define LS1_HOOK_STAGGER()                                                                    \
	do {                                                                                      \
		if (blockIdx.x < 512u) {                                                              \
			const uint32_t h_ = (blockIdx.x * 2654435761u) >> 29;                             \
			for (uint32_t i_ = 0; i_ < h_; ++i_) __builtin_amdgcn_s_sleep(127);               \
		}                                                                                     \
	} while (0)
#else
#define LS1_HOOK_STAGGER() ((void)0)
#endif
