/*
 * ntsm_hooks.h -- the only place where the two non-product build flavours touch the product sources.
 *
 *   NTSM_ABLATION  (tools/ablate.sh builds; such libraries count WRONGLY by construction and are never shipped): the hook
 *                  macros below are defined by ntsm_ablation.inc.  In every other build they expand to nothing (or to the
 *                  plain expression), so the shipped kernels carry no ablation code -- tools/isa_hash.py checks that the ISA
 *                  of the product kernels does not depend on this header's existence.
 *   NTSM_WITH_TAB  (`make tab`: ntsm_amd/libntsm_hip_tab.so, the tabulated k = 19 kernel of DESIGN.md section 4.3, a measured
 *                  negative result kept reproducible): kWithTab switches the list mode of the minimizer-blocked kernel on.
 */
#ifndef NTSM_HOOKS_H
#define NTSM_HOOKS_H

#ifdef NTSM_WITH_TAB
constexpr bool kWithTab = true;
#else
constexpr bool kWithTab = false;
#endif

#ifdef NTSM_ABLATION
#include "ntsm_ablation.inc"
#else
/* device side (kernels_mz.hip) */
#define NTSM_ABL_UNLESS_NO_ATOMICS(p_)
#define NTSM_ABL_STAGE2(v_, p_)
#define NTSM_ABL_STAGE1(v_, p_)
#define NTSM_ABL_BLOCK_INDEX(p_, mz_, ld_, bi_) (__builtin_amdgcn_inverse_ballot_w64(ld_) ? (bi_) : 0xFFFFFFFFu)
/* host side (tables.cpp, runtime.cpp) */
#define NTSM_ABL_BLOCKS_BUILT(blocks_)
#define NTSM_ABL_PREFILTER_LOG2(pl_)
#define NTSM_ABL_PREFILTER_BUILT(prefilter_)
#define NTSM_ABL_LAUNCH_PARAMS(p_)
#endif

#endif
