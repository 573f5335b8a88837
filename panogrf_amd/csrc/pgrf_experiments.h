// Experiment switches of the kernel sources (ablations "timing only, results are then wrong", alternative policies,
// instrumentation) are compile-time -D flags.  NONE of them may reach the product library: a translation unit that sees
// one without -DPGRF_ABLATION_BUILD does not compile, and a library linked from any PGRF_ABLATION_BUILD object reports
// a NEGATIVE pgrf_abi_version(), which panogrf_amd/_lib.py refuses to load unless PGRF_ALLOW_ABLATION_LIB=1 is set
// (tools/build_variant*.sh build such libraries under their own file names, libpanogrf_hip_<NAME>.so).
// Include this header first in every kernel translation unit (before the defaults of the tunables are defined).
#pragma once

#if defined(PGRF_ABL_ELU) || defined(PGRF_ABL_GATHER) || defined(PGRF_ABL_WEIGHTS) || defined(PGRF_ABL_WLDS) || defined(PGRF_NO_W16) || defined(PGRF_STAGE_MARK) || defined(PGRF_MID_FENCE) || defined(PGRF_GATHER_QUAD_MAJOR) || defined(PGRF_DD_FENCE) || defined(PGRF_NO_BARRIER) ||             \
    defined(PGRF_STAGE_TIMING) || defined(PGRF_NO_DEDUPE) || defined(PGRF_NO_FULL_IMG) || defined(PGRF_NO_LAZY_TAPS) ||        \
    defined(PGRF_NO_LDS_IMG) || defined(PGRF_NO_PARK) || defined(PGRF_NO_RECOMPUTE_NF) || defined(PGRF_NO_SMALL) ||           \
    defined(PGRF_NO_STAGE_FENCE) || defined(PGRF_NO_XCD_REMAP) || defined(PGRF_NT_ALL) || defined(PGRF_NT_PAIRS) ||           \
    defined(PGRF_ONE_WG_PER_CU) || defined(PGRF_PV_MFMA) || defined(PGRF_PV_PACKED) || defined(PGRF_TWO_WAVE_MAX_V) ||        \
    defined(PGRF_FENCE_MASK) || defined(PGRF_GROUP_FENCE) || defined(PGRF_LDS_STATE) || defined(PGRF_SMALL_V4) || defined(PGRF_BATCH_GATHER) || defined(PGRF_A_PREFETCH) || defined(PGRF_PAIR_NO_HEAD_SPLIT) ||                                                              \
    defined(PGRF_CONV_ABL_A) || defined(PGRF_CONV_ABL_B) || defined(PGRF_CONV_ABL_NOISSUE) ||                                 \
    defined(PGRF_CONV_ABL_NOREFILL_A) || defined(PGRF_CONV_ABL_NOREFILL_B) || defined(PGRF_CONV_NO_FAST) ||                   \
    defined(PGRF_CONV_NO_FUSE) || defined(PGRF_CONV_SPREAD_REFILL) || defined(PGRF_UPF_ABL) || defined(PGRF_CONV_WAVES) ||                             \
    defined(PGRF_CV_ABL_GATHER) || defined(PGRF_CV_ABL_GEOM) || defined(PGRF_CV_ABL_STORE) || defined(PGRF_CV_MINITEMS) ||    \
    defined(PGRF_CV_WAVES) ||                                                                                                \
    defined(PGRF_WINO_ABL_NOA) || defined(PGRF_WINO_ABL_NOADD) || defined(PGRF_WINO_ABL_NOALOAD) ||                           \
    defined(PGRF_TRAIN_ABL_NO_DW) || defined(PGRF_TRAIN_ABL_NO_STAGE) || defined(PGRF_STAT_PTS) || defined(PGRF_SCATTER_GE) || defined(PGRF_SCATTER_GB) || defined(PGRF_SCATTER_GW) || defined(PGRF_TRAIN_ABL_NO_DELTA) || defined(PGRF_WINO_ABL_NOB) || defined(PGRF_WINO_ABL_NOLDS) || defined(PGRF_WINO_TIMING) || defined(PGRF_WINO_TV_PINGPONG)
#ifndef PGRF_ABLATION_BUILD
#error "experiment switch (-DPGRF_ABL_* / PGRF_NO_* / PGRF_*_TIMING / tunable override) without -DPGRF_ABLATION_BUILD: use tools/build_variant*.sh, never the product Makefile"
#endif
#endif

#ifdef PGRF_ABLATION_BUILD
// weak DEFINITION, one per experiment object; capi.hip holds a weak REFERENCE and turns its presence into a negative ABI version
extern "C" __attribute__((weak, visibility("default"))) int pgrf_ablation_marker = 1;
#endif
