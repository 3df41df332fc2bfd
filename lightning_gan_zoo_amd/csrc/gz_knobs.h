// Every run-time switch of the kernel library, in ONE struct that is filled ONCE (first use).
//
// Product behaviour never depends on the environment: the struct holds the shipped defaults unless the process was
// started with GZ_EXPERIMENTS=1, in which case each field may be overridden by the GZ_* variable named next to it
// (the measurement helpers under tools/ set both).  The launchers read `knobs().field`; no launcher calls getenv.
// The two settings a product caller may change go through the C ABI instead (include/gz_ops.h): gz_set_cu_budget.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <atomic>

namespace gz {

// X(type, field, environment variable, default)
#define GZ_KNOB_LIST(X)                                                                                               \
    /* tile / skeleton choice */                                                                                      \
    X(bool, no_igemm2, "GZ_NO_IGEMM2", false)               /* keep everything on igemm_kernel (round 2) */         \
    X(bool, no_igemm2_tap, "GZ_NO_IGEMM2_TAP", false)       /* ... the gather-loader launches only */               \
    X(bool, no_plane_a, "GZ_NO_PLANE_A", false)             /* 1x1 layers: the 4-byte gather instead of PlaneA2 */   \
    X(bool, no_wg3r, "GZ_NO_WG3R", false)                   /* 3-D weight gradient: the round-1 kernel */            \
    X(int, wg3r_wgs, "GZ_WG3R_WGS", 2)                      /* ... workgroups per CU aimed at by its split */        \
    X(int, wg3r_min_chunks, "GZ_WG3R_MIN_CHUNKS", 24)       /* ... chunks per workgroup at least */                  \
    X(bool, no_fewk_wg, "GZ_NO_FEWK_WG", false)             /* 3x3 Wg with <= 4 output channels: the MFMA kernel */  \
    X(bool, no_fewk_conv, "GZ_NO_FEWK_CONV", false)         /* 3x3 F with <= 4 output channels: the MFMA kernel */   \
    X(int, dg3_tile, "GZ_DG3_TILE", 0)                      /* ConvTranspose3d forward on igemm2: force 256x128 (1) / 256x64 (2) */ \
    X(int, dg3_wgs, "GZ_DG3_WGS", 2)                        /* ... split launches aim at this many workgroups per CU */ \
    X(int, dg3_min_chunks, "GZ_DG3_MIN_CHUNKS", 32)         /* ... of at least this many chunks */                   \
    X(bool, no_igemm2_wg, "GZ_NO_IGEMM2_WG", false)         /* ... the weight gradients only */                     \
    X(bool, no_igemm2w, "GZ_NO_IGEMM2W", false)             /* register-staged instead of LDS-DMA weight gradient */ \
    X(bool, no_igemm2wg, "GZ_NO_IGEMM2WG", false)           /* ... its generic-geometry image only */               \
    X(int, igemm2_tile, "GZ_IGEMM2_TILE", 0)                /* 128 / 256: force the igemm2 N tile */                 \
    X(bool, no_tile64, "GZ_NO_TILE64", false)               /* round 4's 256x64 igemm2 tile off */                   \
    X(bool, no_kg2, "GZ_NO_KG2", false)                     /* round 5's two wave groups per workgroup off */        \
    X(bool, no_pool_plane, "GZ_NO_POOL_PLANE", false)       /* evaluation path: the flat pooling kernel (round 5) */        \
    X(bool, no_pool_group, "GZ_NO_POOL_GROUP", false)       /* evaluation path: no compile-time-window pooling */           \
    X(bool, no_any2, "GZ_NO_ANY2", false)                   /* evaluation path: run-time geometries on igemm_kernel only (round 5) */ \
    X(bool, no_dg5, "GZ_NO_DG5", false)                     /* 5x5 s2 p2 input gradients: the gather loader (round 5) */ \
    X(int, dg5_wgs, "GZ_DG5_WGS", 512)                      /* ConvDg5A2: workgroups a split launch aims at */        \
    X(int, dg5_min_chunks, "GZ_DG5_MIN_CHUNKS", 24)         /* ... and the shortest reduction piece it accepts */    \
    X(int, dg5_min_units, "GZ_DG5_MIN_UNITS", 80)            /* ... tiles x chunks per CU below which the small tiles keep the launch */ \
    X(int, kg2_min_chunks, "GZ_KG2_MIN_CHUNKS", 32)         /* ... chunks per workgroup at least */                  \
    X(int, tap64_min_chunks, "GZ_TAP64_MIN_CHUNKS", 32)     /* gather-loader launches too small for 256x128 take 256x64 tiles with >= this many chunks per piece (0: off) */ \
    X(int, tile, "GZ_TILE", -1)                             /* 0..3: force the igemm_kernel tile */                  \
    X(int, min_wgs, "GZ_MIN_WGS", 0)                        /* > 0: the round-1 tile rule with this target */        \
    X(bool, no_big_split, "GZ_NO_BIG_SPLIT", false)                                                                   \
    X(bool, no_splitk, "GZ_NO_SPLITK", false)                                                                         \
    X(int, split_target, "GZ_SPLIT_TARGET", 1024)                                                                     \
    X(int, split_below, "GZ_SPLIT_BELOW", 512)                                                                        \
    X(bool, no_tapmajor, "GZ_NO_TAPMAJOR", false)                                                                     \
    X(bool, no_row4, "GZ_NO_ROW4", false)                                                                             \
    X(int, fwd2_min_tiles, "GZ_FWD2_MIN_TILES", 16)                                                                   \
    X(int, dg2_min_tiles, "GZ_DG2_MIN_TILES", 64)                                                                     \
    X(int, tap_wgs, "GZ_TAP_WGS", 384)                                                                                \
    X(int, tap_cps_max, "GZ_TAP_CPS_MAX", 96)                                                                         \
    X(int, wg_target, "GZ_WG_TARGET", 1024)                                                                           \
    X(int, wg2_min_chunks, "GZ_WG2_MIN_CHUNKS", 0)                                                                    \
    X(bool, wg_generic, "GZ_WG_GENERIC", false)                                                                       \
    /* direct kernels */                                                                                              \
    X(bool, no_smallc, "GZ_NO_SMALLC", false)                                                                         \
    X(bool, no_smallc5, "GZ_NO_SMALLC5", false)                                                                       \
    X(bool, smallc_one_pos, "GZ_SMALLC_ONE_POS", false)                                                               \
    X(long long, smallc_split_below, "GZ_SMALLC_SPLIT_BELOW", 192 * 1024)                                             \
    X(long long, smallc_split8_below, "GZ_SMALLC_SPLIT8_BELOW", 36 * 1024)                                            \
    X(int, smallc_ks, "GZ_SMALLC_KS", 0)                                                                              \
    X(bool, no_fewc_wg, "GZ_NO_FEWC_WG", false)                                                                       \
    X(bool, no_act_fuse, "GZ_NO_ACT_FUSE", false)                                                                     \
    X(bool, no_pack_k4, "GZ_NO_PACK_K4", false)                                                                       \
    X(bool, no_resample_swizzle, "GZ_NO_RESAMPLE_SWIZZLE", false)                                                     \
    X(int, fewc_wg_blocks, "GZ_FEWC_WG_BLOCKS", 512)                                                                  \
    X(bool, no_smallch_conv, "GZ_NO_SMALLCH_CONV", false)                                                             \
    X(bool, no_smallch_wg, "GZ_NO_SMALLCH_WG", false)                                                                 \
    X(int, c3_gpw, "GZ_C3_GPW", 4)                                                                                    \
    /* launch shape experiments */                                                                                    \
    X(bool, no_xcd_swizzle, "GZ_NO_XCD_SWIZZLE", false)                                                               \
    X(int, dyn_lds, "GZ_DYN_LDS", 0)                                                                                  \
    X(int, igemm2_stagger, "GZ_IGEMM2_STAGGER", 0)                                                                    \
    X(int, igemm2_lds, "GZ_IGEMM2_LDS", 0)                                                                            \
    /* 3-D convolutions, normalisation, resampling */                                                                 \
    X(bool, dg3_even_split, "GZ_DG3_EVEN_SPLIT", false)                                                               \
    X(int, wg3_target, "GZ_WG3_TARGET", 1024)                                                                         \
    X(int, wg3_tile, "GZ_WG3_TILE", -1)                                                                               \
    X(bool, norm_unfused, "GZ_NORM_UNFUSED", false)                                                                   \
    X(int, norm_bwd_cache, "GZ_NORM_BWD_CACHE", 4)                                                                    \
    X(bool, resample_fwd_direct, "GZ_RESAMPLE_FWD_DIRECT", false)                                                     \
    X(int, resample_bwd_mode, "GZ_RESAMPLE_BWD", 0)         /* 1 "scatter" (round 1), 2 "gather" */

struct Knobs {
    bool experiments;       // GZ_EXPERIMENTS=1: the fields below were read from the environment
#define GZ_KNOB_FIELD(type, name, env, dflt) type name;
    GZ_KNOB_LIST(GZ_KNOB_FIELD)
#undef GZ_KNOB_FIELD
};

inline void knob_read(bool& v, const char* e) { v = e != nullptr; }
inline void knob_read(int& v, const char* e) {
    if (!e) return;
    if (!strcmp(e, "scatter")) v = 1;
    else if (!strcmp(e, "gather")) v = 2;
    else v = atoi(e);
}
inline void knob_read(long long& v, const char* e) { if (e) v = atoll(e); }

inline const Knobs& knobs() {
    static const Knobs k = [] {
        Knobs n;
#define GZ_KNOB_INIT(type, name, env, dflt) n.name = dflt;
        GZ_KNOB_LIST(GZ_KNOB_INIT)
#undef GZ_KNOB_INIT
        const char* on = getenv("GZ_EXPERIMENTS");
        n.experiments = on && on[0] && on[0] != '0';
        if (n.experiments) {
#define GZ_KNOB_ENV(type, name, env, dflt) knob_read(n.name, getenv(env));
            GZ_KNOB_LIST(GZ_KNOB_ENV)
#undef GZ_KNOB_ENV
        }
        return n;
    }();
    return k;
}

// CU budget (round 4): how many of the chip's 256 CUs the convolution plans may count on.  Every split / tile plan is
// sized to whole ROUNDS of workgroup slots (2 igemm2 workgroups or 4 igemm workgroups per CU); when part of the chip
// is taken -- RCCL's channel kernels run underneath backward once the gradient exchange overlaps it -- a plan sized
// for 256 CUs needs a second, nearly empty round.  ddp.GradSync sets the budget to 256 - (channels it expects) through
// gz_set_cu_budget; a single-GPU process keeps 256.
inline std::atomic<int>& cu_budget_ref() {
    static std::atomic<int> v{256};
    return v;
}
inline int cus() { return cu_budget_ref().load(std::memory_order_relaxed); }

}  // namespace gz
