// Part of the fp32 implicit-GEMM core (see gz_igemm.h): buffer-load primitives, the tile configuration and the
// operand loaders of the round-1/2 skeleton (generic matrices, 2-D / 3-D convolution geometries, weight-gradient rows).
#pragma once
#include "gz_common.h"
#include "gz_knobs.h"
#include <type_traits>

namespace gz {

constexpr int NT = 256;
constexpr int BK = 16;
// number of kernel taps k = ((parity + P) % S) + S*t below KS that a transposed-conv output phase of that parity has
__host__ __device__ constexpr int dg_taps(int KS, int S, int P, int parity) {
    return (KS - ((parity + P) % S) + S - 1) / S;
}

constexpr uint32_t OOB = 0x80000000u;  // voffset that is out of range for every tensor (< 2 GiB)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// LDS-DMA (buffer_load ... lds): the wave's 64 lanes land at lds_base + lane*size, no VGPR staging and no
// ds_write; lds_base must be wave-uniform.  Out-of-range lanes write 0.
#ifndef GZ_IGEMM_NO_DMA
#define GZ_IGEMM_DMA 1
#else
#define GZ_IGEMM_DMA 0
#endif
__device__ __forceinline__ void bload_lds4(__amdgpu_buffer_rsrc_t r, float* lds_wave_base, uint32_t voff,
                                           uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, voff, soff, 0,
                                             0);
}
__device__ __forceinline__ void bload_lds16(__amdgpu_buffer_rsrc_t r, float* lds_wave_base, uint32_t voff,
                                            uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff,
                                             0, 0);
}

template <int WM_, int WN_, int TM_, int TN_>
struct TileCfg {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
};

// ---------------------------------------------------------------------------
// generic matrix loaders
// ---------------------------------------------------------------------------

// element (k, mn) at base[k*ld + mn]; mn contiguous; scalar loads.
template <int BMN>
struct MContigLoader {
    struct Params {
        const float* base;
        int K, MN, ld;
        long long batch_stride;  // elements, indexed by blockIdx.y
    };
    static constexpr int LD = BMN;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr int EPT = BMN * BK / NT;
    static constexpr int STEP = NT / BMN;
    static constexpr bool DMA = GZ_IGEMM_DMA;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t col_off;
    int kb, mn_l, K, ld;
    bool col_ok;
    float r[DMA ? 1 : EPT];
    __device__ __forceinline__ void issue_lds(int kc, float* dst) {
        float* wbase = dst + (mn_l - (int)(threadIdx.x & 63));
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            int kl = kb + STEP * j;
            int k = kc * BK + kl;
            uint32_t v = (col_ok && k < K) ? (uint32_t)k * (uint32_t)ld * 4u + col_off : OOB;
            bload_lds4(rsrc, wbase + kl * LD, v, 0);
        }
    }
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        rsrc = make_rsrc(p.base + (long long)y * p.batch_stride, (uint32_t)p.K * p.ld * 4u);
        mn_l = tid % BMN;
        kb = tid / BMN;
        int mn = tile * BMN + mn_l;
        col_ok = mn < p.MN;
        col_off = (uint32_t)mn * 4u;
        K = p.K;
        ld = p.ld;
    }
    __device__ __forceinline__ void issue(int kc) {
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                int k = kc * BK + kb + STEP * j;
                uint32_t v = (col_ok && k < K) ? (uint32_t)k * (uint32_t)ld * 4u + col_off : OOB;
                r[j] = bload(rsrc, v, 0);
            }
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) dst[(kb + STEP * j) * LD + mn_l] = r[j];
        }
    }
};

// same layout, 16-byte loads; requires ld % 4 == 0, MN % 4 == 0 and a 16-byte aligned base.
template <int BMN>
struct MContigLoader4 {
    using Params = typename MContigLoader<BMN>::Params;
    static constexpr int LD = BMN;
    static constexpr int C4 = BMN / 4;                 // float4 columns
    static constexpr int ROWS = NT / C4;               // rows covered per pass
    static constexpr int PASSES = (BK + ROWS - 1) / ROWS;
    static constexpr bool DMA = GZ_IGEMM_DMA;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t col_off;
    int kb, c4, K, ld;
    bool col_ok;
    f32x4 r[DMA ? 1 : PASSES];
    static constexpr int NPARTS = PASSES;
    __device__ __forceinline__ void issue_lds_part(int kc, float* dst, int j) {
        const int lane = threadIdx.x & 63;
        const int kb0 = kb - lane / C4;
        int kl = kb + ROWS * j;
        if (kb0 + ROWS * j < BK) {
            int k = kc * BK + kl;
            uint32_t v = (col_ok && k < K) ? (uint32_t)k * (uint32_t)ld * 4u + col_off : OOB;
            bload_lds16(rsrc, dst + (kb0 + ROWS * j) * LD, v, 0);
        }
    }
    // a wave covers 64/C4 whole rows of the [k][BMN] image = one contiguous 1 KiB piece (LD == BMN)
    __device__ __forceinline__ void issue_lds(int kc, float* dst) {
        const int lane = threadIdx.x & 63;
        const int kb0 = kb - lane / C4;           // first row of this wave (wave-uniform)
#pragma unroll
        for (int j = 0; j < PASSES; ++j) {
            int kl = kb + ROWS * j;
            if (kb0 + ROWS * j < BK) {            // wave-uniform: 64/C4 divides BK
                int k = kc * BK + kl;
                uint32_t v = (col_ok && k < K) ? (uint32_t)k * (uint32_t)ld * 4u + col_off : OOB;
                bload_lds16(rsrc, dst + (kb0 + ROWS * j) * LD, v, 0);
            }
        }
    }
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        rsrc = make_rsrc(p.base + (long long)y * p.batch_stride, (uint32_t)p.K * p.ld * 4u);
        c4 = tid % C4;
        kb = tid / C4;
        int mn = tile * BMN + c4 * 4;
        col_ok = mn < p.MN;
        col_off = (uint32_t)mn * 4u;
        K = p.K;
        ld = p.ld;
    }
    __device__ __forceinline__ void issue(int kc) {
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < PASSES; ++j) {
                int kl = kb + ROWS * j;
                int k = kc * BK + kl;
                bool ok = col_ok && k < K && kl < BK;
                uint32_t v = ok ? (uint32_t)k * (uint32_t)ld * 4u + col_off : OOB;
                r[j] = bload4(rsrc, v, 0);
            }
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < PASSES; ++j) {
                int kl = kb + ROWS * j;
                if (kl < BK) *reinterpret_cast<f32x4*>(dst + kl * LD + c4 * 4) = r[j];
            }
        }
    }
};

// element (mn, k) at base[mn*ld + k]; k contiguous; transposed on the way into LDS.
// LD = BMN + 2 makes the transposing ds_write_b32 conflict-free: a half-wave
// holds 16 k x 2 mn and lands on banks (2k + mn) mod 32.
template <int BMN>
struct KContigLoader {
    struct Params {
        const float* base;
        int K, MN, ld;
        long long batch_stride;
    };
    static constexpr int LD = BMN + 2;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr bool DMA = false;
    __device__ __forceinline__ void issue_lds(int, float*) {}
    static constexpr int EPT = BMN / 16;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[EPT];
    int kl, mn_l, K;
    float r[EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        rsrc = make_rsrc(p.base + (long long)y * p.batch_stride, (uint32_t)p.MN * p.ld * 4u);
        kl = tid & 15;
        mn_l = tid >> 4;
        K = p.K;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            int mn = tile * BMN + mn_l + 16 * j;
            voff[j] = mn < p.MN ? ((uint32_t)mn * (uint32_t)p.ld + kl) * 4u : OOB;
        }
    }
    __device__ __forceinline__ void issue(int kc) {
        bool kok = kc * BK + kl < K;
#pragma unroll
        for (int j = 0; j < EPT; ++j) r[j] = bload(rsrc, kok ? voff[j] : OOB, (uint32_t)kc * BK * 4u);
    }
    __device__ __forceinline__ void commit(float* dst) const {
#pragma unroll
        for (int j = 0; j < EPT; ++j) dst[kl * LD + mn_l + 16 * j] = r[j];
    }
};

// ---------------------------------------------------------------------------
// convolution geometry
// ---------------------------------------------------------------------------
// x: [N, C, H, W]   ("image side": conv input, dgrad output)
// y: [N, K, OH, OW] ("feature side": conv output, dgrad input)
// w: [K, C, KH, KW] (Conv2d weight; a ConvTranspose2d weight [Cin_T, Cout_T, KH, KW]
//                    is the same array with K = Cin_T, C = Cout_T)
struct ConvShape {
    int N, C, H, W, K, OH, OW;
};

// A operand of the forward GEMM: A[m = (n, oy, ox)][k = (c, ky, kx)] = x[n][c][oy*S-P+ky][ox*S-P+kx]
template <int BM, int KH, int KW, int S, int P>
struct ConvFwdALoader {
    struct Params {
        const float* x;
        ConvShape s;
        FastDiv div_ohw, div_ow;
    };
    static constexpr int LD = BM;
    static constexpr int EPT = BM * BK / NT;
    static constexpr int STEP = NT / BM;
    static constexpr bool FIXED = (KH * KW == BK);  // a chunk is exactly one input channel
    static constexpr bool DMA = GZ_IGEMM_DMA;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[FIXED ? EPT : 1];
    uint32_t nbase;
    int kb, m_l, iy0, ix0, C, H, W;
    bool m_ok;
    float r[DMA ? 1 : EPT];
    __device__ __forceinline__ uint32_t tap_voff(int kc, int j) const {
        if constexpr (FIXED) {
            return voff[j];
        } else {
            int k = kc * BK + kb + STEP * j;
            int c = k / (KH * KW);
            int tap = k - c * (KH * KW);
            int iy = iy0 + tap / KW, ix = ix0 + tap % KW;
            bool ok = m_ok && c < C && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            return ok ? (nbase + (uint32_t)((c * H + iy) * W + ix)) * 4u : OOB;
        }
    }
    static constexpr int NPARTS = EPT;
    __device__ __forceinline__ void issue_lds_part(int kc, float* dst, int j) {
        float* wbase = dst + (m_l - (int)(threadIdx.x & 63));
        const uint32_t soff = FIXED ? (uint32_t)kc * (uint32_t)(H * W) * 4u : 0u;
        bload_lds4(rsrc, wbase + (kb + STEP * j) * LD, tap_voff(kc, j), soff);
    }
    // a wave's 64 lanes are 64 consecutive m of one k row: one contiguous 256-byte LDS piece
    __device__ __forceinline__ void issue_lds(int kc, float* dst) {
        float* wbase = dst + (m_l - (int)(threadIdx.x & 63));
        const uint32_t soff = FIXED ? (uint32_t)kc * (uint32_t)(H * W) * 4u : 0u;
#pragma unroll
        for (int j = 0; j < EPT; ++j) bload_lds4(rsrc, wbase + (kb + STEP * j) * LD, tap_voff(kc, j), soff);
    }
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.x, (uint32_t)s.N * s.C * s.H * s.W * 4u);
        m_l = tid % BM;
        kb = tid / BM;
        uint32_t m = (uint32_t)tile * BM + m_l;
        m_ok = m < (uint32_t)s.N * s.OH * s.OW;
        uint32_t n = fdiv(m, p.div_ohw);
        uint32_t pix = m - n * (uint32_t)(s.OH * s.OW);
        uint32_t oy = fdiv(pix, p.div_ow);
        uint32_t ox = pix - oy * (uint32_t)s.OW;
        iy0 = (int)oy * S - P;
        ix0 = (int)ox * S - P;
        nbase = n * (uint32_t)(s.C * s.H * s.W);
        C = s.C; H = s.H; W = s.W;
        if constexpr (FIXED) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                int tap = kb + STEP * j;
                int iy = iy0 + tap / KW, ix = ix0 + tap % KW;
                bool ok = m_ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                voff[j] = ok ? (nbase + (uint32_t)(iy * W + ix)) * 4u : OOB;
            }
        }
    }
    __device__ __forceinline__ void issue(int kc) {
        if constexpr (!DMA) {
            const uint32_t soff = FIXED ? (uint32_t)kc * (uint32_t)(H * W) * 4u : 0u;
#pragma unroll
            for (int j = 0; j < EPT; ++j) r[j] = bload(rsrc, tap_voff(kc, j), soff);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) dst[(kb + STEP * j) * LD + m_l] = r[j];
        }
    }
};

// Tap-major reduction order for kernels whose tap count does not divide a chunk (3x3: 9, 5x5: 25): with the
// usual k = (c, tap) order every element of every chunk needs its own (c, ky, kx) decode and bounds test -- ~25
// VALU instructions per 4-byte load, which made the 16-channel 128x128 ResNet layers VALU-bound on address
// arithmetic.  With k = (tap, c), c padded to a multiple of BK, a chunk is ONE tap and BK consecutive channels:
// per lane the voffsets are loop-invariant (pixel + its BK/STEP channel rows), the tap and the channel block
// advance through the wave-uniform scalar offset, and the padding test is one compare pair per chunk.
// The descriptor base is moved back by the largest negative tap shift so that voffsets stay non-negative; taps
// in the padding use the out-of-range voffset and are never dereferenced.
constexpr int round_bk(int v) { return (v + BK - 1) / BK * BK; }

// A[m = (n, oy, ox)][k = (tap, c)] = x[n][c][oy*S-P+ky][ox*S-P+kx]
template <int BM, int KH, int KW, int S, int P>
struct ConvFwdALoaderTap {
    using Params = typename ConvFwdALoader<BM, KH, KW, S, P>::Params;
    static constexpr int LD = BM;
    static constexpr int EPT = BM * BK / NT;
    static constexpr int STEP = NT / BM;
    static constexpr bool DMA = GZ_IGEMM_DMA;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[EPT];
    int kb, m_l, iy0, ix0, C, H, W, cblocks;
    bool m_ok;
    float r[DMA ? 1 : EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        const uint32_t shift = (uint32_t)(P * s.W + P) * 4u;
        rsrc = make_rsrc(reinterpret_cast<const char*>(p.x) - shift, (uint32_t)s.N * s.C * s.H * s.W * 4u + shift);
        m_l = tid % BM;
        kb = tid / BM;
        uint32_t m = (uint32_t)tile * BM + m_l;
        m_ok = m < (uint32_t)s.N * s.OH * s.OW;
        uint32_t n = fdiv(m, p.div_ohw);
        uint32_t pix = m - n * (uint32_t)(s.OH * s.OW);
        uint32_t oy = fdiv(pix, p.div_ow);
        uint32_t ox = pix - oy * (uint32_t)s.OW;
        iy0 = (int)oy * S - P;
        ix0 = (int)ox * S - P;
        C = s.C; H = s.H; W = s.W;
        cblocks = round_bk(s.C) / BK;
        const int pos = (int)(n * (uint32_t)(s.C * s.H * s.W)) + (iy0 + P) * W + (ix0 + P);   // >= 0 (shifted base)
#pragma unroll
        for (int j = 0; j < EPT; ++j) voff[j] = (uint32_t)(pos + (kb + STEP * j) * H * W) * 4u;
    }
    // chunk -> (tap, channel block): wave-uniform
    __device__ __forceinline__ void chunk(int kc, uint32_t& soff, bool& ok, int& cb) const {
        const int tap = kc / cblocks;
        cb = (kc - tap * cblocks) * BK;
        const int dy = tap / KW, dx = tap - dy * KW;
        soff = (uint32_t)(cb * H * W + dy * W + dx) * 4u;
        ok = m_ok && (unsigned)(iy0 + dy) < (unsigned)H && (unsigned)(ix0 + dx) < (unsigned)W;
    }
    __device__ __forceinline__ void issue_lds(int kc, float* dst) {
        float* wbase = dst + (m_l - (int)(threadIdx.x & 63));
        uint32_t soff; bool ok; int cb;
        chunk(kc, soff, ok, cb);
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            bload_lds4(rsrc, wbase + (kb + STEP * j) * LD, (ok && cb + kb + STEP * j < C) ? voff[j] : OOB, soff);
    }
    __device__ __forceinline__ void issue(int kc) {
        if constexpr (!DMA) {
            uint32_t soff; bool ok; int cb;
            chunk(kc, soff, ok, cb);
#pragma unroll
            for (int j = 0; j < EPT; ++j) r[j] = bload(rsrc, (ok && cb + kb + STEP * j < C) ? voff[j] : OOB, soff);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) dst[(kb + STEP * j) * LD + m_l] = r[j];
        }
    }
};

// The same tap-major forward loader with the geometry as run-time values (rectangular kernels, per-axis stride and
// padding): the evaluation path's InceptionV3 has 3x3 s2 p0, 5x5 s1 p2, 1x7 / 7x1, 1x3 / 3x1 ... layers, none of
// which is worth a template instantiation of its own (forward only, no training step runs through them).
template <int BM>
struct ConvFwdALoaderTapAny {
    struct Params {
        const float* x;
        ConvShape s;
        FastDiv div_ohw, div_ow;
        int KH, KW, SH, SW, PH, PW;
    };
    static constexpr int LD = BM;
    static constexpr int EPT = BM * BK / NT;
    static constexpr int STEP = NT / BM;
    static constexpr bool DMA = GZ_IGEMM_DMA;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[EPT];
    int kb, m_l, iy0, ix0, C, H, W, cblocks, KW;
    bool m_ok;
    float r[DMA ? 1 : EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        const uint32_t shift = (uint32_t)(p.PH * s.W + p.PW) * 4u;
        rsrc = make_rsrc(reinterpret_cast<const char*>(p.x) - shift, (uint32_t)s.N * s.C * s.H * s.W * 4u + shift);
        m_l = tid % BM;
        kb = tid / BM;
        uint32_t m = (uint32_t)tile * BM + m_l;
        m_ok = m < (uint32_t)s.N * s.OH * s.OW;
        uint32_t n = fdiv(m, p.div_ohw);
        uint32_t pix = m - n * (uint32_t)(s.OH * s.OW);
        uint32_t oy = fdiv(pix, p.div_ow);
        uint32_t ox = pix - oy * (uint32_t)s.OW;
        iy0 = (int)oy * p.SH - p.PH;
        ix0 = (int)ox * p.SW - p.PW;
        C = s.C; H = s.H; W = s.W; KW = p.KW;
        cblocks = round_bk(s.C) / BK;
        const int pos = (int)(n * (uint32_t)(s.C * s.H * s.W)) + (iy0 + p.PH) * W + (ix0 + p.PW);   // >= 0 (shifted base)
#pragma unroll
        for (int j = 0; j < EPT; ++j) voff[j] = (uint32_t)(pos + (kb + STEP * j) * H * W) * 4u;
    }
    __device__ __forceinline__ void chunk(int kc, uint32_t& soff, bool& ok, int& cb) const {
        const int tap = kc / cblocks;
        cb = (kc - tap * cblocks) * BK;
        const int dy = tap / KW, dx = tap - dy * KW;
        soff = (uint32_t)(cb * H * W + dy * W + dx) * 4u;
        ok = m_ok && (unsigned)(iy0 + dy) < (unsigned)H && (unsigned)(ix0 + dx) < (unsigned)W;
    }
    __device__ __forceinline__ void issue_lds(int kc, float* dst) {
        float* wbase = dst + (m_l - (int)(threadIdx.x & 63));
        uint32_t soff; bool ok; int cb;
        chunk(kc, soff, ok, cb);
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            bload_lds4(rsrc, wbase + (kb + STEP * j) * LD, (ok && cb + kb + STEP * j < C) ? voff[j] : OOB, soff);
    }
    __device__ __forceinline__ void issue(int kc) {
        if constexpr (!DMA) {
            uint32_t soff; bool ok; int cb;
            chunk(kc, soff, ok, cb);
#pragma unroll
            for (int j = 0; j < EPT; ++j) r[j] = bload(rsrc, (ok && cb + kb + STEP * j < C) ? voff[j] : OOB, soff);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) dst[(kb + STEP * j) * LD + m_l] = r[j];
        }
    }
};

// transposed conv, phase (py, px), tap-major: A[m = (n, a, b)][k = (tap, ko)] = y[n][ko][oy0 - ty][ox0 - tx] with
// tap = ty * nx + tx over the phase's own ny x nx taps (dg_taps); the weight rows of a phase are packed in the
// same order (pack_dgrad_tap_kernel), so phases with fewer taps simply have fewer chunks.
template <int BM, int KH, int KW, int S, int P>
struct ConvDgALoaderTap {
    static constexpr int TY = (KH + S - 1) / S, TX = (KW + S - 1) / S;
    static constexpr int TAPS = TY * TX;
    static constexpr bool UNIFORM = false;      // phases have their own chunk counts (run_dgrad passes them)
    struct Params {
        const float* y;
        ConvShape s;
        int AH, AW;
        FastDiv div_ahw, div_aw;
    };
    static constexpr int LD = BM;
    static constexpr int EPT = BM * BK / NT;
    static constexpr int STEP = NT / BM;
    static constexpr bool DMA = GZ_IGEMM_DMA;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[EPT];
    int kb, m_l, oy0, ox0, K, OH, OW, kblocks, nx_p;
    bool m_ok;
    float r[DMA ? 1 : EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int phase, int tid) {
        const ConvShape& s = p.s;
        const uint32_t shift = (uint32_t)((TY - 1) * s.OW + (TX - 1)) * 4u;
        rsrc = make_rsrc(reinterpret_cast<const char*>(p.y) - shift, (uint32_t)s.N * s.K * s.OH * s.OW * 4u + shift);
        m_l = tid % BM;
        kb = tid / BM;
        const int py = phase / S, px = phase % S;
        nx_p = dg_taps(KW, S, P, px);
        uint32_t m = (uint32_t)tile * BM + m_l;
        m_ok = m < (uint32_t)s.N * p.AH * p.AW;
        uint32_t n = fdiv(m, p.div_ahw);
        uint32_t pix = m - n * (uint32_t)(p.AH * p.AW);
        uint32_t a = fdiv(pix, p.div_aw);
        uint32_t b = pix - a * (uint32_t)p.AW;
        oy0 = (int)a + (py + P) / S;
        ox0 = (int)b + (px + P) / S;
        K = s.K; OH = s.OH; OW = s.OW;
        kblocks = round_bk(s.K) / BK;
        // voffsets address (oy0 - (TY-1), ox0 - (TX-1)) through the shifted base; the chunk's scalar offset
        // walks forward from there to (oy0 - ty, ox0 - tx)
        const int pos = (int)(n * (uint32_t)(s.K * s.OH * s.OW)) + oy0 * OW + ox0;
#pragma unroll
        for (int j = 0; j < EPT; ++j) voff[j] = (uint32_t)(pos + (kb + STEP * j) * OH * OW) * 4u;
    }
    __device__ __forceinline__ void chunk(int kc, uint32_t& soff, bool& ok, int& kob) const {
        const int tap = kc / kblocks;
        kob = (kc - tap * kblocks) * BK;
        const int ty = tap / nx_p, tx = tap - ty * nx_p;
        soff = (uint32_t)(kob * OH * OW + (TY - 1 - ty) * OW + (TX - 1 - tx)) * 4u;
        ok = m_ok && (unsigned)(oy0 - ty) < (unsigned)OH && (unsigned)(ox0 - tx) < (unsigned)OW;
    }
    __device__ __forceinline__ void issue_lds(int kc, float* dst) {
        float* wbase = dst + (m_l - (int)(threadIdx.x & 63));
        uint32_t soff; bool ok; int kob;
        chunk(kc, soff, ok, kob);
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            bload_lds4(rsrc, wbase + (kb + STEP * j) * LD, (ok && kob + kb + STEP * j < K) ? voff[j] : OOB, soff);
    }
    __device__ __forceinline__ void issue(int kc) {
        if constexpr (!DMA) {
            uint32_t soff; bool ok; int kob;
            chunk(kc, soff, ok, kob);
#pragma unroll
            for (int j = 0; j < EPT; ++j) r[j] = bload(rsrc, (ok && kob + kb + STEP * j < K) ? voff[j] : OOB, soff);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) dst[(kb + STEP * j) * LD + m_l] = r[j];
        }
    }
};

// Forward loader for KW == 4, KH*KW == 16 (the k4 s2 p1 layers): the four kx taps of one (c, ky) are four
// consecutive floats of an input row, so each lane gathers them with ONE 16-byte load (dword aligned)
// instead of four 4-byte loads -- a chunk (one input channel) needs 2 vector loads per lane instead of 8.
// Columns that fall into the horizontal padding are zeroed in registers (per-lane 4-bit mask); rows in
// the vertical padding are dropped by the descriptor's range check (which is per dword, so a vector that
// straddles the end of the tensor is safe).  Lanes whose first tap lies left of the image (ox = 0, P = 1)
// load from column 0 and rotate the vector by one in registers: nothing is ever read in front of the
// tensor (an address before a page-aligned allocation faults even if the value is discarded).
template <int BM, int S, int P>
struct ConvFwdALoaderK4V {
    using Params = typename ConvFwdALoader<BM, 4, 4, S, P>::Params;
    static constexpr int LD = BM;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr bool DMA = false;
    static constexpr int VPT = BM * 4 / NT;        // vector loads per lane per chunk (2 for BM=128, 1 for 64)
    static constexpr int KYSTEP = NT / BM;         // 2 or 4
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[VPT];
    static_assert(P <= 1, "left padding wider than one column is not handled");
    int kyb, m_l, HW;
    uint32_t colmask;                              // bit kx set -> column is inside the image
    bool lshift;                                   // first tap is the left padding column
    f32x4 r[VPT];
    __device__ __forceinline__ void issue_lds(int, float*) {}
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.x, (uint32_t)(s.N * s.C * s.H * s.W) * 4u);
        m_l = tid % BM;
        kyb = tid / BM;
        uint32_t m = (uint32_t)tile * BM + m_l;
        const bool m_ok = m < (uint32_t)s.N * s.OH * s.OW;
        uint32_t n = fdiv(m, p.div_ohw);
        uint32_t pix = m - n * (uint32_t)(s.OH * s.OW);
        uint32_t oy = fdiv(pix, p.div_ow);
        uint32_t ox = pix - oy * (uint32_t)s.OW;
        const int iy0 = (int)oy * S - P, ix0 = (int)ox * S - P;
        HW = s.H * s.W;
        colmask = 0;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
            if ((unsigned)(ix0 + kx) < (unsigned)s.W) colmask |= 1u << kx;
        lshift = ix0 < 0;
        const int ixs = lshift ? 0 : ix0;
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            int iy = iy0 + kyb + KYSTEP * j;
            bool ok = m_ok && (unsigned)iy < (unsigned)s.H;
            voff[j] = ok ? (n * (uint32_t)(s.C * HW) + (uint32_t)(iy * s.W + ixs)) * 4u : OOB;
        }
    }
    __device__ __forceinline__ void issue(int kc) {
        const uint32_t soff = (uint32_t)kc * (uint32_t)HW * 4u;
#pragma unroll
        for (int j = 0; j < VPT; ++j) r[j] = bload4(rsrc, voff[j], soff);
    }
    __device__ __forceinline__ void commit(float* dst) const {
#pragma unroll
        for (int j = 0; j < VPT; ++j) {
            const int krow = (kyb + KYSTEP * j) * 4;
            f32x4 v = r[j];
            if (lshift) v = f32x4{0.f, v.x, v.y, v.z};
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) dst[(krow + kx) * LD + m_l] = (colmask >> kx) & 1u ? v[kx] : 0.f;
        }
    }
};

// Row-shared A loader for the k4 s2 p1 FORWARD convolution (round 2; the transposed one is ConvDgALoaderRow4).  The
// im2col image of one input channel is 16 taps x BM pixels = 16*BM floats, of which each input value appears four
// times; K4V gathers it with two 16-byte loads per lane and commits it with EIGHT ds_write_b32.  Here a chunk's LDS
// image is the raw input rows the tile touches -- per output-row segment of OW pixels its four input rows of 2*OW
// columns: 8*BM floats, one 16-byte LDS-DMA per lane, no VGPR staging, no ds_write -- and the taps are applied when
// the MFMA fragment is read: lane (segment, ox), k = (ky, kx) reads  image[(seg*4 + ky) * 2*OW + 2*ox + kx - 1]
// (stride-2 across lanes: a 2-way bank conflict on 2 of the 4 fragment reads per 4 MFMAs -- free).  The column left
// of the image (ox = 0, kx = 0) and right of it (ox = OW-1, kx = 3) are zeroed in the register; rows above / below
// are dropped by the descriptor's range check.  Needs W = 2*OW, H = 2*OH, OW <= BM, BM % OW == 0, 16-byte alignment.
template <int BM>
struct ConvFwdALoaderRow4 {
    using Params = typename ConvFwdALoader<BM, 4, 4, 2, 1>::Params;
    static constexpr int LD = BM;
    static constexpr bool DMA = true;
    static constexpr bool FWDROWS = true;
    static constexpr int NPARTS = 0;
    static constexpr int LANES = 2 * BM;                   // 8*BM floats / 4 per lane
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;
    int tid_, HW, OW, twoOW;
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.x, (uint32_t)(s.N * s.C * s.H * s.W) * 4u);
        tid_ = tid;
        HW = s.H * s.W; OW = s.OW; twoOW = 2 * s.OW;
        const int f = tid * 4;
        const int seg = f / (8 * OW), rem = f - seg * 8 * OW;
        const int ky = rem / twoOW, col = rem - ky * twoOW;
        const uint32_t m = (uint32_t)tile * BM + seg * OW;           // first pixel of the segment
        const bool m_ok = tid < LANES && m < (uint32_t)s.N * s.OH * s.OW;
        const uint32_t n = fdiv(m, p.div_ohw);
        const uint32_t oy = fdiv(m - n * (uint32_t)(s.OH * s.OW), p.div_ow);
        const int iy = (int)oy * 2 - 1 + ky;
        const bool ok = m_ok && (unsigned)iy < (unsigned)s.H;
        voff = ok ? (n * (uint32_t)(s.C * HW) + (uint32_t)(iy * s.W + col)) * 4u : OOB;
    }
    // fragment addressing of the lane that owns tile pixel m_local, half-wave `half` (k = 2s + half):
    //   address(s) = base + (s >> 1) * 2*OW + 2 * (s & 1);  zero the value on even s if z_even, on odd s if z_odd
    __device__ __forceinline__ void frag(int m_local, int half, int& base, bool& z_even, bool& z_odd) const {
        const int seg = m_local / OW, ox = m_local - seg * OW;
        base = seg * 8 * OW + 2 * ox - 1 + half;
        z_even = half == 0 && ox == 0;              // kx = 0
        z_odd = half == 1 && ox == OW - 1;          // kx = 3
    }
    __device__ __forceinline__ void issue_lds(int kc, float* dst) {
        if (tid_ < LANES)          // wave-uniform
            bload_lds16(rsrc, dst + (tid_ & ~63) * 4, voff, (uint32_t)kc * (uint32_t)HW * 4u);
    }
    __device__ __forceinline__ void issue(int) {}
    __device__ __forceinline__ void commit(float*) const {}
};

// Transposed convolution / data gradient, decomposed into S*S output phases.
// Phase (py, px) produces x[n][c][S*a+py][S*b+px]; along each axis it uses the taps
//   t = 0..T-1 :  ky = ((py + P) % S) + S*t ,  oy = a + (py + P)/S - t
// (taps with ky >= KH are zero in the packed weights).  Per phase:
//   A[m = (n, a, b)][k = (ko, ty, tx)] = y[n][ko][a + dy(ty)][b + dx(tx)]
template <int BM, int KH, int KW, int S, int P>
struct ConvDgALoader {
    static constexpr int TY = (KH + S - 1) / S, TX = (KW + S - 1) / S;
    static constexpr int TAPS = TY * TX;
    struct Params {
        const float* y;
        ConvShape s;
        int AH, AW;  // phase grid: H/S, W/S
        FastDiv div_ahw, div_aw;
    };
    static constexpr int LD = BM;
    static constexpr int EPT = BM * BK / NT;
    static constexpr int STEP = NT / BM;
    // every phase has TY x TX real taps iff the stride divides the kernel size; otherwise (k5 s2: 3 or 2 per
    // axis) the phase's own counts ty_p x tx_p index its tightly packed weight rows
    static constexpr bool UNIFORM = (KH % S == 0) && (KW % S == 0);
    static constexpr bool FIXED = UNIFORM && (BK % TAPS == 0);  // a chunk is BK/TAPS whole feature channels
    static constexpr bool DMA = GZ_IGEMM_DMA;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[FIXED ? EPT : 1];
    uint32_t nbase;
    int kb, m_l, oy0, ox0, K, OH, OW, ty_p, tx_p;
    bool m_ok;
    float r[DMA ? 1 : EPT];
    template <int TYP, int TXP>
    __device__ __forceinline__ uint32_t tap_voff_phase(int kc, int j) const {
        constexpr int TP = TYP * TXP;
        if constexpr (TP == 0) {
            return OOB;
        } else {
            int k = kc * BK + kb + STEP * j;
            int ko = k / TP;
            int tap = k - ko * TP;
            int oy = oy0 - tap / TXP, ox = ox0 - tap % TXP;
            bool ok = m_ok && ko < K && (unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW;
            return ok ? (nbase + (uint32_t)((ko * OH + oy) * OW + ox)) * 4u : OOB;
        }
    }
    __device__ __forceinline__ uint32_t tap_voff(int kc, int j) const {
        if constexpr (!UNIFORM) {      // wave-uniform branches: the phase is a workgroup property
            if (ty_p == TY) return tx_p == TX ? tap_voff_phase<TY, TX>(kc, j) : tap_voff_phase<TY, TX - 1>(kc, j);
            return tx_p == TX ? tap_voff_phase<TY - 1, TX>(kc, j) : tap_voff_phase<TY - 1, TX - 1>(kc, j);
        } else if constexpr (FIXED) {
            // feature channels past K only occur in a partial last chunk; their weights are zero-padded
            // but the reads must stay inside the tensor: the soffset is not range checked.
            int kol = (kb + STEP * j) / TAPS;
            return kc * (BK / TAPS) + kol < K ? voff[j] : OOB;
        } else {
            int k = kc * BK + kb + STEP * j;
            int ko = k / TAPS;
            int tap = k - ko * TAPS;
            int oy = oy0 - tap / TX, ox = ox0 - tap % TX;
            bool ok = m_ok && ko < K && (unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW;
            return ok ? (nbase + (uint32_t)((ko * OH + oy) * OW + ox)) * 4u : OOB;
        }
    }
    static constexpr int NPARTS = EPT;
    __device__ __forceinline__ void issue_lds_part(int kc, float* dst, int j) {
        float* wbase = dst + (m_l - (int)(threadIdx.x & 63));
        const uint32_t soff = FIXED ? (uint32_t)kc * (uint32_t)((BK / TAPS) * OH * OW) * 4u : 0u;
        bload_lds4(rsrc, wbase + (kb + STEP * j) * LD, tap_voff(kc, j), soff);
    }
    __device__ __forceinline__ void issue_lds(int kc, float* dst) {
        float* wbase = dst + (m_l - (int)(threadIdx.x & 63));
        const uint32_t soff = FIXED ? (uint32_t)kc * (uint32_t)((BK / TAPS) * OH * OW) * 4u : 0u;
#pragma unroll
        for (int j = 0; j < EPT; ++j) bload_lds4(rsrc, wbase + (kb + STEP * j) * LD, tap_voff(kc, j), soff);
    }
    __device__ __forceinline__ void init(const Params& p, int tile, int phase, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.y, (uint32_t)s.N * s.K * s.OH * s.OW * 4u);
        m_l = tid % BM;
        kb = tid / BM;
        int py = phase / S, px = phase % S;
        ty_p = dg_taps(KH, S, P, py);
        tx_p = dg_taps(KW, S, P, px);
        uint32_t m = (uint32_t)tile * BM + m_l;
        m_ok = m < (uint32_t)s.N * p.AH * p.AW;
        uint32_t n = fdiv(m, p.div_ahw);
        uint32_t pix = m - n * (uint32_t)(p.AH * p.AW);
        uint32_t a = fdiv(pix, p.div_aw);
        uint32_t b = pix - a * (uint32_t)p.AW;
        oy0 = (int)a + (py + P) / S;
        ox0 = (int)b + (px + P) / S;
        nbase = n * (uint32_t)(s.K * s.OH * s.OW);
        K = s.K; OH = s.OH; OW = s.OW;
        if constexpr (FIXED) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                int kk = kb + STEP * j;
                int kol = kk / TAPS, tap = kk % TAPS;
                int oy = oy0 - tap / TX, ox = ox0 - tap % TX;
                bool ok = m_ok && (unsigned)oy < (unsigned)OH && (unsigned)ox < (unsigned)OW;
                voff[j] = ok ? (nbase + (uint32_t)((kol * OH + oy) * OW + ox)) * 4u : OOB;
            }
        }
    }
    __device__ __forceinline__ void issue(int kc) {
        if constexpr (!DMA) {
            const uint32_t soff = FIXED ? (uint32_t)kc * (uint32_t)((BK / TAPS) * OH * OW) * 4u : 0u;
#pragma unroll
            for (int j = 0; j < EPT; ++j) r[j] = bload(rsrc, tap_voff(kc, j), soff);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) dst[(kb + STEP * j) * LD + m_l] = r[j];
        }
    }
};

// Row-shared A loader for the k4 s2 p1 transposed convolution (round 2).  In ConvDgALoader the two horizontal taps of
// a (feature channel, vertical tap) pair are two LDS rows holding the same feature row, one of them shifted by a
// column, and every element arrives by its own 4-byte LDS-DMA (8 per lane and chunk: executing them cost ~8 % of
// the kernel).  Here a chunk's A image is 8 rows = 4 feature channels x 2 vertical taps of UNSHIFTED feature rows,
// landed by ONE 16-byte LDS-DMA per lane (1 KiB per wavefront, 4 aligned pixels per lane), and the horizontal tap is
// applied when the MFMA fragment is read: k-step s reads row s, its first half-wave (tx = 0) and second half-wave
// (tx = 1) at column offsets (px + P) / S - tx, i.e. {0, -1} or {+1, 0}.  The one column that a shifted read takes
// from the neighbouring image row (b = 0 for -1, b = AW - 1 for +1) is zeroed in the register (a lane's b is fixed).
// Half the LDS-DMA bytes, an eighth of the instructions.  Needs AW % 4 == 0 and a 16-byte aligned tensor.
template <int BM, int KH, int KW, int S, int P>
struct ConvDgALoaderRow4 {
    static_assert(KH == 4 && KW == 4 && S == 2 && P == 1, "k4 s2 p1 only");
    static constexpr int TAPS = 4;
    static constexpr bool UNIFORM = true, FIXED = true;
    using Params = typename ConvDgALoader<BM, KH, KW, S, P>::Params;
    static constexpr int LD = BM;
    static constexpr bool DMA = true;
    static constexpr bool ROWSHARE = true;
    static constexpr int NPARTS = 0;
    static constexpr int ROWS = BK / 2;                    // LDS rows per chunk
    static constexpr int LANES = ROWS * BM / 4;            // lanes that load (256 for BM = 128, 128 for BM = 64)
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff;
    int kol, K, OHW, tid_, shift_half1;
    __device__ __forceinline__ void init(const Params& p, int tile, int phase, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.y, (uint32_t)s.N * s.K * s.OH * s.OW * 4u);
        tid_ = tid;
        K = s.K; OHW = s.OH * s.OW;
        const int py = phase / S, px = phase % S;
        shift_half1 = (px + P) / S - 1;                    // tx = 1; tx = 0 reads at (px + P) / S
        const int r = tid / (BM / 4), c4 = tid % (BM / 4);
        kol = r >> 1;
        const int ty = r & 1;
        uint32_t m = (uint32_t)tile * BM + c4 * 4;
        const bool m_ok = tid < LANES && m < (uint32_t)s.N * p.AH * p.AW;
        uint32_t n = fdiv(m, p.div_ahw);
        uint32_t pix = m - n * (uint32_t)(p.AH * p.AW);
        uint32_t a = fdiv(pix, p.div_aw);
        uint32_t b = pix - a * (uint32_t)p.AW;
        const int oy = (int)a + (py + P) / S - ty;
        const bool ok = m_ok && (unsigned)oy < (unsigned)s.OH;
        voff = ok ? (n * (uint32_t)(s.K * OHW) + (uint32_t)((kol * s.OH + oy) * s.OW) + b) * 4u : OOB;
    }
    // column offset the fragment reads of half-wave `half` apply, and whether lane-column b must be zeroed
    __device__ __forceinline__ int frag_shift(int half) const { return half ? shift_half1 : shift_half1 + 1; }
    __device__ __forceinline__ void issue_lds(int kc, float* dst) {
        if (tid_ < LANES) {        // wave-uniform: LANES is a multiple of 64
            const uint32_t soff = (uint32_t)kc * (uint32_t)((BK / TAPS) * OHW) * 4u;
            const uint32_t v = kc * (BK / TAPS) + kol < K ? voff : OOB;
            bload_lds16(rsrc, dst + (tid_ & ~63) * 4, v, soff);
        }
    }
    __device__ __forceinline__ void issue(int) {}
    __device__ __forceinline__ void commit(float*) const {}
};

// Weight gradient: dW[ko][(c, ky, kx)] = sum_{p = (n, oy, ox)} y[n][ko][oy][ox] * x[n][c][oy*S-P+ky][ox*S-P+kx]
// A[m = ko][k = p] : k-contiguous inside one image.
template <int BM>
struct WgALoader {
    struct Params {
        const float* y;
        ConvShape s;
        FastDiv div_ohw;
        int KTOT;  // N*OH*OW
    };
    static constexpr int LD = BM + 2;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr bool DMA = false;
    __device__ __forceinline__ void issue_lds(int, float*) {}
    static constexpr int EPT = BM / 16;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t koff[EPT];
    int kl, m_l, OHW, KOHW, KTOT;
    FastDiv div_ohw;
    float r[EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.y, (uint32_t)s.N * s.K * s.OH * s.OW * 4u);
        kl = tid & 15;
        m_l = tid >> 4;
        OHW = s.OH * s.OW;
        KOHW = s.K * OHW;
        KTOT = p.KTOT;
        div_ohw = p.div_ohw;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            int ko = tile * BM + m_l + 16 * j;
            koff[j] = ko < s.K ? (uint32_t)ko * (uint32_t)OHW : OOB;
        }
    }
    __device__ __forceinline__ void issue(int kc) {
        uint32_t p = (uint32_t)kc * BK + kl;
        uint32_t n = fdiv(p, div_ohw);
        uint32_t pix = p - n * (uint32_t)OHW;
        uint32_t base = n * (uint32_t)KOHW + pix;
        bool ok = p < (uint32_t)KTOT;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            uint32_t v = (ok && koff[j] != OOB) ? (base + koff[j]) * 4u : OOB;
            r[j] = bload(rsrc, v, 0);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
#pragma unroll
        for (int j = 0; j < EPT; ++j) dst[kl * LD + m_l + 16 * j] = r[j];
    }
};

// B[k = p][n = (c, ky, kx)] = x[n_img][c][oy*S-P+ky][ox*S-P+kx]
template <int BN, int KH, int KW, int S, int P>
struct WgBLoader {
    struct Params {
        const float* x;
        ConvShape s;
        FastDiv div_ohw, div_ow;
        int KTOT, NTOT;  // N*OH*OW, C*KH*KW
    };
    static constexpr int LD = BN + 2;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr bool DMA = false;
    __device__ __forceinline__ void issue_lds(int, float*) {}
    static constexpr int EPT = BN / 16;
    __amdgpu_buffer_rsrc_t rsrc;
    int toff[EPT];   // (c*H + ky - P)*W + kx - P, or INT_MIN when the column is out of range
    int ky_[EPT], kx_[EPT];
    int kl, n_l, OHW, OW, H, W, CHW, KTOT;
    FastDiv div_ohw, div_ow;
    float r[EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.x, (uint32_t)s.N * s.C * s.H * s.W * 4u);
        kl = tid & 15;
        n_l = tid >> 4;
        OHW = s.OH * s.OW; OW = s.OW; H = s.H; W = s.W; CHW = s.C * s.H * s.W;
        KTOT = p.KTOT;
        div_ohw = p.div_ohw; div_ow = p.div_ow;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            int col = tile * BN + n_l + 16 * j;
            int c = col / (KH * KW);
            int tap = col - c * (KH * KW);
            ky_[j] = tap / KW - P;
            kx_[j] = tap % KW - P;
            toff[j] = col < p.NTOT ? (c * H + ky_[j]) * W + kx_[j] : INT32_MIN;
        }
    }
    __device__ __forceinline__ void issue(int kc) {
        uint32_t p = (uint32_t)kc * BK + kl;
        uint32_t n = fdiv(p, div_ohw);
        uint32_t pix = p - n * (uint32_t)OHW;
        uint32_t oy = fdiv(pix, div_ow);
        uint32_t ox = pix - oy * (uint32_t)OW;
        int by = (int)oy * S, bx = (int)ox * S;
        int base = (int)n * CHW + by * W + bx;
        bool ok = p < (uint32_t)KTOT;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            bool v = ok && toff[j] != INT32_MIN && (unsigned)(by + ky_[j]) < (unsigned)H &&
                     (unsigned)(bx + kx_[j]) < (unsigned)W;
            r[j] = bload(rsrc, v ? (uint32_t)(base + toff[j]) * 4u : OOB, 0);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
#pragma unroll
        for (int j = 0; j < EPT; ++j) dst[kl * LD + n_l + 16 * j] = r[j];
    }
};

// ---------------------------------------------------------------------------
// Row-aligned weight-gradient loaders.  When a K chunk (16 output pixels) is R = 16/CW whole
// row segments of CW = min(OW, 16) pixels inside one image, the pixel -> (n, oy, ox) decode is
// wave-uniform (scalar unit, once per chunk) and every lane's address is
//     scalar chunk base + loop-invariant per-lane offset,
// exactly like the forward loader; padding is a 4-bit (top, bottom, left, right) test against the
// chunk's position.  Preconditions (checked by the launcher): OW % CW == 0, OH % R == 0,
// S*R >= max(P, KH-1-P), S*CW >= max(P, KW-1-P).
// ---------------------------------------------------------------------------
struct WgRowGeom {
    int CW, R;               // chunk = R rows x CW columns of output pixels
    FastDiv div_ohw, div_ow;
};

template <int BM>
struct WgALoaderRow {
    struct Params {
        const float* y;
        ConvShape s;
        WgRowGeom g;
        int KTOT;
    };
    static constexpr int LD = BM + 2;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr bool DMA = false;
    __device__ __forceinline__ void issue_lds(int, float*) {}
    static constexpr int EPT = BM / 16;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[EPT];
    int kl, m_l, OHW, OW, KOHW;
    FastDiv div_ohw, div_ow;
    float r[EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.y, (uint32_t)s.N * s.K * s.OH * s.OW * 4u);
        kl = tid & 15;
        m_l = tid >> 4;
        OHW = s.OH * s.OW; OW = s.OW; KOHW = s.K * OHW;
        div_ohw = p.g.div_ohw; div_ow = p.g.div_ow;
        const int dy = kl / p.g.CW, dx = kl % p.g.CW;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            int ko = tile * BM + m_l + 16 * j;
            voff[j] = ko < s.K ? (uint32_t)(ko * OHW + dy * OW + dx) * 4u : OOB;
        }
    }
    __device__ __forceinline__ void issue(int kc) {
        // wave-uniform decode of the chunk's first pixel
        uint32_t p0 = (uint32_t)kc * BK;
        uint32_t n = fdiv(p0, div_ohw);
        uint32_t rem = p0 - n * (uint32_t)OHW;
        uint32_t soff = (n * (uint32_t)KOHW + rem) * 4u;
#pragma unroll
        for (int j = 0; j < EPT; ++j) r[j] = bload(rsrc, voff[j], soff);
    }
    __device__ __forceinline__ void commit(float* dst) const {
#pragma unroll
        for (int j = 0; j < EPT; ++j) dst[kl * LD + m_l + 16 * j] = r[j];
    }
};

template <int BN, int KH, int KW, int S, int P>
struct WgBLoaderRow {
    struct Params {
        const float* x;
        ConvShape s;
        WgRowGeom g;
        int KTOT, NTOT;
    };
    static constexpr int LD = BN + 2;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr bool DMA = false;
    __device__ __forceinline__ void issue_lds(int, float*) {}
    static constexpr int EPT = BN / 16;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[EPT];      // loop-invariant byte offset (may wrap for padded taps: those are masked)
    uint32_t flags[EPT];     // bit0 top, bit1 bottom, bit2 left, bit3 right padding; bit4 column out of range
    int kl, n_l, OHW, OW, OH, W, CHW, CW, R;
    FastDiv div_ohw, div_ow;
    float r[EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        // per-lane offsets of interior chunks can be negative (taps above / left of the chunk's first
        // input pixel); voffset is unsigned, so the descriptor base is moved back by the largest
        // negative offset and every voffset forward by the same amount (never dereferenced there:
        // such taps are either inside the tensor or masked)
        const int shift = P * s.W + P;
        rsrc = make_rsrc(p.x - shift, (uint32_t)(s.N * s.C * s.H * s.W + shift) * 4u);
        kl = tid & 15;
        n_l = tid >> 4;
        OHW = s.OH * s.OW; OW = s.OW; OH = s.OH; W = s.W; CHW = s.C * s.H * s.W;
        CW = p.g.CW; R = p.g.R;
        div_ohw = p.g.div_ohw; div_ow = p.g.div_ow;
        const int dy = kl / CW, dx = kl % CW;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            int col = tile * BN + n_l + 16 * j;
            int c = col / (KH * KW);
            int tap = col - c * (KH * KW);
            int ry = S * dy + tap / KW - P;          // input row relative to the chunk's first input row
            int rx = S * dx + tap % KW - P;
            voff[j] = (uint32_t)((c * s.H + ry) * W + rx + shift) * 4u;
            uint32_t f = 0;
            if (ry < 0) f |= 1u;
            if (S * (OH - R) + ry >= s.H) f |= 2u;
            if (rx < 0) f |= 4u;
            if (S * (OW - CW) + rx >= W) f |= 8u;
            if (col >= p.NTOT) f |= 16u;
            flags[j] = f;
        }
    }
    __device__ __forceinline__ void issue(int kc) {
        uint32_t p0 = (uint32_t)kc * BK;
        uint32_t n = fdiv(p0, div_ohw);
        uint32_t rem = p0 - n * (uint32_t)OHW;
        uint32_t oy0 = fdiv(rem, div_ow);
        uint32_t ox0 = rem - oy0 * (uint32_t)OW;
        uint32_t soff = (n * (uint32_t)CHW + (oy0 * S) * (uint32_t)W + ox0 * S) * 4u;
        uint32_t cond = 16u | (oy0 == 0 ? 1u : 0u) | ((int)oy0 + R == OH ? 2u : 0u) | (ox0 == 0 ? 4u : 0u) |
                        ((int)ox0 + CW == OW ? 8u : 0u);
#pragma unroll
        for (int j = 0; j < EPT; ++j) r[j] = bload(rsrc, (flags[j] & cond) ? OOB : voff[j], soff);
    }
    __device__ __forceinline__ void commit(float* dst) const {
#pragma unroll
        for (int j = 0; j < EPT; ++j) dst[kl * LD + n_l + 16 * j] = r[j];
    }
};

// ---------------------------------------------------------------------------
// 3-D (cubic kernel KS, stride S, padding P) variants: HoloGAN's ConvTranspose3d(k3, s2, p1, op1)
// x: [N, C, D, H, W] image side, y: [N, K, OD, OH, OW] feature side, w: [K, C, KS, KS, KS]
// ---------------------------------------------------------------------------
struct Conv3DShape {
    int N, C, D, H, W, K, OD, OH, OW;
};

// forward 3-D conv: A[m = (n, od, oy, ox)][k = (c, kd, ky, kx)]
template <int BM, int KS, int S, int P>
struct Conv3DFwdALoader {
    struct Params {
        const float* x;
        Conv3DShape s;
        FastDiv div_odhw, div_ohw, div_ow;
    };
    static constexpr int LD = BM;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr bool DMA = false;
    __device__ __forceinline__ void issue_lds(int, float*) {}
    static constexpr int EPT = BM * BK / NT;
    static constexpr int STEP = NT / BM;
    static constexpr int T3 = KS * KS * KS;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t nbase;
    int kb, m_l, id0, iy0, ix0, C, D, H, W;
    bool m_ok;
    float r[EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const Conv3DShape& s = p.s;
        rsrc = make_rsrc(p.x, (uint32_t)s.N * s.C * s.D * s.H * s.W * 4u);
        m_l = tid % BM;
        kb = tid / BM;
        uint32_t m = (uint32_t)tile * BM + m_l;
        m_ok = m < (uint32_t)s.N * s.OD * s.OH * s.OW;
        uint32_t n = fdiv(m, p.div_odhw);
        uint32_t v = m - n * (uint32_t)(s.OD * s.OH * s.OW);
        uint32_t od = fdiv(v, p.div_ohw);
        v -= od * (uint32_t)(s.OH * s.OW);
        uint32_t oy = fdiv(v, p.div_ow);
        uint32_t ox = v - oy * (uint32_t)s.OW;
        id0 = (int)od * S - P; iy0 = (int)oy * S - P; ix0 = (int)ox * S - P;
        nbase = n * (uint32_t)(s.C * s.D * s.H * s.W);
        C = s.C; D = s.D; H = s.H; W = s.W;
    }
    __device__ __forceinline__ void issue(int kc) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            int k = kc * BK + kb + STEP * j;
            int c = k / T3;
            int tap = k - c * T3;
            int kd = tap / (KS * KS), ky = (tap / KS) % KS, kx = tap % KS;
            int id = id0 + kd, iy = iy0 + ky, ix = ix0 + kx;
            bool ok = m_ok && c < C && (unsigned)id < (unsigned)D && (unsigned)iy < (unsigned)H &&
                      (unsigned)ix < (unsigned)W;
            uint32_t v = ok ? (nbase + (uint32_t)(((c * D + id) * H + iy) * W + ix)) * 4u : OOB;
            r[j] = bload(rsrc, v, 0);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
#pragma unroll
        for (int j = 0; j < EPT; ++j) dst[(kb + STEP * j) * LD + m_l] = r[j];
    }
};

// tap-major variant (see ConvFwdALoaderTap): A[m = (n, od, oy, ox)][k = (tap, c)], c padded to a multiple of BK
template <int BM, int KS, int S, int P>
struct Conv3DFwdALoaderTap {
    using Params = typename Conv3DFwdALoader<BM, KS, S, P>::Params;
    static constexpr int LD = BM;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr bool DMA = GZ_IGEMM_DMA;
    static constexpr int EPT = BM * BK / NT;
    static constexpr int STEP = NT / BM;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[EPT];
    int kb, m_l, id0, iy0, ix0, C, D, H, W, cblocks;
    bool m_ok;
    float r[DMA ? 1 : EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const Conv3DShape& s = p.s;
        const uint32_t shift = (uint32_t)((P * s.H + P) * s.W + P) * 4u;
        rsrc = make_rsrc(reinterpret_cast<const char*>(p.x) - shift,
                         (uint32_t)s.N * s.C * s.D * s.H * s.W * 4u + shift);
        m_l = tid % BM;
        kb = tid / BM;
        uint32_t m = (uint32_t)tile * BM + m_l;
        m_ok = m < (uint32_t)s.N * s.OD * s.OH * s.OW;
        uint32_t n = fdiv(m, p.div_odhw);
        uint32_t v = m - n * (uint32_t)(s.OD * s.OH * s.OW);
        uint32_t od = fdiv(v, p.div_ohw);
        v -= od * (uint32_t)(s.OH * s.OW);
        uint32_t oy = fdiv(v, p.div_ow);
        uint32_t ox = v - oy * (uint32_t)s.OW;
        id0 = (int)od * S - P; iy0 = (int)oy * S - P; ix0 = (int)ox * S - P;
        C = s.C; D = s.D; H = s.H; W = s.W;
        cblocks = round_bk(s.C) / BK;
        const int pos = (int)(n * (uint32_t)(s.C * s.D * s.H * s.W)) + ((id0 + P) * H + (iy0 + P)) * W + (ix0 + P);
#pragma unroll
        for (int j = 0; j < EPT; ++j) voff[j] = (uint32_t)(pos + (kb + STEP * j) * D * H * W) * 4u;
    }
    __device__ __forceinline__ void chunk(int kc, uint32_t& soff, bool& ok, int& cb) const {
        const int tap = kc / cblocks;
        cb = (kc - tap * cblocks) * BK;
        const int kd = tap / (KS * KS), ky = (tap / KS) % KS, kx = tap % KS;
        soff = (uint32_t)(cb * D * H * W + (kd * H + ky) * W + kx) * 4u;
        ok = m_ok && (unsigned)(id0 + kd) < (unsigned)D && (unsigned)(iy0 + ky) < (unsigned)H &&
             (unsigned)(ix0 + kx) < (unsigned)W;
    }
    __device__ __forceinline__ void issue_lds(int kc, float* dst) {
        float* wbase = dst + (m_l - (int)(threadIdx.x & 63));
        uint32_t soff; bool ok; int cb;
        chunk(kc, soff, ok, cb);
#pragma unroll
        for (int j = 0; j < EPT; ++j)
            bload_lds4(rsrc, wbase + (kb + STEP * j) * LD, (ok && cb + kb + STEP * j < C) ? voff[j] : OOB, soff);
    }
    __device__ __forceinline__ void issue(int kc) {
        if constexpr (!DMA) {
            uint32_t soff; bool ok; int cb;
            chunk(kc, soff, ok, cb);
#pragma unroll
            for (int j = 0; j < EPT; ++j) r[j] = bload(rsrc, (ok && cb + kb + STEP * j < C) ? voff[j] : OOB, soff);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
        if constexpr (!DMA) {
#pragma unroll
            for (int j = 0; j < EPT; ++j) dst[(kb + STEP * j) * LD + m_l] = r[j];
        }
    }
};

// transposed 3-D conv, phase (pd, py, px): A[m = (n, a, b, c)][k = (ko, td, ty, tx)]
template <int BM, int KS, int S, int P>
struct Conv3DDgALoader {
    static constexpr int T = (KS + S - 1) / S;
    static constexpr int TAPS = T * T * T;
    struct Params {
        const float* y;
        Conv3DShape s;
        int AD, AH, AW;
        FastDiv div_adhw, div_ahw, div_aw;
    };
    static constexpr int LD = BM;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr bool DMA = false;
    __device__ __forceinline__ void issue_lds(int, float*) {}
    static constexpr int EPT = BM * BK / NT;
    static constexpr int STEP = NT / BM;
    static_assert(BK % TAPS == 0 && T <= 2, "a chunk must hold whole feature channels; 1 or 2 taps per axis");
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[EPT];
    int kb, m_l, K, OSP, lt;     // lt = log2(taps of this phase): a chunk is BK >> lt whole feature channels
    float r[EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int phase, int tid) {
        const Conv3DShape& s = p.s;
        rsrc = make_rsrc(p.y, (uint32_t)s.N * s.K * s.OD * s.OH * s.OW * 4u);
        m_l = tid % BM;
        kb = tid / BM;
        const int pd = phase / (S * S), py = (phase / S) % S, px = phase % S;
        // per-axis tap counts of this phase (1 or 2), as shifts: the phase's weight rows are packed tightly
        const int sd = dg_taps(KS, S, P, pd) - 1, sy = dg_taps(KS, S, P, py) - 1, sx = dg_taps(KS, S, P, px) - 1;
        lt = sd + sy + sx;
        uint32_t m = (uint32_t)tile * BM + m_l;
        bool m_ok = m < (uint32_t)s.N * p.AD * p.AH * p.AW;
        uint32_t n = fdiv(m, p.div_adhw);
        uint32_t v = m - n * (uint32_t)(p.AD * p.AH * p.AW);
        uint32_t a = fdiv(v, p.div_ahw);
        v -= a * (uint32_t)(p.AH * p.AW);
        uint32_t b = fdiv(v, p.div_aw);
        uint32_t c = v - b * (uint32_t)p.AW;
        const int od0 = (int)a + (pd + P) / S, oy0 = (int)b + (py + P) / S, ox0 = (int)c + (px + P) / S;
        OSP = s.OD * s.OH * s.OW;
        K = s.K;
        uint32_t nbase = n * (uint32_t)(s.K * OSP);
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            int kk = kb + STEP * j;
            int kol = kk >> lt, tap = kk & ((1 << lt) - 1);
            int tx = tap & sx, ty = (tap >> sx) & sy, td = tap >> (sx + sy);
            int od = od0 - td, oy = oy0 - ty, ox = ox0 - tx;
            bool ok = m_ok && (unsigned)od < (unsigned)s.OD && (unsigned)oy < (unsigned)s.OH &&
                      (unsigned)ox < (unsigned)s.OW;
            voff[j] = ok ? (nbase + (uint32_t)(kol * OSP + (od * s.OH + oy) * s.OW + ox)) * 4u : OOB;
        }
    }
    __device__ __forceinline__ void issue(int kc) {
        const int ko_base = kc * (BK >> lt);
        uint32_t soff = (uint32_t)ko_base * (uint32_t)OSP * 4u;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            int kol = (kb + STEP * j) >> lt;
            r[j] = bload(rsrc, ko_base + kol < K ? voff[j] : OOB, soff);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
#pragma unroll
        for (int j = 0; j < EPT; ++j) dst[(kb + STEP * j) * LD + m_l] = r[j];
    }
};

// 3-D weight gradient, B[k = p = (n, od, oy, ox)][col = (c, kd, ky, kx)]
template <int BN, int KS, int S, int P>
struct Wg3DBLoader {
    struct Params {
        const float* x;
        Conv3DShape s;
        FastDiv div_odhw, div_ohw, div_ow;
        int KTOT, NTOT;
    };
    static constexpr int LD = BN + 2;
    static constexpr int NPARTS = 0;
    __device__ __forceinline__ void issue_lds_part(int, float*, int) {}
    static constexpr bool DMA = false;
    __device__ __forceinline__ void issue_lds(int, float*) {}
    static constexpr int EPT = BN / 16;
    static constexpr int T3 = KS * KS * KS;
    __amdgpu_buffer_rsrc_t rsrc;
    // Validity of a tap is kept as 3 + 3 bits per element (tap index 0 / KS-1 along d, y, x: the only taps that can
    // leave the volume below / above), eight elements to a register -- the per-element kd/ky/kx arrays of round 1
    // cost 24 more registers and the 128x128 kernel spilled 22 of them inside the reduction loop.
    static_assert(EPT <= 16 && KS == 3 && S == 2 && P == 1, "only tap 0 / tap KS-1 may leave the volume");
    using Mask = std::conditional_t<(EPT > 8), unsigned long long, uint32_t>;      // (BN = 256: 16 elements, 48 bits)
    static constexpr Mask REP = (Mask)0x249249249249ull;                            // 001 repeated: one bit per element
    int toff[EPT];
    Mask lo_bits, hi_bits;           // 3 bits per element j at 3*j: (kd == 0, ky == 0, kx == 0) / (== KS-1)
    int kl, n_l, ODHW, OHW, OW, D, H, W, CDHW, KTOT;
    FastDiv div_odhw, div_ohw, div_ow;
    float r[EPT];
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const Conv3DShape& s = p.s;
        rsrc = make_rsrc(p.x, (uint32_t)s.N * s.C * s.D * s.H * s.W * 4u);
        kl = tid & 15;
        n_l = tid >> 4;
        ODHW = s.OD * s.OH * s.OW; OHW = s.OH * s.OW; OW = s.OW;
        D = s.D; H = s.H; W = s.W; CDHW = s.C * s.D * s.H * s.W;
        KTOT = p.KTOT;
        div_odhw = p.div_odhw; div_ohw = p.div_ohw; div_ow = p.div_ow;
        lo_bits = hi_bits = 0;
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            int col = tile * BN + n_l + 16 * j;
            int c = col / T3;
            int tap = col - c * T3;
            const int kd = tap / (KS * KS), ky = (tap / KS) % KS, kx = tap % KS;
            const bool in = col < p.NTOT;
            toff[j] = in ? ((c * D + kd - P) * H + ky - P) * W + kx - P : INT32_MIN;     // INT32_MIN: column past the end
            lo_bits |= (Mask)((kd == 0) | (ky == 0) << 1 | (kx == 0) << 2) << (3 * j);
            hi_bits |= (Mask)((kd == KS - 1) | (ky == KS - 1) << 1 | (kx == KS - 1) << 2) << (3 * j);
        }
    }
    __device__ __forceinline__ void issue(int kc) {
        uint32_t p = (uint32_t)kc * BK + kl;
        uint32_t n = fdiv(p, div_odhw);
        uint32_t v = p - n * (uint32_t)ODHW;
        uint32_t od = fdiv(v, div_ohw);
        v -= od * (uint32_t)OHW;
        uint32_t oy = fdiv(v, div_ow);
        uint32_t ox = v - oy * (uint32_t)OW;
        int bd = (int)od * S, by = (int)oy * S, bx = (int)ox * S;
        int base = (int)n * CDHW + (bd * H + by) * W + bx;
        bool ok = p < (uint32_t)KTOT;
        // this output position's taps 0 reach below the volume when base coordinate - P < 0, taps KS-1 above it when
        // base + KS-1-P >= extent
        const uint32_t pos_lo = (uint32_t)(bd < P) | (uint32_t)(by < P) << 1 | (uint32_t)(bx < P) << 2;
        const uint32_t pos_hi = (uint32_t)(bd + KS - 1 - P >= D) | (uint32_t)(by + KS - 1 - P >= H) << 1 |
                                (uint32_t)(bx + KS - 1 - P >= W) << 2;
        const Mask bad = (lo_bits & ((Mask)pos_lo * REP)) | (hi_bits & ((Mask)pos_hi * REP));
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            bool v2 = ok && toff[j] != INT32_MIN && ((uint32_t)(bad >> (3 * j)) & 7u) == 0;
            r[j] = bload(rsrc, v2 ? (uint32_t)(base + toff[j]) * 4u : OOB, 0);
        }
    }
    __device__ __forceinline__ void commit(float* dst) const {
#pragma unroll
        for (int j = 0; j < EPT; ++j) dst[kl * LD + n_l + 16 * j] = r[j];
    }
};

}  // namespace gz
