// C-ABI entry points for the cubic-kernel 3-D convolution family on the fp32 MFMA implicit-GEMM core:
// HoloGAN's ConvTranspose3d(k3, s2, p1, output_padding 1) forward (= Dg), its input gradient (= F)
// and weight gradient (= Wg).  Reference call sites: core/models/hologan_generator.py:29-30,55-58.
#include "gz_igemm.h"
#include "../../include/gz_ops.h"

namespace gz {

using C3_128x128 = TileCfg<2, 2, 2, 2>;
using C3_128x64 = TileCfg<2, 2, 2, 1>;
using C3_128x32 = TileCfg<4, 1, 1, 1>;
using C3_64x64 = TileCfg<2, 2, 1, 1>;

static inline int r4(int v) { return (v + 3) & ~3; }

// split-K of under-filled F / Dg launches: same policy as gz_conv.hip (plan_split)
static int plan3(int tile, long long M, long long N, int Kdim, int ny) {
    const bool off = knobs().no_splitk;
    if (off) return 1;
    const int bm = tile == 3 ? 64 : 128, bn = tile == 0 ? 128 : (tile == 2 ? 32 : 64);
    const long long tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * ny;
    const int chunks = (Kdim + BK - 1) / BK;
    if (chunks < 16 || tiles >= 2 * cus()) return 1;
    long long want = (4 * cus() + tiles - 1) / tiles, cap = chunks / 8;
    long long sp = want < cap ? want : cap;
    return sp < 2 ? 1 : (int)sp;
}

static size_t bytes3(int splits, long long M, long long N, int Kdim, int ny) {
    return splits > 1 ? (size_t)split_nz(Kdim, splits) * ny * M * N * 4 : 0;
}

// Split-K of the transposed convolution's 8 phases (1, 2, 2, 2, 4, 4, 4, 8 taps for k3 s2 p1): the chunk count per
// workgroup is chosen from the TOTAL work so that ~1024 workgroups of equal length come out; a phase gets as many
// slabs as its own reduction needs (GridMap::phase_nz), not the longest phase's count.
struct DgPlan3 {
    int splits;         // of the longest phase
    long long slabs;    // sum over the phases
};

static DgPlan3 plan3_dg(int tile, long long Mp, long long C, int K, int KS, int S, int P) {
    DgPlan3 none{1, 0};
    const bool off = knobs().no_splitk;
    const bool even = knobs().dg3_even_split;      // experiment: the round-1 plan (same slab count per phase)
    const int bm = tile == 3 ? 64 : 128, bn = tile == 0 ? 128 : (tile == 2 ? 32 : 64);
    const long long tp = ((Mp + bm - 1) / bm) * ((C + bn - 1) / bn);
    int pc[8], maxpc = 0;
    long long total = 0;
    for (int ph = 0; ph < S * S * S; ++ph) {
        pc[ph] = (K * dg_taps(KS, S, P, ph / (S * S)) * dg_taps(KS, S, P, (ph / S) % S) * dg_taps(KS, S, P, ph % S) +
                  BK - 1) / BK;
        total += pc[ph];
        maxpc = pc[ph] > maxpc ? pc[ph] : maxpc;
    }
    if (off || maxpc < 16 || tp * S * S * S >= 2 * cus()) return none;
    int splits;
    if (even) {
        splits = plan3(tile, Mp, C, K * 8, 8);
    } else {
        long long cps = (total * tp + 4 * cus() - 1) / (4 * cus());
        if (cps < 8) cps = 8;
        splits = (int)((maxpc + cps - 1) / cps);
    }
    if (splits < 2) return none;
    const int cps = (maxpc + splits - 1) / splits;
    long long slabs = 0;
    for (int ph = 0; ph < S * S * S; ++ph) slabs += (pc[ph] + cps - 1) / cps;
    return DgPlan3{splits, slabs};
}

static int pick3(long long M, long long N, int ny) {
    if (N <= 32) return 2;
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * ny; };
    if (N <= 64) return tiles(128, 64) >= cus() ? 1 : 3;
    if (tiles(128, 128) >= cus()) return 0;
    if (tiles(128, 64) >= cus()) return 1;
    return 3;
}

// wp[col][ld] = w[r][col] (r < R rows of length COLS), zero padded to ld
__global__ __launch_bounds__(256) void transpose_pad3_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int R, int COLS, int ld) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < COLS) ? src[(long long)r * COLS + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < COLS && r < ld) dst[(long long)c * ld + r] = tile[tx][ty + 8 * i];
    }
}

// forward weights in tap-major order (Conv3DFwdALoaderTap): wp[(tap, c)][ld] = w[ko][c][tap], c padded to BK
static bool fwd3_tap_major(int C) {
    const bool off = knobs().no_tapmajor;
    return !off && C >= BK;
}

__global__ __launch_bounds__(256) void pack_fwd3_tap_kernel(const float* __restrict__ w, float* __restrict__ wp, int K,
                                                            int C, int taps, int cpad, int ld) {
    const long long total = (long long)taps * cpad * ld;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int ko = (int)(i % ld);
        long long row = i / ld;
        int c = (int)(row % cpad), tap = (int)(row / cpad);
        wp[i] = (ko < K && c < C) ? w[((long long)ko * C + c) * taps + tap] : 0.f;
    }
}

// wp[phase][(ko, td, ty, tx)][ldc] = w[ko][c][kd][ky][kx], k* = ((p* + P) % S) + S * t*.  Only the taps with
// k* < KS exist (k3 s2 p1: 1 tap on an even output coordinate, 2 on an odd one -> 1..8 per phase, 27 over the 8
// phases instead of 8 x 8); a phase's rows are packed tightly, the rest of its fixed-size region is zero.
__global__ __launch_bounds__(256) void pack_dgrad3_kernel(const float* __restrict__ w, float* __restrict__ wp, int K,
                                                          int C, int KS, int S, int P, int T, int ldc) {
    const int ko = blockIdx.x, phase = blockIdx.y;
    const int pd = phase / (S * S), py = (phase / S) % S, px = phase % S;
    const int rd = (pd + P) % S, ry = (py + P) % S, rx = (px + P) % S;
    const int nd = dg_taps(KS, S, P, pd), ny = dg_taps(KS, S, P, py), nx = dg_taps(KS, S, P, px);
    const int taps = nd * ny * nx, pad = T * T * T - taps;
    float* dst = wp + (long long)phase * K * T * T * T * ldc;
    for (int i = threadIdx.x; i < taps * ldc; i += blockDim.x) {
        int tap = i / ldc, c = i - tap * ldc;
        int kd = rd + S * (tap / (ny * nx)), ky = ry + S * ((tap / nx) % ny), kx = rx + S * (tap % nx);
        dst[((long long)ko * taps + tap) * ldc + c] =
            c < C ? w[((((long long)ko * C + c) * KS + kd) * KS + ky) * KS + kx] : 0.f;
    }
    for (int i = threadIdx.x; i < pad * ldc; i += blockDim.x)
        dst[((long long)K * taps + (long long)ko * pad) * ldc + i] = 0.f;
}

__global__ __launch_bounds__(256) void reduce_slabs3_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                            int S, long long count) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += slab[(long long)s * count + i];
    out[i] = acc;
}

static bool shape3_ok(const Conv3DShape& s, int KS, int S, int P) {
    if (s.N <= 0 || s.C <= 0 || s.K <= 0 || s.D <= 0 || s.H <= 0 || s.W <= 0) return false;
    auto o = [&](int v) { return (v + 2 * P - KS) / S + 1; };
    return s.OD == o(s.D) && s.OH == o(s.H) && s.OW == o(s.W);
}

static bool big3(long long e) { return e * 4 >= (1ll << 31); }

template <class Cfg, int KS, int S, int P>
static int run_fwd3(const float* x, const float* wp, const float* bias, float* y, const Conv3DShape& s, int act,
                    float slope, hipStream_t st, int splits, float* slab) {
    using AL = Conv3DFwdALoader<Cfg::BM, KS, S, P>;
    using BL = MContigLoader4<Cfg::BN>;
    const int osp = s.OD * s.OH * s.OW;
    typename AL::Params pa{x, s, make_fastdiv(osp), make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW)};
    int M = s.N * osp;
    EpiNCHW::Params pe{y, M, s.K, osp, make_fastdiv(osp), bias, act, slope};
    if (fwd3_tap_major(s.C)) {
        using ALT = Conv3DFwdALoaderTap<Cfg::BM, KS, S, P>;
        int Kt = KS * KS * KS * round_bk(s.C);
        typename BL::Params pbt{wp, Kt, r4(s.K), r4(s.K), 0};
        return launch_igemm<Cfg, ALT, BL, EpiNCHW>(pa, pbt, pe, M, s.K, Kt, 1, splits, st, slab);
    }
    int Kg = s.C * KS * KS * KS;
    typename BL::Params pb{wp, Kg, r4(s.K), r4(s.K), 0};
    return launch_igemm<Cfg, AL, BL, EpiNCHW>(pa, pb, pe, M, s.K, Kg, 1, splits, st, slab);
}

template <class Cfg, int KS, int S, int P>
static int run_dgrad3(const float* y, const float* wp, const float* bias, float* x, const Conv3DShape& s, int act,
                      float slope, hipStream_t st, int splits, float* slab) {
    using AL = Conv3DDgALoader<Cfg::BM, KS, S, P>;
    using BL = MContigLoader4<Cfg::BN>;
    using Epi = EpiPhase3D<S>;
    const int AD = s.D / S, AH = s.H / S, AW = s.W / S;
    typename AL::Params pa{y, s, AD, AH, AW, make_fastdiv(AD * AH * AW), make_fastdiv(AH * AW), make_fastdiv(AW)};
    int Kg = s.K * AL::TAPS;
    int ldc = r4(s.C);
    typename BL::Params pb{wp, Kg, ldc, ldc, (long long)Kg * ldc};
    int M = s.N * AD * AH * AW;
    typename Epi::Params pe{x, M, s.C, s.D, s.H, s.W, AD, AH, AW, make_fastdiv(AD * AH * AW), make_fastdiv(AH * AW),
                            make_fastdiv(AW), bias, act, slope};
    int pc[8];
    for (int ph = 0; ph < S * S * S; ++ph)
        pc[ph] = (s.K * dg_taps(KS, S, P, ph / (S * S)) * dg_taps(KS, S, P, (ph / S) % S) * dg_taps(KS, S, P, ph % S) +
                  BK - 1) / BK;
    return launch_igemm<Cfg, AL, BL, Epi>(pa, pb, pe, M, s.C, Kg, S * S * S, splits, st, slab, pc);
}

static int splits3(long long tiles, int chunks) {
    const int target = knobs().wg3_target * cus() / 256;
    if (tiles >= cus()) return 1;
    long long want = (target + tiles - 1) / tiles;
    long long cap = chunks / 8 > 0 ? chunks / 8 : 1;
    long long s = want < cap ? want : cap;
    return (int)(s < 1 ? 1 : s);
}

template <class Cfg, int KS, int S, int P>
static int run_wgrad3(const float* x, const float* y, float* dw, float* ws, size_t ws_bytes, const Conv3DShape& s,
                      hipStream_t st) {
    using AL = WgALoader<Cfg::BM>;
    using BL = Wg3DBLoader<Cfg::BN, KS, S, P>;
    const int osp = s.OD * s.OH * s.OW;
    const int KTOT = s.N * osp;
    const int NTOT = s.C * KS * KS * KS;
    ConvShape flat{s.N, s.C, 1, 1, s.K, osp, 1};      // WgALoader only needs N, K and OH*OW
    typename AL::Params pa{y, flat, make_fastdiv(osp), KTOT};
    typename BL::Params pb{x, s, make_fastdiv(osp), make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW), KTOT, NTOT};
    long long tiles = (long long)((s.K + Cfg::BM - 1) / Cfg::BM) * ((NTOT + Cfg::BN - 1) / Cfg::BN);
    int chunks = (KTOT + BK - 1) / BK;
    int splits = splits3(tiles, chunks);
    long long count = (long long)s.K * NTOT;
    if (splits > 1) {
        long long max_splits = (long long)(ws_bytes / 4) / count;
        if (max_splits < 2) splits = 1;
        else if (splits > max_splits) splits = (int)max_splits;
    }
    int cps = (chunks + splits - 1) / splits;
    int nz = (chunks + cps - 1) / cps;
    float* out = nz > 1 ? ws : dw;
    EpiRowMajor::Params pe{out, s.K, NTOT, NTOT, count, nullptr, ACT_NONE, 0.f};
    int rc = launch_igemm<Cfg, AL, BL, EpiRowMajor>(pa, pb, pe, s.K, NTOT, KTOT, 1, splits, st);
    if (rc != GZ_OK) return rc;
    if (nz > 1) {
        hipLaunchKernelGGL(reduce_slabs3_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ws, dw, nz,
                           count);
        rc = launch_status();
    }
    return rc;
}

// ---- the igemm2 skeleton for the two gather-fed directions (round 4) -----------------------------------------
// The round-1 kernels above run HoloGAN's two ConvTranspose3d layers at 89-100 TFLOP/s forward and 95-104 on the input
// gradient; the one-wavefront-per-SIMD skeleton with 4-byte LDS-DMA gathers (Conv3DDgTapA2 / Conv3DTapA2) takes them
// when the GEMM's column count is a multiple of 64.  tile: 0 = not taken, 1 = 256x128, 2 = 256x64.
using C3_256x128 = TileCfg2<2, 2, 2, 2>;
using C3_256x64 = TileCfg2<2, 2, 1, 3>;

struct Plan3T {
    int tile, splits;
    long long slabs;      // split launches of the transposed form: slabs summed over the phases
};

// transposed form: weight rows tap-major per phase (pack_dgrad3_tap_kernel) -- the pack and the launch must agree.
// Multiples of 128 columns only: with 64 (HoloGAN's block2, 8-64 chunks per workgroup) the 256x64 tile measured 164 us
// against 147 us on the round-1 kernel, and a 512x64 tile needs more gather pieces per k-step than the stream holds.
static bool dg3_tap2(int K, int C) {
    return !knobs().no_igemm2 && !knobs().no_igemm2_tap && K >= BK && (C & 127) == 0;
}

static Plan3T plan3_dgtap2(long long Mp, int C, int K, int KS, int S, int P) {
    if (!dg3_tap2(K, C) || S * S * S > 8) return Plan3T{0, 1, 0};
    const int kblocks = round_bk(K) / BK;
    int pc[8], maxpc = 0;
    long long total = 0;
    for (int ph = 0; ph < S * S * S; ++ph) {
        pc[ph] = dg_taps(KS, S, P, ph / (S * S)) * dg_taps(KS, S, P, (ph / S) % S) * dg_taps(KS, S, P, ph % S) * kblocks;
        total += pc[ph];
        maxpc = pc[ph] > maxpc ? pc[ph] : maxpc;
    }
    const long long tm = (Mp + 255) / 256;
    // 256x128 also when its tiles alone do not fill the chip (HoloGAN block1 at bs 64, 128 tiles, 27 slabs: 184 us
    // against 193 us on 256x64 tiles with 21 slabs; the round-1 kernel: 223 us)
    const int tile = knobs().dg3_tile == 2 ? 2 : 1;
    const long long tp = tm * (tile == 1 ? C / 128 : C / 64);
    // phases of 1..8 taps: with >= 4 workgroups per CU the longest-first launch order balances them; below that the
    // reduction is cut into pieces of equal length (>= 32 chunks), ~2 workgroups per CU
    if (knobs().no_splitk || tp * S * S * S >= 4LL * cus()) return Plan3T{tile, 1, 0};
    const int wgs = knobs().dg3_wgs * cus();
    long long cps = (total * tp + wgs - 1) / wgs;
    if (cps < knobs().dg3_min_chunks) cps = knobs().dg3_min_chunks;
    const int splits = (int)((maxpc + cps - 1) / cps);
    if (splits < 2) return Plan3T{tile, 1, 0};
    const int per = (maxpc + splits - 1) / splits;
    long long slabs = 0;
    for (int ph = 0; ph < S * S * S; ++ph) slabs += (pc[ph] + per - 1) / per;
    return Plan3T{tile, splits, slabs};
}

// plain strided form (the transposed layer's input gradient): the tap-major forward image is the one already packed
static Plan3T plan3_fwdtap2(long long M, int K, int C, int KS) {
    if (knobs().no_igemm2 || knobs().no_igemm2_tap || !fwd3_tap_major(C) || (K & 63)) return Plan3T{0, 1, 0};
    const int chunks = KS * KS * KS * round_bk(C) / BK;
    const long long tm = (M + 255) / 256, t64 = tm * (K / 64), t128 = (K & 127) ? 0 : tm * (K / 128);
    const int cu = cus();
    if (t128 >= cu * 7 / 8) return Plan3T{1, 1, 0};
    if (t64 >= cu * 7 / 8) return Plan3T{2, 1, 0};
    if (knobs().no_splitk) return Plan3T{0, 1, 0};
    for (int tile = t128 ? 1 : 2; tile <= 2; ++tile) {
        const long long tiles = tile == 1 ? t128 : t64;
        int splits = (int)((cu + tiles - 1) / tiles);
        while (splits > 1 && chunks / splits < 48) --splits;
        if (splits > 1 && tiles * splits >= cu * 3 / 4) return Plan3T{tile, splits, 0};
    }
    return Plan3T{0, 1, 0};
}

// wp[phase][(tap, ko)][ldc] = w[ko][c][kd][ky][kx] over the phase's own nd x ny x nx taps (tap = (td * ny + ty) * nx +
// tx, k* = ((p* + P) % S) + S * t*), ko padded to a multiple of BK with zero rows; the unused tail of a phase's
// fixed-size T^3 * kpad-row region is never read
__global__ __launch_bounds__(256) void pack_dgrad3_tap_kernel(const float* __restrict__ w, float* __restrict__ wp, int K,
                                                              int C, int KS, int S, int P, int T, int kpad, int ldc) {
    const int phase = blockIdx.y;
    const int pd = phase / (S * S), py = (phase / S) % S, px = phase % S;
    const int rd = (pd + P) % S, ry = (py + P) % S, rx = (px + P) % S;
    const int nd = dg_taps(KS, S, P, pd), ny = dg_taps(KS, S, P, py), nx = dg_taps(KS, S, P, px);
    float* dst = wp + (long long)phase * T * T * T * kpad * ldc;
    const long long total = (long long)nd * ny * nx * kpad * ldc;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % ldc);
        const long long row = i / ldc;
        const int ko = (int)(row % kpad), tap = (int)(row / kpad);
        const int kd = rd + S * (tap / (ny * nx)), ky = ry + S * ((tap / nx) % ny), kx = rx + S * (tap % nx);
        dst[i] = (ko < K && c < C) ? w[((((long long)ko * C + c) * KS + kd) * KS + ky) * KS + kx] : 0.f;
    }
}

__global__ __launch_bounds__(256) void act_inplace3_kernel(float* __restrict__ x, long long total4, int act, float slope) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += stride) {
        f32x4 v = reinterpret_cast<f32x4*>(x)[i];
        v[0] = act_fwd(v[0], act, slope); v[1] = act_fwd(v[1], act, slope);
        v[2] = act_fwd(v[2], act, slope); v[3] = act_fwd(v[3], act, slope);
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}

// weight gradient on the igemm2 skeleton with the register-staged loaders above (igemm2r_kernel, as the 2-D layers
// without an LDS-DMA image).  Split so that two workgroups per CU come out, >= 24 chunks each: HoloGAN block1 (54 tiles
// x 9 splits) 238 -> 178 us, block2 (7 tiles x 73) 210 -> 205 us; pieces of 48-64 chunks leave a partial second round
// of workgroups and measured 240-310 us.
// (256 x 128 tiles, or 128 x 256 with fewer than 256 output channels)
using C3_128x256 = TileCfg2<1, 4, 2, 2>;
static bool wg3r_ok(const Conv3DShape& s) { return !knobs().no_igemm2 && !knobs().no_wg3r && s.K >= 128; }

static int wg3r_splits(const Conv3DShape& s, int KS) {
    const long long ntot = (long long)s.C * KS * KS * KS;
    const long long tiles = s.K >= 256 ? (long long)((s.K + 255) / 256) * ((ntot + 127) / 128)
                                       : (long long)((s.K + 127) / 128) * ((ntot + 255) / 256);
    const int chunks = (s.N * s.OD * s.OH * s.OW + BK - 1) / BK;
    int splits = (int)(knobs().wg3r_wgs * cus() / tiles);
    if (splits < 1) splits = 1;
    while (splits > 1 && chunks / splits < knobs().wg3r_min_chunks) --splits;
    return splits;
}

template <class Cfg, int KS, int S, int P>
static int run_wgrad3r(const float* x, const float* y, float* dw, float* ws, size_t ws_bytes, const Conv3DShape& s,
                       hipStream_t st) {
    using AL = WgALoader<Cfg::BM>;
    using BL = Wg3DBLoader<Cfg::BN, KS, S, P>;
    const int osp = s.OD * s.OH * s.OW;
    const int KTOT = s.N * osp;
    const int NTOT = s.C * KS * KS * KS;
    ConvShape flat{s.N, s.C, 1, 1, s.K, osp, 1};
    typename AL::Params pa{y, flat, make_fastdiv(osp), KTOT};
    typename BL::Params pb{x, s, make_fastdiv(osp), make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW), KTOT, NTOT};
    const int chunks = (KTOT + BK - 1) / BK;
    int splits = wg3r_splits(s, KS);
    const long long count = (long long)s.K * NTOT;
    if (splits > 1) {
        const long long max_splits = (long long)(ws_bytes / 4) / count;
        if (max_splits < 2) splits = 1;
        else if (splits > max_splits) splits = (int)max_splits;
    }
    const int cps = (chunks + splits - 1) / splits;
    const int nz = (chunks + cps - 1) / cps;
    float* out = nz > 1 ? ws : dw;
    EpiRowMajorB::Params pe{out, s.K, NTOT, NTOT, count, nullptr, ACT_NONE, 0.f};
    int rc = launch_igemm2r<Cfg, AL, BL, EpiRowMajorB>(pa, pb, pe, s.K, NTOT, KTOT, splits, st);
    if (rc != GZ_OK) return rc;
    if (nz > 1) {
        hipLaunchKernelGGL(reduce_slabs3_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ws, dw, nz,
                           count);
        rc = launch_status();
    }
    return rc;
}

template <class Cfg, int KS, int S, int P>
static int run_dgradtap3(const float* y, const float* wp, const float* bias, float* x, const Conv3DShape& s, int act,
                         float slope, hipStream_t st, int splits, float* slab) {
    using AL = Conv3DDgTapA2<Cfg::BM, KS, S, P>;
    using BL = MContigB2<Cfg::BN>;
    using Epi = EpiPhase3DB<S>;
    const int AD = s.D / S, AH = s.H / S, AW = s.W / S;
    typename AL::Params pa{y, s, AD, AH, AW, make_fastdiv(AD * AH * AW), make_fastdiv(AH * AW), make_fastdiv(AW)};
    const int kpad = round_bk(s.K), Kt = AL::T * AL::T * AL::T * kpad, ldc = r4(s.C);
    typename BL::Params pb{wp, Kt, ldc, ldc, (long long)Kt * ldc};
    const int M = s.N * AD * AH * AW;
    typename Epi::Params pe{x, M, s.C, s.D, s.H, s.W, AD, AH, AW, make_fastdiv(AD * AH * AW), make_fastdiv(AH * AW),
                            make_fastdiv(AW), bias, act, slope};
    int pc[8];
    for (int ph = 0; ph < S * S * S; ++ph)
        pc[ph] = dg_taps(KS, S, P, ph / (S * S)) * dg_taps(KS, S, P, (ph / S) % S) * dg_taps(KS, S, P, ph % S) * (kpad / BK);
    const int rc = launch_igemm2<Cfg, AL, BL, Epi>(pa, pb, pe, M, s.C, Kt, S * S * S, splits, st, slab, pc);
    if (rc != GZ_OK || act == ACT_NONE) return rc;
    // (EpiPhase3DB adds the bias only; no shipped model puts an activation on a ConvTranspose3d)
    const long long total4 = (long long)s.N * s.C * s.D * s.H * s.W / 4;       // D, H, W even: a multiple of 8 elements
    hipLaunchKernelGGL(act_inplace3_kernel, dim3((unsigned)((total4 + 255) / 256 > 4096 ? 4096 : (total4 + 255) / 256)),
                       dim3(256), 0, st, x, total4, act, slope);
    return launch_status();
}

template <class Cfg, int KS, int S, int P>
static int run_fwdtap3(const float* x, const float* wp, const float* bias, float* y, const Conv3DShape& s, int act,
                       float slope, hipStream_t st, int splits, float* slab) {
    using AL = Conv3DTapA2<Cfg::BM, KS, S, P>;
    using BL = MContigB2<Cfg::BN>;
    const int osp = s.OD * s.OH * s.OW;
    typename AL::Params pa{x, s, make_fastdiv(osp), make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW)};
    const int M = s.N * osp, Kt = KS * KS * KS * round_bk(s.C);
    EpiNCHWB::Params pe{y, M, s.K, osp, make_fastdiv(osp), bias, act, slope};
    typename BL::Params pb{wp, Kt, r4(s.K), r4(s.K), 0};
    return launch_igemm2<Cfg, AL, BL, EpiNCHWB>(pa, pb, pe, M, s.K, Kt, 1, splits, st, slab);
}

}  // namespace gz

using namespace gz;

#define GZ3_TILE_SWITCH(T, CALL)              \
    switch (T) {                              \
        case 0: return CALL(C3_128x128);      \
        case 1: return CALL(C3_128x64);       \
        case 2: return CALL(C3_128x32);       \
        default: return CALL(C3_64x64);       \
    }

extern "C" {

long long gz_conv3d_pack_fwd_elems(int K, int C, int KS) {
    return (long long)(fwd3_tap_major(C) ? round_bk(C) : C) * KS * KS * KS * r4(K);
}

long long gz_conv3d_pack_dgrad_elems(int K, int C, int KS, int S) {
    int T = (KS + S - 1) / S;
    return (long long)S * S * S * (dg3_tap2(K, C) ? round_bk(K) : K) * T * T * T * r4(C);
}

int gz_conv3d_pack_fwd(const float* w, float* wp, int K, int C, int KS, hipStream_t stream) {
    gz::clear_stale_error();
    if (K <= 0 || C <= 0 || KS <= 0) return GZ_ERR_BAD_SHAPE;
    int Kg = C * KS * KS * KS, ld = r4(K);
    if (fwd3_tap_major(C)) {
        long long total = (long long)KS * KS * KS * round_bk(C) * ld;
        hipLaunchKernelGGL(pack_fwd3_tap_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)),
                           dim3(256), 0, stream, w, wp, K, C, KS * KS * KS, round_bk(C), ld);
        return launch_status();
    }
    hipLaunchKernelGGL(transpose_pad3_kernel, dim3((Kg + 31) / 32, (ld + 31) / 32), dim3(256), 0, stream, w, wp, K, Kg,
                       ld);
    return launch_status();
}

int gz_conv3d_pack_dgrad(const float* w, float* wp, int K, int C, int KS, int S, int P, hipStream_t stream) {
    gz::clear_stale_error();
    if (K <= 0 || C <= 0 || KS <= 0 || S <= 0) return GZ_ERR_BAD_SHAPE;
    int T = (KS + S - 1) / S;
    if (dg3_tap2(K, C)) {
        const long long total = (long long)T * T * T * round_bk(K) * r4(C);
        const unsigned bx = (unsigned)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
        hipLaunchKernelGGL(pack_dgrad3_tap_kernel, dim3(bx, S * S * S), dim3(256), 0, stream, w, wp, K, C, KS, S, P, T,
                           round_bk(K), r4(C));
        return launch_status();
    }
    hipLaunchKernelGGL(pack_dgrad3_kernel, dim3(K, S * S * S), dim3(256), 0, stream, w, wp, K, C, KS, S, P, T, r4(C));
    return launch_status();
}

size_t gz_conv3d_fwd_workspace_bytes(int N, int C, int K, int OD, int OH, int OW, int KS) {
    long long M = (long long)N * OD * OH * OW;
    int Kg = (fwd3_tap_major(C) ? round_bk(C) : C) * KS * KS * KS;
    const Plan3T p2 = plan3_fwdtap2(M, K, C, KS);
    if (p2.tile) return bytes3(p2.splits, M, K, Kg, 1);
    return bytes3(plan3(pick3(M, K, 1), M, K, Kg, 1), M, K, Kg, 1);
}

size_t gz_conv3d_dgrad_workspace_bytes(int N, int C, int K, int OD, int OH, int OW, int KS) {
    long long M = (long long)N * OD * OH * OW;      // per phase: the S = 2 output grid has OD*OH*OW cells per phase
    const Plan3T p2 = plan3_dgtap2(M, C, K, KS, 2, 1);
    if (p2.tile) return p2.splits > 1 ? (size_t)p2.slabs * M * C * 4 : 0;
    DgPlan3 pl = plan3_dg(pick3(M, C, 8), M, C, K, KS, 2, 1);
    return pl.splits > 1 ? (size_t)pl.slabs * M * C * 4 : 0;
}

int gz_conv3d_fwd(const float* x, const float* wpack, const float* bias, float* y, float* workspace, size_t ws_bytes,
                  int N, int C, int D, int H, int W, int K, int OD, int OH, int OW, int KS, int S, int P, int act,
                  float slope, hipStream_t stream) {
    gz::clear_stale_error();
    Conv3DShape s{N, C, D, H, W, K, OD, OH, OW};
    if (KS != 3 || S != 2 || P != 1) return GZ_ERR_UNSUPPORTED;
    if (!shape3_ok(s, KS, S, P)) return GZ_ERR_BAD_SHAPE;
    if (big3((long long)N * C * D * H * W) || big3((long long)N * K * OD * OH * OW)) return GZ_ERR_TOO_LARGE;
    if (((uintptr_t)wpack & 15) || ((uintptr_t)y & 15)) return GZ_ERR_BAD_SHAPE;
    const int kdim = (fwd3_tap_major(C) ? round_bk(C) : C) * 27;
    const Plan3T p2 = plan3_fwdtap2((long long)N * OD * OH * OW, K, C, KS);
    if (p2.tile && !((uintptr_t)x & 3)) {
        int sp = p2.splits;
        if (sp > 1 && (!workspace || ws_bytes < bytes3(sp, (long long)N * OD * OH * OW, K, kdim, 1))) return GZ_ERR_WORKSPACE;
        float* sl = sp > 1 ? workspace : nullptr;
        return p2.tile == 1 ? run_fwdtap3<C3_256x128, 3, 2, 1>(x, wpack, bias, y, s, act, slope, stream, sp, sl)
                            : run_fwdtap3<C3_256x64, 3, 2, 1>(x, wpack, bias, y, s, act, slope, stream, sp, sl);
    }
    int t = pick3((long long)N * OD * OH * OW, K, 1);
    int splits = plan3(t, (long long)N * OD * OH * OW, K, kdim, 1);
    if (splits > 1 && (!workspace || ws_bytes < bytes3(splits, (long long)N * OD * OH * OW, K, kdim, 1))) splits = 1;
    float* slab = splits > 1 ? workspace : nullptr;
#define CALL(CFG) run_fwd3<CFG, 3, 2, 1>(x, wpack, bias, y, s, act, slope, stream, splits, slab)
    GZ3_TILE_SWITCH(t, CALL)
#undef CALL
}

int gz_conv3d_dgrad(const float* y, const float* wpack, const float* bias, float* x, float* workspace,
                    size_t ws_bytes, int N, int C, int D, int H, int W, int K, int OD, int OH, int OW, int KS, int S,
                    int P, int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    Conv3DShape s{N, C, D, H, W, K, OD, OH, OW};
    if (KS != 3 || S != 2 || P != 1) return GZ_ERR_UNSUPPORTED;
    if (!shape3_ok(s, KS, S, P) || D % S || H % S || W % S) return GZ_ERR_BAD_SHAPE;
    if (big3((long long)N * C * D * H * W) || big3((long long)N * K * OD * OH * OW)) return GZ_ERR_TOO_LARGE;
    if ((uintptr_t)wpack & 15) return GZ_ERR_BAD_SHAPE;
    const long long Mp = (long long)N * (D / S) * (H / S) * (W / S);
    if (dg3_tap2(K, C)) {       // (the packed image is tap-major: this path or none)
        const Plan3T p2 = plan3_dgtap2(Mp, C, K, KS, S, P);
        if (!p2.tile) return GZ_ERR_UNSUPPORTED;
        if (p2.splits > 1 && (!workspace || ws_bytes < (size_t)p2.slabs * Mp * C * 4)) return GZ_ERR_WORKSPACE;
        float* sl = p2.splits > 1 ? workspace : nullptr;
        return p2.tile == 1 ? run_dgradtap3<C3_256x128, 3, 2, 1>(y, wpack, bias, x, s, act, slope, stream, p2.splits, sl)
                            : run_dgradtap3<C3_256x64, 3, 2, 1>(y, wpack, bias, x, s, act, slope, stream, p2.splits, sl);
    }
    int t = pick3(Mp, C, S * S * S);
    DgPlan3 pl = plan3_dg(t, Mp, C, K, KS, S, P);
    int splits = pl.splits;
    if (splits > 1 && (!workspace || ws_bytes < (size_t)pl.slabs * Mp * C * 4)) splits = 1;
    float* slab = splits > 1 ? workspace : nullptr;
#define CALL(CFG) run_dgrad3<CFG, 3, 2, 1>(y, wpack, bias, x, s, act, slope, stream, splits, slab)
    GZ3_TILE_SWITCH(t, CALL)
#undef CALL
}

size_t gz_conv3d_wgrad_workspace_bytes(int N, int C, int K, int OD, int OH, int OW, int KS) {
    long long count = (long long)K * C * KS * KS * KS;
    int chunks = (N * OD * OH * OW + BK - 1) / BK;
    long long tiles = (long long)((K + 127) / 128) * ((C * KS * KS * KS + 127) / 128);
    int splits = splits3(tiles, chunks);
    const Conv3DShape s{N, C, 2 * OD, 2 * OH, 2 * OW, K, OD, OH, OW};
    if (wg3r_ok(s)) splits = wg3r_splits(s, KS);
    return splits > 1 ? (size_t)splits * count * 4 : 0;
}

int gz_conv3d_wgrad(const float* x, const float* y, float* dw, float* workspace, size_t ws_bytes, int N, int C, int D,
                    int H, int W, int K, int OD, int OH, int OW, int KS, int S, int P, hipStream_t stream) {
    gz::clear_stale_error();
    Conv3DShape s{N, C, D, H, W, K, OD, OH, OW};
    if (KS != 3 || S != 2 || P != 1) return GZ_ERR_UNSUPPORTED;
    if (!shape3_ok(s, KS, S, P)) return GZ_ERR_BAD_SHAPE;
    if (big3((long long)N * C * D * H * W) || big3((long long)N * K * OD * OH * OW)) return GZ_ERR_TOO_LARGE;
    if (wg3r_ok(s))
        return K >= 256 ? run_wgrad3r<C3_256x128, 3, 2, 1>(x, y, dw, workspace, ws_bytes, s, stream)
                        : run_wgrad3r<C3_128x256, 3, 2, 1>(x, y, dw, workspace, ws_bytes, s, stream);
    long long NTOT = (long long)C * KS * KS * KS;
    int t = NTOT <= 32 ? 2 : ((NTOT <= 64 || K <= 64) ? (K <= 64 ? 3 : 1) : 0);
    const int force = knobs().wg3_tile;      // experiment
    if (force >= 0 && t == 0) t = force;
#define CALL(CFG) run_wgrad3<CFG, 3, 2, 1>(x, y, dw, workspace, ws_bytes, s, stream)
    GZ3_TILE_SWITCH(t, CALL)
#undef CALL
}

}  // extern "C"
