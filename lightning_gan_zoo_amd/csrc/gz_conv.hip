// C-ABI entry points for the convolution family {F, Dg, Wg} and the plain GEMM,
// all on the fp32 MFMA implicit-GEMM core (gz_igemm.h).
//
//   F  : y  = conv(x, w)                 Conv2d forward; ConvTranspose2d input-gradient
//   Dg : x  = conv_transpose(y, w)       Conv2d input-gradient; ConvTranspose2d forward
//   Wg : dw = sum_pixels y (x) patch(x)  weight gradient of either
//
// Each one's two partial derivatives are the other two with arguments permuted
// (SURVEY.md appendix C), so these three close the gradient-penalty double backward.
// Replaces, for the hot path, the aten ops behind torch.nn.Conv2d / ConvTranspose2d at
// reference core/models/standard_networks.py:20-24,36-43,60-73,80-87.
#include <cstdio>
#include <type_traits>

#include "gz_igemm.h"
#include "gz_reduce.h"
#include "../../include/gz_ops.h"

namespace gz {

using Cfg128x128 = TileCfg<2, 2, 2, 2>;
using Cfg128x64 = TileCfg<2, 2, 2, 1>;
using Cfg128x32 = TileCfg<4, 1, 1, 1>;
using Cfg64x64 = TileCfg<2, 2, 1, 1>;

// round 3 (gz_igemm.h, igemm2): one wavefront per SIMD, 128x128 / 128x64 accumulators per wavefront
using Cfg256x256 = TileCfg2<2, 2, 4, 1>;
using Cfg256x128 = TileCfg2<2, 2, 2, 2>;
using Cfg512x64 = TileCfg2<4, 1, 2, 2>;       // 64 output channels in all (D.block1-size input gradients)
using Cfg128x256 = TileCfg2<1, 4, 2, 2>;      // weight gradient with 128 output channels
// round 4: 256 x 64 (wavefronts 2 x 2 of 128 x 32, 64 accumulator registers): twice the tiles of 256x128 for launches
// that would otherwise put one workgroup on a CU or split their reduction; three workgroups per CU (37-42 KB of LDS)
using Cfg256x64 = TileCfg2<2, 2, 1, 3>;

enum TileId { T128x128 = 0, T128x64 = 1, T128x32 = 2, T64x64 = 3, T256x256 = 4, T256x128 = 5, T512x64 = 6, T128x256 = 7,
              T256x64 = 8,
              T256x128P = 9 };   // (a label only: ConvDg5A2's 256 pixels x (2 column phases x 64 channels), gz_conv2d_tile)
static bool is_tile2(TileId t) { return t >= T256x256; }

// Which launches take the igemm2 skeleton: its workgroup is a whole CU's worth of matrix pipes (one wavefront per
// SIMD), so 256 tiles already fill the chip and anything from there up runs at the loop's rate; fewer would leave
// CUs idle (those launches keep igemm_kernel + split-K).  256x128 tiles keep two workgroups per CU, whose prologues /
// epilogues overlap each other's main loops: preferred unless that halves a long reduction's operand reuse for nothing.
static TileId pick_tile2(long long M, long long N, int ny, int kdim) {
    const bool off = knobs().no_igemm2;
    if (!off && N > 32 && N <= 64 && kdim >= 256) {             // one 64-wide column of 512-pixel tiles
        const long long t = ((M + 511) / 512) * ny;
        if (t >= 2 * cus()) return T512x64;
        // (round 4: fewer than that -- D.block1's input gradient in a bs-128 generator step, 256 tiles of 512x64 -- go to
        // 256x64 tiles when those give every CU a workgroup)
        if (!knobs().no_tile64 && knobs().igemm2_tile == 0 && N == 64 && ((M + 255) / 256) * ny >= cus()) return T256x64;
        return T64x64;
    }
    if (off || N < 128 || kdim < 256) return T64x64;            // "not applicable"
    const long long t128 = ((M + 255) / 256) * ((N + 127) / 128) * ny;
    const long long t256 = ((M + 255) / 256) * ((N + 255) / 256) * ny;
    const int force = knobs().igemm2_tile;
    const int cu = cus();
    // round 4: 256x64 tiles (three workgroups per CU) where 256x128 would put fewer than two workgroups on a CU and
    // 256x64 gives every CU at least one: bs 128, D.block1 forward 95 -> 115 TFLOP/s, D.block2's input gradient 92 ->
    // 111; bs 256 (the stacked discriminator pass of bs 128): D.block2 forward 114 -> 124, D.block3's input gradient
    // 107 -> 119; every launch with >= 512 tiles of 256x128 keeps them (tools/conv_bench2.py, GZ_NO_TILE64=1)
    if (force == 256 && N >= 256 && t256 >= cu) return T256x256;
    if (force == 128 && t128 >= cu) return T256x128;
    if (t128 >= 2 * cu) return T256x128;
    if (N >= 256 && t256 >= cu) return T256x256;        // (round 3's choice where it applies: kept)
    if (t128 >= cu) return T256x128;
    // round 4: 256x64 tiles (three workgroups per CU) where 256x128 tiles would not give every CU a workgroup and the
    // launch would split its reduction instead: bs 128, D.block1 forward 95 -> 115 TFLOP/s, D.block2's input gradient
    // 92 -> 111; bs 256 (the stacked discriminator pass of bs 128): D.block2 forward 114 -> 124, D.block3's input
    // gradient 107 -> 119.  Launches with 256-511 tiles of 256x128 keep them (bs 512: 256x64 measured 2-5 % slower
    // there).  tools/conv_bench2.py, GZ_NO_TILE64=1.
    if (!knobs().no_tile64 && force == 0 && (N & 63) == 0) {
        const long long t64 = ((M + 255) / 256) * ((N + 63) / 64) * ny;
        if (t64 >= cu) return T256x64;
    }
    return T64x64;
}

static int forced_tile() { return knobs().tile; }

// Tile choice.  Measured on the DCGAN layers at bs 512 (tools/conv_bench.py, round 2, 4 / 6 / 8 co-resident
// workgroups per CU for the three shapes): a launch that fills the chip runs at ~125 (128x128), ~112 (128x64) and
// ~108 TFLOP/s (64x64); one that leaves workgroup slots empty loses in proportion (D.block3's dgrad: 512 tiles of
// 128x128 on 1024 slots -> the 2048 64x64 tiles win), and between one and two rounds part of the second round is
// exposed.  Score = shape efficiency x fill and take the best; narrow N gets narrow tiles.
static TileId pick_tile(long long M, long long N, int ny, int kdim = 0) {
    int f = forced_tile();
    if (f >= 0 && f <= 3) {
        if (!(f == T128x128 && N <= 64) ) return (TileId)f;
    }
    if (N <= 32) return T128x32;
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * ny; };
    // A launch with >= 34 GFLOP of work (65536 tile-chunks of 128x128x16) always has enough of it for ~1024 workgroups of
    // the 128x128 shape with >= 64 chunks each once the reduction is split (plan_split), and that beats the smaller
    // shapes whatever the tile count says: bs 128, G.block2's input gradient 0.370 -> 0.305 ms, G.block3's 0.309 ->
    // 0.295, their forward 0.337 -> 0.324 / 0.299 -> 0.293; the 8.6 GFLOP discriminator layers of that batch lose 10 %
    // the same way (16 chunks per workgroup: prologue, slab write and finish dominate) and stay with the score.
    const bool no_big = knobs().no_big_split;
    if (!no_big && kdim > 0 && N > 64 && tiles(128, 128) * ((kdim + BK - 1) / BK) >= 65536) return T128x128;
    if (knobs().min_wgs > 0) {            // round-1 rule, kept for experiments
        const long long want = knobs().min_wgs;
        if (N <= 64) return tiles(128, 64) >= want ? T128x64 : T64x64;
        if (tiles(128, 128) >= want) return T128x128;
        if (tiles(128, 64) >= want) return T128x64;
        return T64x64;
    }
    auto score = [&](int bm, int bn, double eff, int per_cu) {
        const double rounds = (double)tiles(bm, bn) / ((double)cus() * per_cu);
        double fill = 1.0;
        if (rounds <= 1.0) fill = rounds;
        else if (rounds < 2.0) fill = 0.5 + 0.5 * rounds / 2.0;      // 1 < rounds < 2: half of the tail is hidden
        return eff * fill;
    };
    const double s64 = score(64, 64, 0.86, 8), s128x64 = score(128, 64, 0.90, 6);
    if (N <= 64) return s128x64 >= s64 ? T128x64 : T64x64;
    const double s128 = score(128, 128, 1.0, 4);
    if (s128 >= s128x64 && s128 >= s64) return T128x128;
    return s128x64 >= s64 ? T128x64 : T64x64;
}

// Split-K for F / Dg / GEMM launches whose output has too few tiles to fill 256 CUs (deep 4x4 / 8x8 feature
// maps at small batch, the nn.Linear heads: M = batch rows, K = 8192).  Such a launch runs one workgroup per
// CU or less and is bound by the per-chunk load -> LDS -> barrier latency, which only other resident workgroups
// can hide.  Keep the tile pick_tile chose and cut the reduction so that ~1024 workgroups are in flight, each
// with >= 8 chunks (128 reduction steps); the raw partial tiles go to workspace slabs, splitk_finish_kernel
// sums them in a fixed order and applies the epilogue.
struct SplitPlan {
    TileId tile;
    int splits;
};

static long long tile_count(TileId t, long long M, long long N, int ny) {
    if (t == T256x256 || t == T256x128) return ((M + 255) / 256) * ((N + (t == T256x256 ? 255 : 127)) / (t == T256x256 ? 256 : 128)) * ny;
    if (t == T512x64) return ((M + 511) / 512) * ((N + 63) / 64) * ny;
    if (t == T256x64) return ((M + 255) / 256) * ((N + 63) / 64) * ny;
    const int bm = t == T64x64 ? 64 : 128;
    const int bn = t == T128x128 ? 128 : (t == T128x32 ? 32 : 64);
    return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * ny;
}

static SplitPlan plan_split(long long M, long long N, int Kdim, int ny, TileId normal) {
    SplitPlan none{normal, 1};
    const int target = knobs().split_target * cus() / 256, below = knobs().split_below * cus() / 256;
    if (knobs().no_splitk) return none;
    const int chunks = (Kdim + BK - 1) / BK;
    const long long tiles = tile_count(normal, M, N, ny);
    if (chunks < 16 || tiles >= below) return none;
    long long want = (target + tiles - 1) / tiles, cap = chunks / 8;
    int splits = (int)(want < cap ? want : cap);
    if (splits < 2) return none;
    return SplitPlan{normal, splits};
}

static size_t split_bytes(const SplitPlan& sp, long long M, long long N, int Kdim, int ny) {
    if (sp.splits <= 1) return 0;
    return (size_t)split_nz(Kdim, sp.splits) * ny * M * N * 4;
}

// ---------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int round4(int v) { return (v + 3) & ~3; }

// The pack kernels exist twice: as their own launches (bx / by / gx = blockIdx / gridDim) and as the bodies of
// pack_multi_kernel, which re-packs every weight of a network in one launch (a job table maps a block to its tensor).
// dst[c][ld] (c < COLS) = src[r][c] transposed: dst[c*ld + r] = src[r*COLS + c]; zero for r in [R, ld)
__device__ __forceinline__ void transpose_pad_body(const float* __restrict__ src, float* __restrict__ dst, int R,
                                                   int COLS, int ld, int bx, int by, float scale = 1.f) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int c0 = bx * 32, r0 = by * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < R && c < COLS) ? src[(long long)r * COLS + c] * scale : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < COLS && r < ld) dst[(long long)c * ld + r] = tile[tx][ty + 8 * i];
    }
}

__global__ __launch_bounds__(256) void transpose_pad_kernel(const float* __restrict__ src,
                                                            float* __restrict__ dst, int R, int COLS, int ld) {
    transpose_pad_body(src, dst, R, COLS, ld, blockIdx.x, blockIdx.y);
}

// Tap-major reduction order (gz_igemm.h: ConvFwdALoaderTap / ConvDgALoaderTap) is used when the tap count does not
// divide a chunk (3x3, 5x5) and there are enough channels to fill the BK-wide channel blocks.
static bool fwd_tap_major(int C, int KH, int KW) {
    return !knobs().no_tapmajor && (BK % (KH * KW) != 0) && C >= BK;
}

static bool dgrad_tap_major(int K, int KH, int KW, int S) {
    const bool off = knobs().no_tapmajor;
    const int taps = ((KH + S - 1) / S) * ((KW + S - 1) / S);
    const bool fixed = (KH % S == 0) && (KW % S == 0) && (BK % taps == 0);
    return !off && !fixed && K >= BK;
}

// wp[(tap, c)][ld] = w[ko][c][tap], c padded to a multiple of BK with zero rows
__device__ __forceinline__ void pack_fwd_tap_body(const float* __restrict__ w, float* __restrict__ wp, int K, int C,
                                                  int taps, int cpad, int ld, int bx, int gx, float scale = 1.f) {
    const long long total = (long long)taps * cpad * ld;
    for (long long i = (long long)bx * 256 + threadIdx.x; i < total; i += (long long)gx * 256) {
        int ko = (int)(i % ld);
        long long row = i / ld;
        int c = (int)(row % cpad), tap = (int)(row / cpad);
        wp[i] = (ko < K && c < C) ? w[((long long)ko * C + c) * taps + tap] * scale : 0.f;
    }
}

__global__ __launch_bounds__(256) void pack_fwd_tap_kernel(const float* __restrict__ w, float* __restrict__ wp, int K,
                                                           int C, int taps, int cpad, int ld) {
    pack_fwd_tap_body(w, wp, K, C, taps, cpad, ld, blockIdx.x, gridDim.x);
}

// wp[phase][(tap, ko)][ldc] = w[ko][c][ky][kx] over the phase's own ny x nx taps (tap = ty * nx + tx), ko padded
// to a multiple of BK; the unused tail of the phase's fixed-size TY*TX*kpad-row region is never read
__device__ __forceinline__ void pack_dgrad_tap_body(const float* __restrict__ w, float* __restrict__ wp, int K, int C,
                                                    int KH, int KW, int S, int P, int TY, int TX, int kpad, int ldc,
                                                    int bx, int phase, int gx, float scale = 1.f) {
    const int py = phase / S, px = phase % S;
    const int ry = (py + P) % S, rx = (px + P) % S;
    const int ny = dg_taps(KH, S, P, py), nx = dg_taps(KW, S, P, px);
    float* dst = wp + (long long)phase * TY * TX * kpad * ldc;
    const long long total = (long long)ny * nx * kpad * ldc;
    for (long long i = (long long)bx * 256 + threadIdx.x; i < total; i += (long long)gx * 256) {
        int c = (int)(i % ldc);
        long long row = i / ldc;
        int ko = (int)(row % kpad), tap = (int)(row / kpad);
        int ky = ry + S * (tap / nx), kx = rx + S * (tap % nx);
        dst[i] = (ko < K && c < C) ? w[(((long long)ko * C + c) * KH + ky) * KW + kx] * scale : 0.f;
    }
}

__global__ __launch_bounds__(256) void pack_dgrad_tap_kernel(const float* __restrict__ w, float* __restrict__ wp, int K,
                                                             int C, int KH, int KW, int S, int P, int TY, int TX,
                                                             int kpad, int ldc) {
    pack_dgrad_tap_body(w, wp, K, C, KH, KW, S, P, TY, TX, kpad, ldc, blockIdx.x, blockIdx.y, gridDim.x);
}

// dgrad pack: wp[phase][(ko, ty, tx)][ldc] = w[ko][c][ky][kx], ky = ((py+P)%S) + S*ty.  A phase only has the taps
// whose ky < KH (kx < KW): ny(py) * nx(px) of them (dg_taps); its rows are packed tightly and the rest of the
// phase's fixed-size K*TY*TX-row region is zero.  (k5 s2: 9/6/6/4 taps instead of 4 x 9.)
__device__ __forceinline__ void pack_dgrad_body(const float* __restrict__ w, float* __restrict__ wp, int K, int C,
                                                int KH, int KW, int S, int P, int TY, int TX, int ldc, int ko,
                                                int phase, float scale = 1.f) {
    const int py = phase / S, px = phase % S;
    const int ry = (py + P) % S, rx = (px + P) % S;
    const int ny = dg_taps(KH, S, P, py), nx = dg_taps(KW, S, P, px);
    const int taps = ny * nx, pad = TY * TX - taps;
    float* dst = wp + (long long)phase * K * TY * TX * ldc;
    for (int i = threadIdx.x; i < taps * ldc; i += 256) {
        int tap = i / ldc, c = i - tap * ldc;
        int ky = ry + S * (tap / nx), kx = rx + S * (tap % nx);
        dst[((long long)ko * taps + tap) * ldc + c] =
            c < C ? w[(((long long)ko * C + c) * KH + ky) * KW + kx] * scale : 0.f;
    }
    for (int i = threadIdx.x; i < pad * ldc; i += 256)
        dst[((long long)K * taps + (long long)ko * pad) * ldc + i] = 0.f;
}

// The k4 s2 p1 case of the same image (every DCGAN layer), one workgroup per ko and ALL four phases (round 5): the body
// above reads w[ko][c][ky][kx] along c -- a 64-byte stride, one 64-byte segment per lane and load -- once per phase.
// Here the C x 16 values of the ko are read once, contiguously (16-byte loads), turned in LDS 64 channels at a time,
// and leave as the sixteen (phase, tap) rows with 16-byte stores.  Same bytes out, bit for bit.
__device__ __forceinline__ void pack_dgrad_k4_body(const float* __restrict__ w, float* __restrict__ wp, int K, int C,
                                                   int ldc, int ko, float scale = 1.f) {
    __shared__ float tile[64][17];
    const int t = threadIdx.x;
    const int r = t >> 4, cs = (t & 15) * 4;                  // output row (ky, kx) and channel quad of this thread
    const int ky = r >> 2, kx = r & 3;
    const int phase = ((ky + 1) & 1) * 2 + ((kx + 1) & 1), tap = (ky >> 1) * 2 + (kx >> 1);
    float* dst = wp + (((long long)phase * K + ko) * 4 + tap) * ldc;
    const float* src = w + (long long)ko * C * 16;
    for (int c0 = 0; c0 < ldc; c0 += 64) {
        const int c = c0 + (t >> 2), e = (t & 3) * 4;         // this thread's 4 consecutive (ky, kx) values of channel c
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < C) v = *reinterpret_cast<const f32x4*>(src + (long long)c * 16 + e);
        __syncthreads();
        tile[t >> 2][e + 0] = v.x * scale;
        tile[t >> 2][e + 1] = v.y * scale;
        tile[t >> 2][e + 2] = v.z * scale;
        tile[t >> 2][e + 3] = v.w * scale;
        __syncthreads();
        if (c0 + cs < ldc) {
            f32x4 o = {tile[cs][r], tile[cs + 1][r], tile[cs + 2][r], tile[cs + 3][r]};
            *reinterpret_cast<f32x4*>(dst + c0 + cs) = o;
        }
    }
}

__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                         int K, int C, int KH, int KW, int S, int P, int TY,
                                                         int TX, int ldc) {
    pack_dgrad_body(w, wp, K, C, KH, KW, S, P, TY, TX, ldc, blockIdx.x, blockIdx.y);
}

// One launch for many packs: jobs[j] describes one (weight, packed image) pair and the grid it would have had as a
// launch of its own; block b belongs to the job with block0 <= b < block0 + gx * gy.
struct PackJob {
    const float* w;
    float* wp;
    int kind;                 // 0 transpose_pad (forward), 1 forward tap-major, 2 dgrad, 3 dgrad tap-major
    int K, C, KH, KW, S, P;
    int gx, gy, block0;
};

__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJob* __restrict__ jobs, int njobs) {
    const int b = blockIdx.x;
    int j = 0;
    while (j + 1 < njobs && jobs[j + 1].block0 <= b) ++j;
    const PackJob jb = jobs[j];
    const int l = b - jb.block0, bx = l % jb.gx, by = l / jb.gx;
    const int TY = (jb.KH + jb.S - 1) / jb.S, TX = (jb.KW + jb.S - 1) / jb.S;
    switch (jb.kind) {
        case 0: transpose_pad_body(jb.w, jb.wp, jb.K, jb.C * jb.KH * jb.KW, round4(jb.K), bx, by); break;
        case 1: pack_fwd_tap_body(jb.w, jb.wp, jb.K, jb.C, jb.KH * jb.KW, round_bk(jb.C), round4(jb.K), bx, jb.gx); break;
        case 2: pack_dgrad_body(jb.w, jb.wp, jb.K, jb.C, jb.KH, jb.KW, jb.S, jb.P, TY, TX, round4(jb.C), bx, by); break;
        case 5: pack_dgrad_k4_body(jb.w, jb.wp, jb.K, jb.C, round4(jb.C), bx); break;
        default:
            pack_dgrad_tap_body(jb.w, jb.wp, jb.K, jb.C, jb.KH, jb.KW, jb.S, jb.P, TY, TX, round_bk(jb.K), round4(jb.C), bx,
                                by, jb.gx);
    }
}

// The same bodies with the job table passed BY VALUE and an optional scale 1 / sigma[0] read from the device:
// spectral normalisation's w = weight_orig / sigma is a fresh tensor at every discriminator call, so its images cannot
// live in the persistent table above; one launch writes w itself (kind 4) and both packed images of every
// spectral-normalised layer (functional.spectral_normalize_multi) -- they were a div_scalar and two pack launches per
// layer and call.
constexpr int PACK_TABLE_MAX = 12;
struct PackTable {
    int njobs, pad;
    PackJob jobs[PACK_TABLE_MAX];
    const float* sigma[PACK_TABLE_MAX];
};

__global__ __launch_bounds__(256) void pack_table_kernel(PackTable t) {
    const int b = blockIdx.x;
    int j = 0;
    while (j + 1 < t.njobs && t.jobs[j + 1].block0 <= b) ++j;
    const PackJob& jb = t.jobs[j];
    const float scale = t.sigma[j] ? 1.f / t.sigma[j][0] : 1.f;
    const int l = b - jb.block0, bx = l % jb.gx, by = l / jb.gx;
    const int TY = (jb.KH + jb.S - 1) / jb.S, TX = (jb.KW + jb.S - 1) / jb.S;
    switch (jb.kind) {
        case 0: transpose_pad_body(jb.w, jb.wp, jb.K, jb.C * jb.KH * jb.KW, round4(jb.K), bx, by, scale); break;
        case 1:
            pack_fwd_tap_body(jb.w, jb.wp, jb.K, jb.C, jb.KH * jb.KW, round_bk(jb.C), round4(jb.K), bx, jb.gx, scale);
            break;
        case 2:
            pack_dgrad_body(jb.w, jb.wp, jb.K, jb.C, jb.KH, jb.KW, jb.S, jb.P, TY, TX, round4(jb.C), bx, by, scale);
            break;
        case 3:
            pack_dgrad_tap_body(jb.w, jb.wp, jb.K, jb.C, jb.KH, jb.KW, jb.S, jb.P, TY, TX, round_bk(jb.K), round4(jb.C), bx,
                                by, jb.gx, scale);
            break;
        case 5: pack_dgrad_k4_body(jb.w, jb.wp, jb.K, jb.C, round4(jb.C), bx, scale); break;
        default: {      // 4: wp = w * scale, same layout
            const long long total = (long long)jb.K * jb.C * jb.KH * jb.KW;
            for (long long i = (long long)bx * 256 + threadIdx.x; i < total; i += (long long)jb.gx * 256)
                jb.wp[i] = jb.w[i] * scale;
        }
    }
}

// out[i] = sum_s slab[s][i].  Small weight tensors reach here with hundreds of slabs (split-K over a 1M-long
// reduction), so the slabs are spread over the 16 wavefronts of a workgroup (64 outputs per workgroup, fixed
// summation order) instead of being walked by one thread.
__global__ __launch_bounds__(256) void reduce_few_slabs_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                               int S, long long count) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += slab[(long long)s * count + i];
    out[i] = acc;
}

constexpr int RS_WAVES = 16;
__global__ __launch_bounds__(64 * RS_WAVES) void reduce_slabs_kernel(const float* __restrict__ slab,
                                                                     float* __restrict__ out, int S,
                                                                     long long count, long long stride,
                                                                     float* __restrict__ out2, long long split) {
    __shared__ float part[RS_WAVES][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < count) {
        const float* p = slab + i;
        int s = wave;
        for (; s + 3 * RS_WAVES < S; s += 4 * RS_WAVES) {        // `stride` = slab row length (>= count)
            a0 += p[(long long)s * stride];
            a1 += p[(long long)(s + RS_WAVES) * stride];
            a2 += p[(long long)(s + 2 * RS_WAVES) * stride];
            a3 += p[(long long)(s + 3 * RS_WAVES) * stride];
        }
        for (; s < S; s += RS_WAVES) a0 += p[(long long)s * stride];
    }
    part[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && i < count) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) acc += part[w][lane];
        if (out2 && i >= split) out2[i - split] = acc;     // tail of the slab row: the fused bias gradient
        else out[i] = acc;
    }
}


// Round 4: ONE launch sums the slabs of MANY weight gradients (the pack_multi idea applied to the other end of the
// step).  A split weight-gradient launch may leave its slabs unreduced (gz_conv2d_wgrad_partial); at the end of a
// backward pass -- or when a gradient bucket of the data-parallel exchange is complete -- gz_reduce_multi adds, per
// parameter, the slabs of every launch that contributed (a discriminator used on a real and a fake batch has two
// sources) and either writes or ACCUMULATES into the gradient (beta = 1: p.grad already holds earlier contributions;
// under data parallelism p.grad is a view of the flat exchange buffer).  Replaces, per DCGAN pair, 12 reduce launches
// + 11 framework `add_` launches of gradient accumulation by 2-4 launches.  The table travels as a kernel argument
// (no staging copy).  Summation order is fixed: wavefront w of a workgroup takes slabs w, w+4, ... of source 0, then
// of source 1, ...; the four partial sums meet in LDS in wavefront order.
constexpr int REDUCE_MAX_JOBS = 24;
struct ReduceJob {
    float* out;
    long long count;           // floats, a multiple of 4
    int beta, nsrc, block0, pad;
    ReduceSrc src[REDUCE_MAX_SRC];
};
struct ReduceTable {
    int njobs, pad;
    ReduceJob jobs[REDUCE_MAX_JOBS];
};

__global__ __launch_bounds__(256) void reduce_multi_kernel(ReduceTable t) {
    __shared__ f32x4 part[3][64];
    const int b = blockIdx.x;
    int j = 0;
    while (j + 1 < t.njobs && t.jobs[j + 1].block0 <= b) ++j;
    const ReduceJob& jb = t.jobs[j];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long i = ((long long)(b - jb.block0) * 64 + lane) * 4;
    const bool live = i < jb.count;
    const f32x4 r0 = reduce_sources(jb.src, jb.nsrc, i, live, lane, wave, part);
    if (wave == 0 && live) {
        f32x4 r = r0;
        f32x4* o = reinterpret_cast<f32x4*>(jb.out + i);
        if (jb.beta) r += *o;
        *o = r;
    }
}

// set by gz_conv2d_wgrad_partial around its dispatch: the split launch reports its slabs instead of reducing them
struct WgDefer {
    int nz;
    long long stride;
};
static thread_local WgDefer* tl_wg_defer = nullptr;
static bool defer_reduce(int nz, long long stride) {
    if (!tl_wg_defer) return false;
    tl_wg_defer->nz = nz;
    tl_wg_defer->stride = stride;
    return true;
}

template <int KH, int KW, int S, int P>
struct Geo {
    static constexpr int kh = KH, kw = KW, s = S, p = P;
};

static bool shape_ok(const ConvShape& s, int KH, int KW, int S, int P) {
    if (s.N <= 0 || s.C <= 0 || s.K <= 0 || s.H <= 0 || s.W <= 0) return false;
    if (s.OH != (s.H + 2 * P - KH) / S + 1 || s.OW != (s.W + 2 * P - KW) / S + 1) return false;
    return true;
}

static bool too_large(long long elems) { return elems * 4 >= (1ll << 31); }

// ---------------------------------------------------------------------------
// F
// ---------------------------------------------------------------------------
template <class G, class Cfg>
static int run_fwd(const float* x, const float* wp, const float* bias, float* y, const ConvShape& s, int act,
                   float slope, hipStream_t st, int splits = 1, float* slab = nullptr, float* stats = nullptr) {
#ifndef GZ_NO_K4V
    using AL = std::conditional_t<G::kh == 4 && G::kw == 4, ConvFwdALoaderK4V<Cfg::BM, G::s, G::p>,
                                  ConvFwdALoader<Cfg::BM, G::kh, G::kw, G::s, G::p>>;
#else
    using AL = ConvFwdALoader<Cfg::BM, G::kh, G::kw, G::s, G::p>;
#endif
    using BL = MContigLoader4<Cfg::BN>;
    typename AL::Params pa{x, s, make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW)};
    int M = s.N * s.OH * s.OW;
    EpiNCHWB::Params pe{y, M, s.K, s.OH * s.OW, make_fastdiv(s.OH * s.OW), bias, act, slope,
                       reinterpret_cast<f32x2*>(stats)};
    if constexpr (BK % (G::kh * G::kw) != 0) {
        if (fwd_tap_major(s.C, G::kh, G::kw)) {
            using ALT = ConvFwdALoaderTap<Cfg::BM, G::kh, G::kw, G::s, G::p>;
            int Kt = G::kh * G::kw * round_bk(s.C);
            typename BL::Params pbt{wp, Kt, round4(s.K), round4(s.K), 0};
            return launch_igemm<Cfg, ALT, BL, EpiNCHWB>(pa, pbt, pe, M, s.K, Kt, 1, splits, st, slab);
        }
    }
    int Kg = s.C * G::kh * G::kw;
    typename BL::Params pb{wp, Kg, round4(s.K), round4(s.K), 0};
    if constexpr (G::kh == 4 && G::kw == 4 && G::s == 2 && G::p == 1) {
        const bool no_row4 = knobs().no_row4;          // experiment: the K4V loader
        // OW >= 16 only: with shorter rows the stride-2 fragment reads of the lanes of a half-wave fall on 2*OW / 2
        // banks (OW = 4: an 8-way conflict; measured 113 -> 108 TFLOP/s on D.block3, 122 -> 117 on G.block2's
        // backward), where K4V's im2col image stays conflict-free; at OW = 16 / 32 it is +5 % (D.block1) or neutral
        if (!no_row4 && s.W == 2 * s.OW && s.H == 2 * s.OH && s.OW >= 16 && s.OW <= Cfg::BM && Cfg::BM % s.OW == 0 &&
            (((uintptr_t)x) & 15) == 0) {
            using AR = ConvFwdALoaderRow4<Cfg::BM>;
            return launch_igemm<Cfg, AR, BL, EpiNCHWB>(pa, pb, pe, M, s.K, Kg, 1, splits, st, slab);
        }
    }
    return launch_igemm<Cfg, AL, BL, EpiNCHWB>(pa, pb, pe, M, s.K, Kg, 1, splits, st, slab);
}

// k4 s2 p1 forward convolution on the igemm2 skeleton (raw input rows by LDS-DMA, taps on the fragment read)
template <class Cfg, int OWC>
static int run_fwd2(const float* x, const float* wp, const float* bias, float* y, const ConvShape& s, int act,
                    float slope, hipStream_t st, int splits, float* slab, float* stats) {
    using AL = ConvFwdA2<Cfg::BM, OWC>;
    using BL = MContigB2<Cfg::BN>;
    typename AL::Params pa{x, s, make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW)};
    const int M = s.N * s.OH * s.OW;
    EpiNCHWB::Params pe{y, M, s.K, s.OH * s.OW, make_fastdiv(s.OH * s.OW), bias, act, slope,
                        reinterpret_cast<f32x2*>(stats)};
    const int Kg = s.C * 16;
    typename BL::Params pb{wp, Kg, round4(s.K), round4(s.K), 0};
    return launch_igemm2<Cfg, AL, BL, EpiNCHWB>(pa, pb, pe, M, s.K, Kg, 1, splits, st, slab);
}

// any other geometry whose reduction is tap-major (5x5 s2 p2, 3x3 s1 p1, 1x1 ...): gather loader, 4-byte LDS-DMA
template <class G, class Cfg>
static int run_fwdtap2_impl(const float* x, const float* wp, const float* bias, float* y, const ConvShape& s, int act,
                            float slope, hipStream_t st, int splits, float* slab, float* stats) {
    using AL = ConvTapA2<Cfg::BM, G::kh, G::kw, G::s, G::p>;
    using BL = MContigB2<Cfg::BN>;
    typename AL::Params pa{x, s, make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW)};
    const int M = s.N * s.OH * s.OW;
    EpiNCHWB::Params pe{y, M, s.K, s.OH * s.OW, make_fastdiv(s.OH * s.OW), bias, act, slope,
                        reinterpret_cast<f32x2*>(stats)};
    const int Kt = G::kh * G::kw * round_bk(s.C);
    // (1x1: the plain [C][K] weight image IS the tap-major one; it has no padding rows, the descriptor ends at row C)
    typename BL::Params pb{wp, G::kh * G::kw == 1 ? s.C : Kt, round4(s.K), round4(s.K), 0};
    if constexpr (G::kh * G::kw == 1 && G::s == 1 && G::p == 0) {
        if (!knobs().no_plane_a && ((s.H * s.W) & 3) == 0 && !((uintptr_t)x & 15)) {      // plain GEMM: 16-byte pieces
            using AP = PlaneA2<Cfg::BM>;
            typename AP::Params pp{x, s.C, s.H * s.W, M, make_fastdiv(s.H * s.W)};
            return launch_igemm2<Cfg, AP, BL, EpiNCHWB>(pp, pb, pe, M, s.K, Kt, 1, splits, st, slab);
        }
    }
    return launch_igemm2<Cfg, AL, BL, EpiNCHWB>(pa, pb, pe, M, s.K, Kt, 1, splits, st, slab);
}

// k4 s2 p1 has loaders of its own (ConvFwdA2 / ConvDgA2) and the planners never pair it with the gather loaders
// (fwdtap2_plan / dgradtap2_plan return "not applicable" for it): not instantiating them for that geometry takes eight
// never-launched kernels out of the library (round 5, tools/kernel_reach.sh).
template <class G>
constexpr bool has_own_igemm2_loaders() { return G::kh == 4 && G::kw == 4 && G::s == 2 && G::p == 1; }

template <class G, class Cfg>
static int run_fwdtap2(const float* x, const float* wp, const float* bias, float* y, const ConvShape& s, int act,
                       float slope, hipStream_t st, int splits, float* slab, float* stats) {
    if constexpr (has_own_igemm2_loaders<G>()) return GZ_ERR_UNSUPPORTED;
    else return run_fwdtap2_impl<G, Cfg>(x, wp, bias, y, s, act, slope, st, splits, slab, stats);
}

template <class Cfg>
static int run_fwd2_ow(const float* x, const float* wp, const float* bias, float* y, const ConvShape& s, int act,
                       float slope, hipStream_t st, int splits = 1, float* slab = nullptr, float* stats = nullptr) {
    switch (s.OW) {
        case 4: return run_fwd2<Cfg, 4>(x, wp, bias, y, s, act, slope, st, splits, slab, stats);
        case 8: return run_fwd2<Cfg, 8>(x, wp, bias, y, s, act, slope, st, splits, slab, stats);
        case 16: return run_fwd2<Cfg, 16>(x, wp, bias, y, s, act, slope, st, splits, slab, stats);
        case 32: return run_fwd2<Cfg, 32>(x, wp, bias, y, s, act, slope, st, splits, slab, stats);
        case 64: return run_fwd2<Cfg, 64>(x, wp, bias, y, s, act, slope, st, splits, slab, stats);
        default: return GZ_ERR_UNSUPPORTED;
    }
}

template <class G>
static bool fwd2_ok(const ConvShape& s) {
    return G::kh == 4 && G::kw == 4 && G::s == 2 && G::p == 1 && s.H == 2 * s.OH && s.W == 2 * s.OW &&
           (s.OW == 4 || s.OW == 8 || s.OW == 16 || s.OW == 32 || s.OW == 64);
}

template <class G>
static int fwd_kdim(const ConvShape& s) {
    return G::kh * G::kw * (fwd_tap_major(s.C, G::kh, G::kw) ? round_bk(s.C) : s.C);
}

// Forward tile.  One correction to the score: k4 s2 p1 with output rows shorter than 16 pixels runs on the K4V gather
// loader, whose 64x64 form is its weak spot (bs 128, G.block2's input gradient: 93 vs 107 TFLOP/s; bs 512, D.block2
// forward: 110 vs 113) -- with >= 512 tiles of 128x128 the chip is half full and that kernel still wins.
static TileId pick_tile_fwd(long long M, int K, int OW, int kh, int kw, int stride, int kdim) {
    TileId t = pick_tile(M, K, 1, kdim);
    if (forced_tile() < 0 && kh == 4 && kw == 4 && stride == 2 && OW < 16 && t == T64x64 && K > 64 &&
        tile_count(T128x128, M, K, 1) >= 2 * cus())
        t = T128x128;
    return t;
}

// igemm2 plan of a forward launch: tile and reduction splits (few M tiles: the reduction is cut so that >= 256
// workgroups exist, each with >= 32 chunks)
// tap-major geometries on the igemm2 skeleton (ConvTapA2): >= 128 output channels, and enough tiles x reduction
// splits to give every CU a workgroup, each with >= 32 chunks
template <class G>
static SplitPlan fwdtap2_plan(const ConvShape& s) {
    const bool off = knobs().no_igemm2 || knobs().no_igemm2_tap;
    if constexpr (BK % (G::kh * G::kw) == 0 && G::kh * G::kw != 1) return SplitPlan{T64x64, 1};
    if (off || !(G::kh * G::kw == 1 ? s.C >= BK : fwd_tap_major(s.C, G::kh, G::kw)) || s.K < 128 || (s.K & 3))
        return SplitPlan{T64x64, 1};
    const long long M = (long long)s.N * s.OH * s.OW;
    const long long tiles = ((M + 255) / 256) * ((s.K + 127) / 128);
    const int chunks = G::kh * G::kw * round_bk(s.C) / BK;
    const int cu = cus();
    if (tiles >= cu * 7 / 8) return SplitPlan{T256x128, 1};
    // split launches pay off from ~64 chunks per workgroup (HoloGAN EXT-128's blocks: 100 -> 115-118 TFLOP/s against
    // 80-94 on the 64x64 tiles; the 64x64-image blocks would get 33 chunks each and lose 5 %)
    if (tiles >= 8 && chunks >= 128) {
        int splits = (int)((cu + tiles - 1) / tiles);
        while (splits > 1 && chunks / splits < 64) --splits;
        if (splits > 1 && tiles * splits >= cu * 3 / 4) return SplitPlan{T256x128, splits};
    }
    // round 4: 256x64 tiles for the launches too small for the rule above (HoloGAN's 5x5 s2 p2 critic blocks at 64x64,
    // 6.7 GFLOP each, bs 64: 76 / 77 / 65 -> 92 / 92 / 76 TFLOP/s; 16, 32 or 48 chunks per piece measure the same)
    const int mc = knobs().tap64_min_chunks;
    if (mc > 0 && !knobs().no_tile64 && (s.K & 63) == 0 && chunks >= mc) {
        const long long t64 = ((M + 255) / 256) * ((s.K + 63) / 64);
        if (t64 >= cu * 7 / 8) return SplitPlan{T256x64, 1};
        int splits = (int)((cu + t64 - 1) / t64);
        while (splits > 1 && chunks / splits < mc) --splits;
        if (t64 >= 8 && t64 * splits >= cu * 3 / 4) return SplitPlan{T256x64, splits};
    }
    return SplitPlan{T64x64, 1};
}

template <class G>
static SplitPlan fwd2_plan(const ConvShape& s) {
    if (!fwd2_ok<G>(s)) return fwdtap2_plan<G>(s);
    const long long M = (long long)s.N * s.OH * s.OW;
    TileId t = s.K > 64 ? pick_tile2(M, s.K, 1, s.C * 16) : T64x64;
    if (t == T256x256 || t == T256x128 || t == T256x64) return SplitPlan{t, 1};
    const bool off = knobs().no_igemm2;
    if (off || s.K < 128) return SplitPlan{T64x64, 1};
    // split launches take the 256x64 tile when the channel count allows: twice the tiles, half the slabs
    const bool t64 = !knobs().no_tile64 && (s.K & 63) == 0;
    const long long tiles = ((M + 255) / 256) * (t64 ? (s.K + 63) / 64 : (s.K + 127) / 128);
    const int chunks = s.C;
    const int min_tiles = knobs().fwd2_min_tiles * (t64 ? 2 : 1);
    if (tiles >= min_tiles && chunks >= 64) {
        int splits = (int)((cus() + tiles - 1) / tiles);
        while (splits > 1 && chunks / splits < 32) --splits;
        if (splits > 1 && tiles * splits >= cus()) return SplitPlan{t64 ? T256x64 : T256x128, splits};
    }
    return SplitPlan{T64x64, 1};
}

template <class G>
static SplitPlan fwd_plan(const ConvShape& s) {
    long long M = (long long)s.N * s.OH * s.OW;
    {
        const SplitPlan p2 = fwd2_plan<G>(s);
        if (p2.tile == T256x256 || p2.tile == T256x128 || p2.tile == T256x64) return p2;
    }
    return plan_split(M, s.K, fwd_kdim<G>(s), 1, pick_tile_fwd(M, s.K, s.OW, G::kh, G::kw, G::s, fwd_kdim<G>(s)));
}

template <class G>
static size_t fwd_ws_bytes(const ConvShape& s) {
    return split_bytes(fwd_plan<G>(s), (long long)s.N * s.OH * s.OW, s.K, fwd_kdim<G>(s), 1);
}

template <class G>
static int dispatch_fwd(const float* x, const float* wp, const float* bias, float* y, const ConvShape& s,
                        int act, float slope, float* ws, size_t ws_bytes, hipStream_t st) {
    long long M = (long long)s.N * s.OH * s.OW;
    SplitPlan sp = fwd_plan<G>(s);
    if (sp.splits > 1 && (!ws || ws_bytes < fwd_ws_bytes<G>(s)))
        sp = SplitPlan{pick_tile_fwd(M, s.K, s.OW, G::kh, G::kw, G::s, 0), 1};
    float* slab = sp.splits > 1 ? ws : nullptr;
    if (is_tile2(sp.tile) && fwd2_ok<G>(s) && (((uintptr_t)x) & 15) != 0) {
        sp = SplitPlan{pick_tile_fwd(M, s.K, s.OW, G::kh, G::kw, G::s, 0), 1};      // unaligned tensor
        slab = nullptr;
    }
    switch (sp.tile) {
        case T256x256: return run_fwd2_ow<Cfg256x256>(x, wp, bias, y, s, act, slope, st, sp.splits, slab);
        case T256x64:
            if (!fwd2_ok<G>(s)) return run_fwdtap2<G, Cfg256x64>(x, wp, bias, y, s, act, slope, st, sp.splits, slab, nullptr);
            return run_fwd2_ow<Cfg256x64>(x, wp, bias, y, s, act, slope, st, sp.splits, slab);
        case T256x128:
            if (!fwd2_ok<G>(s)) return run_fwdtap2<G, Cfg256x128>(x, wp, bias, y, s, act, slope, st, sp.splits, slab, nullptr);
            return run_fwd2_ow<Cfg256x128>(x, wp, bias, y, s, act, slope, st, sp.splits, slab);
        case T128x128: return run_fwd<G, Cfg128x128>(x, wp, bias, y, s, act, slope, st, sp.splits, slab);
        case T128x64: return run_fwd<G, Cfg128x64>(x, wp, bias, y, s, act, slope, st, sp.splits, slab);
        case T128x32: return run_fwd<G, Cfg128x32>(x, wp, bias, y, s, act, slope, st, sp.splits, slab);
        default: return run_fwd<G, Cfg64x64>(x, wp, bias, y, s, act, slope, st, sp.splits, slab);
    }
}

// ---------------------------------------------------------------------------
// F with run-time geometry (evaluation path: InceptionV3).  Always tap-major, channels padded to a chunk.
// ---------------------------------------------------------------------------
struct AnyGeom {
    int KH, KW, SH, SW, PH, PW;
};

template <class Cfg>
static int run_fwd_any(const float* x, const float* wp, const float* bias, float* y, const ConvShape& s,
                       const AnyGeom& g, int act, float slope, hipStream_t st, int splits, float* slab) {
    using AL = ConvFwdALoaderTapAny<Cfg::BM>;
    using BL = MContigLoader4<Cfg::BN>;
    typename AL::Params pa{x, s, make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW), g.KH, g.KW, g.SH, g.SW, g.PH, g.PW};
    int M = s.N * s.OH * s.OW;
    EpiNCHW::Params pe{y, M, s.K, s.OH * s.OW, make_fastdiv(s.OH * s.OW), bias, act, slope};
    int Kt = g.KH * g.KW * round_bk(s.C);
    typename BL::Params pb{wp, Kt, round4(s.K), round4(s.K), 0};
    return launch_igemm<Cfg, AL, BL, EpiNCHW>(pa, pb, pe, M, s.K, Kt, 1, splits, st, slab);
}

// round 6: the run-time geometries on the igemm2 skeleton (ConvTapAnyA2: 4-byte LDS-DMA gather, 16 channels at one tap
// per chunk; EpiNCHWBiasAct: lean stores with bias + ReLU) for launches that fill the chip unsplit -- most of InceptionV3
// at the evaluation batch.  64-column tiles when the output channels are a multiple of 64 but not of 128 (192, 320, 448).
template <class Cfg>
static int run_fwd_any2(const float* x, const float* wp, const float* bias, float* y, const ConvShape& s, const AnyGeom& g,
                        int act, float slope, hipStream_t st, int y_image_channels) {
    using AL = ConvTapAnyA2<Cfg::BM>;
    using BL = MContigB2<Cfg::BN>;
    typename AL::Params pa{x, s, make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW), g.KH, g.KW, g.SH, g.SW, g.PH, g.PW};
    const int M = s.N * s.OH * s.OW;
    EpiNCHWBiasAct::Params pe{y, M, s.K, s.OH * s.OW, make_fastdiv(s.OH * s.OW), bias, act, slope, y_image_channels};
    const int Kt = g.KH * g.KW * round_bk(s.C);
    typename BL::Params pb{wp, Kt, round4(s.K), round4(s.K), 0};
    return launch_igemm2<Cfg, AL, BL, EpiNCHWBiasAct>(pa, pb, pe, M, s.K, Kt, 1, 1, st);
}
// 0: not applicable; 64: the tile's column count (256x64 tiles, three workgroups per CU: the 256x128 instantiation with
// the bias + ReLU epilogue needed 256 registers + 180 bytes of scratch and was not kept)
static int fwd_any2_cols(const ConvShape& s, const AnyGeom& g, const float* bias, int act) {
    if (knobs().no_igemm2 || knobs().no_any2 || !bias || !(act == ACT_NONE || act == ACT_RELU)) return 0;
    // (output channels: any multiple of 16 from 48 up -- a ragged last column tile stores through EpiNCHW's element-wise
    // path: 80, 96 and 48 of InceptionV3 fill 63-75 % of their last tile and still beat the 64x64 igemm_kernel tiles)
    if (s.C < BK || (s.K & 15) || s.K < 48 || g.KH * g.KW * (round_bk(s.C) / BK) < 8) return 0;
    const long long M = (long long)s.N * s.OH * s.OW;
    const long long t64 = ((M + 255) / 256) * ((s.K + 63) / 64);
    return t64 >= cus() * 7 / 8 ? 64 : 0;
}

static SplitPlan fwd_any_plan(const ConvShape& s, const AnyGeom& g) {
    long long M = (long long)s.N * s.OH * s.OW;
    return plan_split(M, s.K, g.KH * g.KW * round_bk(s.C), 1, pick_tile(M, s.K, 1));
}

// ---------------------------------------------------------------------------
// Dg with <= 4 image channels (the generator's output layer, 128 -> 3 @ 32 -> 64, and the
// discriminator's input gradient in the gradient penalty): an MFMA tile would waste 29 of 32
// columns, and the layer is HBM-bound anyway (reads 268 MB, 6.4 GFLOP at bs 512).  Direct VALU
// kernel: one lane per input position (n, a, b) produces the 2x2 output pixels of all channels
// from the 3x3 neighbourhood of y; lanes run along b so loads and the 8-byte stores coalesce; the
// per-(phase, ko, tap) weights are wave-uniform 16-byte rows of the packed dgrad image (scalar
// loads).  k4 s2 p1 only.
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void dgrad_smallc_k4s2p1_kernel(const float* __restrict__ y,
                                                                  const float* __restrict__ wp,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ x, ConvShape s,
                                                                  FastDiv div_ohw, FastDiv div_ow, int act,
                                                                  float slope) {
    const int OHW = s.OH * s.OW;
    const uint32_t M = (uint32_t)s.N * OHW;
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    const bool m_ok = m < M;
    const uint32_t n = fdiv(m, div_ohw);
    const uint32_t pix = m - n * (uint32_t)OHW;
    const int a = (int)fdiv(pix, div_ow);
    const int b = (int)(pix - (uint32_t)a * (uint32_t)s.OW);
    __amdgpu_buffer_rsrc_t rsrc = make_rsrc(y, (uint32_t)s.N * s.K * OHW * 4u);
    uint32_t voff[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int oy = a + dy - 1, ox = b + dx - 1;
            bool ok = m_ok && (unsigned)oy < (unsigned)s.OH && (unsigned)ox < (unsigned)s.OW;
            voff[dy][dx] = ok ? (n * (uint32_t)(s.K * OHW) + (uint32_t)(oy * s.OW + ox)) * 4u : OOB;
        }
    float acc[2][2][C];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[i][j][c] = 0.f;
    const long long phase_stride = (long long)s.K * 16;    // floats: K * 4 taps * ldc(4)
    for (int ko = 0; ko < s.K; ++ko) {
        float v[3][3];
        const uint32_t soff = (uint32_t)ko * (uint32_t)OHW * 4u;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v[dy][dx] = bload(rsrc, voff[dy][dx], soff);
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const float* wrow = wp + (py * 2 + px) * phase_stride + (long long)ko * 16;
#pragma unroll
                for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                    for (int tx = 0; tx < 2; ++tx) {
                        // oy = a + (py+1)/2 - ty  -> neighbourhood row index (oy - a + 1)
                        const float yv = v[(py + 1) / 2 - ty + 1][(px + 1) / 2 - tx + 1];
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wrow + (ty * 2 + tx) * 4);
#pragma unroll
                        for (int c = 0; c < C; ++c) acc[py][px][c] = fmaf(yv, w4[c], acc[py][px][c]);
                    }
            }
    }
    if (!m_ok) return;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float bv = bias ? bias[c] : 0.f;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            f32x2 o;
            o.x = act_fwd(acc[py][0][c] + bv, act, slope);
            o.y = act_fwd(acc[py][1][c] + bv, act, slope);
            float* dst = x + (((long long)n * C + c) * s.H + (2 * a + py)) * s.W + 2 * b;
            *reinterpret_cast<f32x2*>(dst) = o;
        }
    }
}

// Same operation, 4 input positions per lane (round 2).  The one-position kernel issues 9 dword loads per lane
// and feature channel -- every y value is requested 9 times, and at 256 B per wave-instruction the vector cache,
// not HBM, sets the pace (186 us for the 268 MB of G's last layer = 1.6 TB/s).  Here a lane owns (n, a, b..b+3):
// per channel it loads the three rows a-1, a, a+1 as ONE aligned 16-byte vector each and takes the two halo
// columns from its neighbour lanes (wave shuffles; at the image edge they are zero), i.e. 0.75 load instructions
// per position instead of 9, then runs the same 48 FMAs per position.  The eight outputs of an output row are two
// 16-byte stores.  Needs OW % 4 == 0 and 16-byte aligned rows.
// The four wavefronts of a workgroup own the SAME 64 lane positions and every fourth feature channel each (the
// channel loop is the only long dimension: at bs 128 one wavefront per 64 positions would leave the chip with 512
// wavefronts); their partial sums meet in LDS and wavefront 0 applies bias / activation and stores.
// KH = 5 (round 3): the 5x5 s2 p2 transposed convolution has the same 3 x 3 neighbourhood (rows a-1 .. a+1) and
// 9 / 6 / 6 / 4 taps per phase; its weights come in the tap-major pack (pack_dgrad_tap: [phase][tap][ko padded][4]).
// Round 5: the wavefront index is read into a scalar register (readfirstlane).  As a per-lane value it made the
// channel index "divergent" for the compiler: the 16 weight rows of a channel were fetched with sixteen 64-lane
// vector loads of ONE address each (1 KB through the vector cache per instruction, 16 KB per channel and wavefront --
// the vector cache, not the 96 packed FMAs, set the pace: 41 us at bs 128) and every y load sat in a waterfall loop.
// With a uniform index they are scalar loads into SGPRs again, as in the unsplit form.  KS = 8 (512 threads): eight
// wavefronts per 64 lane positions -- 4 per SIMD at bs 128 instead of 2; partial sums meet in a binary tree in LDS.
template <int C, int KS, int KH = 4>      // KS = 4 / 8: channel loop split over the workgroup's wavefronts; KS = 1: 256 lane positions
__global__ __launch_bounds__(KS > 4 ? 64 * KS : 256) void dgrad_smallc4_k4s2p1_kernel(const float* __restrict__ y,
                                                                   const float* __restrict__ wp,
                                                                   const float* __restrict__ bias,
                                                                   float* __restrict__ x, ConvShape s,
                                                                   FastDiv div_ohw4, FastDiv div_ow4, int act,
                                                                   float slope, const float* __restrict__ mask = nullptr,
                                                                   float mask_neg = 0.f) {
    // mask != nullptr (round 5, the input gradient of `LeakyReLU(conv(.))`): y is the gradient with respect to the
    // activation's OUTPUT and `mask` the saved forward output, same shape -- y * (mask > 0 ? 1 : mask_neg) is formed on
    // load (three more 16-byte loads per channel instead of an act_bwd launch: read 2, write 1, read 1 of the tensor)
    constexpr int P = KH == 4 ? 1 : 2, TMAX = (KH + 1) / 2;
    __shared__ float part[KS > 1 ? KS / 2 : 1][KS > 1 ? 16 * C : 1][64];
    const int OW4 = s.OW >> 2, OHW = s.OH * s.OW;
    const uint32_t M4 = (uint32_t)s.N * s.OH * OW4;
    const int lane = threadIdx.x & 63, wave = KS > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
    const uint32_t m = KS > 1 ? blockIdx.x * 64u + lane : blockIdx.x * 256u + threadIdx.x;
    const bool m_ok = m < M4;
    const uint32_t n = fdiv(m, div_ohw4);
    const uint32_t pix = m - n * (uint32_t)(s.OH * OW4);
    const int a = (int)fdiv(pix, div_ow4);
    const int b = (int)(pix - (uint32_t)a * (uint32_t)OW4) * 4;
    __amdgpu_buffer_rsrc_t rsrc = make_rsrc(y, (uint32_t)s.N * s.K * OHW * 4u);
    uint32_t voff[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        int oy = a + dy - 1;
        bool ok = m_ok && (unsigned)oy < (unsigned)s.OH;
        voff[dy] = ok ? (n * (uint32_t)(s.K * OHW) + (uint32_t)(oy * s.OW + b)) * 4u : OOB;
    }
    // halo columns come from the neighbour lanes; a lane at the left / right image edge has none (the lane next to
    // it then belongs to another row, or to another wave: both cases are exactly the edge cases)
    const bool has_l = b > 0, has_r = b + 4 < s.OW;
    float acc[4][2][2][C];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int c = 0; c < C; ++c) acc[q][i][j][c] = 0.f;
    const int kpad = round_bk(s.K);
    const long long phase_stride = KH == 4 ? (long long)s.K * 16             // floats: K * 4 taps * ldc(4)
                                           : (long long)TMAX * TMAX * kpad * 4;
    // the next channel's three rows are requested before this channel's 192 FMAs (round 4: with two wavefronts per SIMD
    // nothing else covers the load latency; G's last layer at bs 128: 56 us with the loads issued in place)
    // (the 256-position form, KS = 1, runs 4-7 wavefronts per SIMD and keeps its loads in place: with the prefetch's 26
    // extra registers it measured 80 -> 110 us at bs 256 and 113 -> 133 us at bs 512)
    const __amdgpu_buffer_rsrc_t rmask = make_rsrc(mask ? mask : y, (uint32_t)s.N * s.K * OHW * 4u);
    f32x4 nxt[3], mnx[3];
    if constexpr (KS > 1) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            nxt[dy] = bload4(rsrc, wave < s.K ? voff[dy] : OOB, (uint32_t)wave * (uint32_t)OHW * 4u);
            if (mask) mnx[dy] = bload4(rmask, wave < s.K ? voff[dy] : OOB, (uint32_t)wave * (uint32_t)OHW * 4u);
        }
    }
    for (int ko = wave; ko < s.K; ko += KS) {
        float v[3][6];
        f32x4 cur[3], mcur[3];
        if constexpr (KS > 1) {
            const bool more = ko + KS < s.K;
            const uint32_t soff = (uint32_t)(more ? ko + KS : ko) * (uint32_t)OHW * 4u;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                cur[dy] = nxt[dy];
                nxt[dy] = bload4(rsrc, more ? voff[dy] : OOB, soff);
                if (mask) {
                    mcur[dy] = mnx[dy];
                    mnx[dy] = bload4(rmask, more ? voff[dy] : OOB, soff);
                }
            }
        } else {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                cur[dy] = bload4(rsrc, voff[dy], (uint32_t)ko * (uint32_t)OHW * 4u);
                if (mask) mcur[dy] = bload4(rmask, voff[dy], (uint32_t)ko * (uint32_t)OHW * 4u);
            }
        }
        if (mask) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int e = 0; e < 4; ++e) cur[dy][e] = mcur[dy][e] > 0.f ? cur[dy][e] : cur[dy][e] * mask_neg;
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const f32x4 r = cur[dy];
            const float l = __shfl_up(r.w, 1), rr = __shfl_down(r.x, 1);
            v[dy][0] = has_l ? l : 0.f;
            v[dy][1] = r.x; v[dy][2] = r.y; v[dy][3] = r.z; v[dy][4] = r.w;
            v[dy][5] = has_r ? rr : 0.f;
        }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int ny = dg_taps(KH, 2, P, py), nx = dg_taps(KH, 2, P, px);
                const float* wrow = KH == 4 ? wp + (py * 2 + px) * phase_stride + (long long)ko * 16
                                            : wp + (py * 2 + px) * phase_stride + (long long)ko * 4;
#pragma unroll
                for (int ty = 0; ty < TMAX; ++ty)
#pragma unroll
                    for (int tx = 0; tx < TMAX; ++tx) {
                        if (ty < ny && tx < nx) {      // (folded after unrolling)
                            const f32x4 w4 = *reinterpret_cast<const f32x4*>(
                                KH == 4 ? wrow + (ty * 2 + tx) * 4 : wrow + (long long)(ty * nx + tx) * kpad * 4);
                            const int ry = (py + P) / 2 - ty + 1, rx = (px + P) / 2 - tx + 1;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int c = 0; c < C; ++c)
                                    acc[q][py][px][c] = fmaf(v[ry][q + rx], w4[c], acc[q][py][px][c]);
                        }
                    }
            }
    }
    if constexpr (KS > 1) {
        // binary tree, fixed order: wavefronts [h, 2h) hand their sums to wavefronts [0, h), h = KS/2 .. 1
#pragma unroll
        for (int h = KS / 2; h >= 1; h >>= 1) {
            if (wave >= h && wave < 2 * h) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int c = 0; c < C; ++c)
                                part[wave - h][((q * 2 + i) * 2 + j) * C + c][lane] = acc[q][i][j][c];
            }
            __syncthreads();
            if (wave < h) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int c = 0; c < C; ++c)
                                acc[q][i][j][c] += part[wave][((q * 2 + i) * 2 + j) * C + c][lane];
            }
            __syncthreads();
        }
        // the totals go back through LDS once more so that bias, activation (tanh in G's last layer: ~60 instructions
        // per value) and the 16-byte stores are shared by all KS wavefronts instead of wavefront 0 doing all 16 * C
        if (wave == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int c = 0; c < C; ++c) part[0][((q * 2 + i) * 2 + j) * C + c][lane] = acc[q][i][j][c];
        }
        __syncthreads();
        if (!m_ok) return;
        for (int g = wave; g < 4 * C; g += KS) {           // g = (c, py, h): one 16-byte store each
            const int c = g >> 2, py = (g >> 1) & 1, h = g & 1;
            const float bv = bias ? bias[c] : 0.f;
            float* dst = x + (((long long)n * C + c) * s.H + (2 * a + py)) * s.W + 2 * b;
            f32x4 o;
            o.x = act_fwd(part[0][(((2 * h) * 2 + py) * 2 + 0) * C + c][lane] + bv, act, slope);
            o.y = act_fwd(part[0][(((2 * h) * 2 + py) * 2 + 1) * C + c][lane] + bv, act, slope);
            o.z = act_fwd(part[0][(((2 * h + 1) * 2 + py) * 2 + 0) * C + c][lane] + bv, act, slope);
            o.w = act_fwd(part[0][(((2 * h + 1) * 2 + py) * 2 + 1) * C + c][lane] + bv, act, slope);
            *reinterpret_cast<f32x4*>(dst + 4 * h) = o;
        }
        return;
    }
    if (!m_ok) return;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float bv = bias ? bias[c] : 0.f;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            float* dst = x + (((long long)n * C + c) * s.H + (2 * a + py)) * s.W + 2 * b;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 o;
                o.x = act_fwd(acc[2 * h][py][0][c] + bv, act, slope);
                o.y = act_fwd(acc[2 * h][py][1][c] + bv, act, slope);
                o.z = act_fwd(acc[2 * h + 1][py][0][c] + bv, act, slope);
                o.w = act_fwd(acc[2 * h + 1][py][1][c] + bv, act, slope);
                *reinterpret_cast<f32x4*>(dst + 4 * h) = o;
            }
        }
    }
}

// How many wavefronts share the channel loop of 64 lane positions (the KS of dgrad_smallc4_k4s2p1_kernel): enough that
// every SIMD has about four wavefronts to switch between -- a launch has M4 / 64 * KS of them on 1024 SIMDs.
static int smallc_split(long long M4, int K) {
    const int forced = knobs().smallc_ks;
    if (forced == 1 || forced == 4 || forced == 8) return K >= 2 * forced || forced == 1 ? forced : 1;
    if (K < 16) return 1;
    if (M4 < knobs().smallc_split8_below && K >= 32) return 8;
    return M4 < knobs().smallc_split_below ? 4 : 1;
}

// 5x5 s2 p2 onto <= 4 channels (HoloGAN's critic: the gradient of its first convolution with respect to the image):
// the four-positions kernel only (rows of OW/4 lanes inside a wavefront, 16-byte aligned tensors, tap-major pack)
static bool dgrad_direct5_ok(const float* y, const float* x, const ConvShape& s) {
    const bool off = knobs().no_smallc || knobs().no_smallc5;
    return !off && s.C <= 4 && s.H == 2 * s.OH && s.W == 2 * s.OW && s.OW % 4 == 0 && 64 % (s.OW / 4) == 0 &&
           (((uintptr_t)y | (uintptr_t)x) & 15) == 0 && dgrad_tap_major(s.K, 5, 5, 2);
}

template <int C>
static int run_dgrad_smallc5(const float* y, const float* wp, const float* bias, float* x, const ConvShape& s, int act,
                             float slope, hipStream_t st) {
    const long long M4 = (long long)s.N * s.OH * s.OW / 4;
    const int ks = smallc_split(M4, s.K);
    if (ks == 8)
        hipLaunchKernelGGL((dgrad_smallc4_k4s2p1_kernel<C, 8, 5>), dim3((unsigned)((M4 + 63) / 64)), dim3(512), 0, st, y, wp,
                           bias, x, s, make_fastdiv(s.OH * (s.OW / 4)), make_fastdiv(s.OW / 4), act, slope,
                           (const float*)nullptr, 0.f);
    else if (ks == 4)
        hipLaunchKernelGGL((dgrad_smallc4_k4s2p1_kernel<C, 4, 5>), dim3((unsigned)((M4 + 63) / 64)), dim3(256), 0, st, y, wp,
                           bias, x, s, make_fastdiv(s.OH * (s.OW / 4)), make_fastdiv(s.OW / 4), act, slope,
                           (const float*)nullptr, 0.f);
    else
        hipLaunchKernelGGL((dgrad_smallc4_k4s2p1_kernel<C, 1, 5>), dim3((unsigned)((M4 + 255) / 256)), dim3(256), 0, st, y,
                           wp, bias, x, s, make_fastdiv(s.OH * (s.OW / 4)), make_fastdiv(s.OW / 4), act, slope,
                           (const float*)nullptr, 0.f);
    return launch_status();
}

template <int C>
static int run_dgrad_smallc(const float* y, const float* wp, const float* bias, float* x, const ConvShape& s, int act,
                            float slope, hipStream_t st, const float* mask = nullptr, float mask_neg = 0.f) {
    long long M = (long long)s.N * s.OH * s.OW;
    const bool one_pos = knobs().smallc_one_pos;          // experiment: the round-1 kernel
    // a row of OW/4 lanes must not straddle two wavefronts (the halo columns come from the neighbour LANES)
    if (!one_pos && s.OW % 4 == 0 && 64 % (s.OW / 4) == 0 && (((uintptr_t)y | (uintptr_t)x) & 15) == 0) {
        const long long M4 = M / 4;
        // few lane positions (bs 128 at 32x32: 512 wavefronts): split the channel loop over the workgroup instead
        // (0.080 -> 0.056 ms there; at bs 512 the unsplit form is 2x faster)
        // (round 3: measured crossover between bs 128 and bs 160 at 32x32 feature maps -- 32768 / 40960 lane positions;
        // bs 256: 114 -> 91 us for G's last layer without the split)
        const int ks = smallc_split(M4, s.K);
        if (ks == 8)
            hipLaunchKernelGGL((dgrad_smallc4_k4s2p1_kernel<C, 8>), dim3((unsigned)((M4 + 63) / 64)), dim3(512), 0, st, y,
                               wp, bias, x, s, make_fastdiv(s.OH * (s.OW / 4)), make_fastdiv(s.OW / 4), act, slope, mask,
                               mask_neg);
        else if (ks == 4)
            hipLaunchKernelGGL((dgrad_smallc4_k4s2p1_kernel<C, 4>), dim3((unsigned)((M4 + 63) / 64)), dim3(256), 0, st, y,
                               wp, bias, x, s, make_fastdiv(s.OH * (s.OW / 4)), make_fastdiv(s.OW / 4), act, slope, mask,
                               mask_neg);
        else
            hipLaunchKernelGGL((dgrad_smallc4_k4s2p1_kernel<C, 1>), dim3((unsigned)((M4 + 255) / 256)), dim3(256), 0, st,
                               y, wp, bias, x, s, make_fastdiv(s.OH * (s.OW / 4)), make_fastdiv(s.OW / 4), act, slope, mask,
                               mask_neg);
        return launch_status();
    }
    if (mask) return GZ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(dgrad_smallc_k4s2p1_kernel<C>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, y, wp, bias,
                       x, s, make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW), act, slope);
    return launch_status();
}

// ---------------------------------------------------------------------------
// Dg
// ---------------------------------------------------------------------------
template <class G, class Cfg>
static int run_dgrad(const float* y, const float* wp, const float* bias, float* x, const ConvShape& s, int act,
                     float slope, hipStream_t st, int splits = 1, float* slab = nullptr, float* stats = nullptr) {
    using AL = ConvDgALoader<Cfg::BM, G::kh, G::kw, G::s, G::p>;
    using BL = MContigLoader4<Cfg::BN>;
    using Epi = EpiPhaseB<G::s>;
    const int AH = s.H / G::s, AW = s.W / G::s;
    typename AL::Params pa{y, s, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW)};
    int Kg = s.K * AL::TAPS;
    int ldc = round4(s.C);
    typename BL::Params pb{wp, Kg, ldc, ldc, (long long)Kg * ldc};
    int M = s.N * AH * AW;
    typename Epi::Params pe{x, M, s.C, s.H, s.W, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW), bias, act, slope,
                            reinterpret_cast<f32x2*>(stats),
                            (splits > 1 && slab) ? (M + 31) / 32 : ((M + Cfg::BM - 1) / Cfg::BM) * Cfg::WM};
    if constexpr (!AL::FIXED) {
        if (dgrad_tap_major(s.K, G::kh, G::kw, G::s)) {
            using ALT = ConvDgALoaderTap<Cfg::BM, G::kh, G::kw, G::s, G::p>;
            typename ALT::Params pat{y, s, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW)};
            const int kpad = round_bk(s.K);
            int Kt = AL::TAPS * kpad;
            typename BL::Params pbt{wp, Kt, ldc, ldc, (long long)Kt * ldc};
            int pc[8];
            for (int ph = 0; ph < G::s * G::s; ++ph)
                pc[ph] = dg_taps(G::kh, G::s, G::p, ph / G::s) * dg_taps(G::kw, G::s, G::p, ph % G::s) * (kpad / BK);
            return launch_igemm<Cfg, ALT, BL, Epi>(pat, pbt, pe, M, s.C, Kt, G::s * G::s, splits, st, slab, pc);
        }
    }
    if constexpr (!AL::UNIFORM) {
        // phases differ in their tap count, hence in the length of their reduction
        int pc[8];
        for (int ph = 0; ph < G::s * G::s; ++ph)
            pc[ph] = (s.K * dg_taps(G::kh, G::s, G::p, ph / G::s) * dg_taps(G::kw, G::s, G::p, ph % G::s) + BK - 1) / BK;
        return launch_igemm<Cfg, AL, BL, Epi>(pa, pb, pe, M, s.C, Kg, G::s * G::s, splits, st, slab, pc);
    }
    if constexpr (G::kh == 4 && G::kw == 4 && G::s == 2 && G::p == 1) {
        const bool no_row4 = knobs().no_row4;          // experiment: the per-element loader
        if (!no_row4 && AW % 4 == 0 && (((uintptr_t)y) & 15) == 0) {
            using AR = ConvDgALoaderRow4<Cfg::BM, 4, 4, 2, 1>;
            return launch_igemm<Cfg, AR, BL, Epi>(pa, pb, pe, M, s.C, Kg, G::s * G::s, splits, st, slab);
        }
    }
    return launch_igemm<Cfg, AL, BL, Epi>(pa, pb, pe, M, s.C, Kg, G::s * G::s, splits, st, slab);
}

// k4 s2 p1 transposed convolution on the igemm2 skeleton (row-shared A rows by LDS-DMA, packed per-phase weights)
template <class Cfg>
static int run_dgrad2(const float* y, const float* wp, const float* bias, float* x, const ConvShape& s, int act,
                      float slope, hipStream_t st, float* stats = nullptr, int splits = 1, float* slab = nullptr) {
    using AL = ConvDgA2<Cfg::BM>;
    using BL = MContigB2<Cfg::BN>;
    using Epi = EpiPhaseB<2>;
    const int AH = s.H / 2, AW = s.W / 2;
    typename AL::Params pa{y, s, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW)};
    const int Kg = s.K * 4;
    const int ldc = round4(s.C);
    typename BL::Params pb{wp, Kg, ldc, ldc, (long long)Kg * ldc};
    const int M = s.N * AH * AW;
    // rows of partial statistics per phase: one per wavefront row of a tile, or -- split launch: the finish kernel
    // runs the epilogue per 32 x 32 block -- one per 32 pixels
    typename Epi::Params pe{x, M, s.C, s.H, s.W, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW), bias, act, slope,
                            reinterpret_cast<f32x2*>(stats),
                            (splits > 1 && slab) ? (M + 31) / 32 : ((M + Cfg::BM - 1) / Cfg::BM) * Cfg::WM};
    return launch_igemm2<Cfg, AL, BL, Epi>(pa, pb, pe, M, s.C, Kg, 4, splits, st, slab);
}

// transposed convolutions whose reduction is tap-major and whose phases differ in length (5x5 s2 p2: 9 / 6 / 6 / 4
// taps): gather loader with 4-byte LDS-DMA; the reduction is cut into pieces of ~48 chunks so that the phases'
// workgroups balance (unsplit, the 9-tap phase's workgroups would run 2.25x longer than the 4-tap phase's)
template <class G, class Cfg>
static int run_dgradtap2_impl(const float* y, const float* wp, const float* bias, float* x, const ConvShape& s, int act,
                              float slope, hipStream_t st, int splits, float* slab) {
    using AL = ConvDgTapA2<Cfg::BM, G::kh, G::kw, G::s, G::p>;
    using BL = MContigB2<Cfg::BN>;
    using Epi = EpiPhaseB<G::s>;
    const int AH = s.H / G::s, AW = s.W / G::s;
    typename AL::Params pa{y, s, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW)};
    const int kpad = round_bk(s.K);
    const int Kt = AL::TY * AL::TX * kpad;
    const int ldc = round4(s.C);
    // (1x1: the plain [K][C] image, without padding rows)
    const int Kb = G::kh * G::kw == 1 ? s.K : Kt;
    typename BL::Params pb{wp, Kb, ldc, ldc, (long long)Kb * ldc};
    const int M = s.N * AH * AW;
    typename Epi::Params pe{x, M, s.C, s.H, s.W, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW), bias, act, slope,
                            nullptr, 0};
    int pc[8];
    for (int ph = 0; ph < G::s * G::s; ++ph)
        pc[ph] = dg_taps(G::kh, G::s, G::p, ph / G::s) * dg_taps(G::kw, G::s, G::p, ph % G::s) * (kpad / BK);
    if constexpr (G::kh * G::kw == 1 && G::s == 1 && G::p == 0) {
        if (!knobs().no_plane_a && ((s.OH * s.OW) & 3) == 0 && !((uintptr_t)y & 15)) {
            using AP = PlaneA2<Cfg::BM>;
            typename AP::Params pp{y, s.K, s.OH * s.OW, M, make_fastdiv(s.OH * s.OW)};
            return launch_igemm2<Cfg, AP, BL, Epi>(pp, pb, pe, M, s.C, Kt, 1, splits, st, slab, pc);
        }
    }
    return launch_igemm2<Cfg, AL, BL, Epi>(pa, pb, pe, M, s.C, Kt, G::s * G::s, splits, st, slab, pc);
}

template <class G, class Cfg>
static int run_dgradtap2(const float* y, const float* wp, const float* bias, float* x, const ConvShape& s, int act,
                         float slope, hipStream_t st, int splits, float* slab) {
    if constexpr (has_own_igemm2_loaders<G>()) return GZ_ERR_UNSUPPORTED;     // (see run_fwdtap2)
    else return run_dgradtap2_impl<G, Cfg>(y, wp, bias, x, s, act, slope, st, splits, slab);
}

template <class G>
static SplitPlan dgradtap2_plan(const ConvShape& s) {
    const bool off = knobs().no_igemm2 || knobs().no_igemm2_tap;
    constexpr int TY = (G::kh + G::s - 1) / G::s, TX = (G::kw + G::s - 1) / G::s;
    constexpr bool one_by_one = G::kh * G::kw == 1;
    if constexpr (G::s * G::s > 8 || (!one_by_one && G::kh % G::s == 0 && G::kw % G::s == 0 && BK % (TY * TX) == 0))
        return SplitPlan{T64x64, 1};        // (k4 s2 p1 has its own loaders)
    // round 4: 256x64 tiles for the launches too small for 256x128; they also admit 64 image-side channels (tools/
    // tap_bench.py 64, TFLOP/s: HoloGAN 64x64 D.block2 / D.block3 input gradients 60 / 52 -> 77 / 67; EXT-128's
    // D.block1, 64 channels, 102 -> 123; with 48 instead of 32 chunks per piece 57 / 53 -- worse than the old kernel)
    const int mc64 = (!knobs().no_tile64 && G::s != 1 && (s.C & 63) == 0) ? knobs().tap64_min_chunks : 0;
    if (off || !(one_by_one ? s.K >= BK : dgrad_tap_major(s.K, G::kh, G::kw, G::s)) || s.C < (mc64 > 0 ? 64 : 128) ||
        (s.C & 3) || s.H % G::s || s.W % G::s)
        return SplitPlan{T64x64, 1};
    const long long M = (long long)s.N * (s.H / G::s) * (s.W / G::s);
    const long long tiles = ((M + 255) / 256) * ((s.C + 127) / 128);
    const int kblocks = round_bk(s.K) / BK;
    long long total = 0;
    for (int ph = 0; ph < G::s * G::s; ++ph)
        total += (long long)dg_taps(G::kh, G::s, G::p, ph / G::s) * dg_taps(G::kw, G::s, G::p, ph % G::s) * kblocks;
    const int maxchunks = TY * TX * kblocks;
    // one phase (stride 1): nothing to balance -- split only when the tiles alone do not fill the chip
    if (G::s == 1) {
        if (tiles >= cus() * 7 / 8 && maxchunks >= 8) return SplitPlan{T256x128, 1};
        if (tiles * total < 2LL * cus() * 32) return SplitPlan{T64x64, 1};
        int splits = (int)((cus() + tiles - 1) / tiles);
        while (splits > 1 && maxchunks / splits < 64) --splits;
        return (splits > 1 && tiles * splits >= cus() * 3 / 4) ? SplitPlan{T256x128, splits} : SplitPlan{T64x64, 1};
    }
    if (tiles * total < 2LL * cus() * 32 || s.C < 128) {       // too little work for the big tile
        if (mc64 > 0) {
            const long long t64 = ((M + 255) / 256) * (s.C / 64);
            // (64 channels = one column of tiles: only with enough of them -- the 64x64-image D.block1 has 64 tiles and
            // loses, 84 -> 76, on this skeleton)
            if (t64 * total >= 1LL * cus() * mc64 && (s.C >= 128 || t64 >= cus() * 3 / 4)) {
                const int wgs64 = knobs().tap_wgs * cus() / 256;
                long long cps = (t64 * total + wgs64 - 1) / wgs64;
                if (cps < mc64) cps = mc64;
                if (cps > knobs().tap_cps_max) cps = knobs().tap_cps_max;
                const int splits = (int)((maxchunks + cps - 1) / cps);
                return SplitPlan{T256x64, splits < 1 ? 1 : splits};
            }
        }
        return SplitPlan{T64x64, 1};
    }
    // ~384 workgroups of <= 96 chunks (measured on HoloGAN EXT-128's blocks, TFLOP/s of D.block2 / D.block3:
    // 256 workgroups 84 / 67, 384: 99 / 99, 512: 99 / 89, 768: 90 / 86; the round-2 kernels: 89 / 74)
    const int wgs = knobs().tap_wgs * cus() / 256, cps_max = knobs().tap_cps_max;
    long long cps = (tiles * total + wgs - 1) / wgs;
    if (cps < 32) cps = 32;
    if (cps > cps_max) cps = cps_max;
    const int splits = (int)((maxchunks + cps - 1) / cps);
    return SplitPlan{T256x128, splits < 1 ? 1 : splits};
}

// ---- 5x5 s2 p2 transposed convolution on row-shared LDS rows, all four output phases per workgroup (gz_igemm2.h:
// ConvDg5A2, DgQuadB2, EpiPhaseQuadB; round 6).  Tile = 256 pixels x 32 channels x (py, px); columns N = 4 C, one grid
// phase; 3 K/8 chunks of 12 k-steps, every workgroup identical.  Unsplit when every CU gets a workgroup, else the
// reduction is cut for ~512 workgroups (slabs + the finish pass).  Launches with little work per CU stay on the gather
// loader: measured (tools/tap_bench.py 64 / 128, TFLOP/s, this kernel vs the gather loader): EXT-128's critic at bs 64
// D.block1 125 vs 123, D.block2 111 vs 99, D.block3 102 vs 99, at bs 128 128 / 128 / 117; the 6.7-GFLOP layers of the
// 64 x 64 critic 65-72 vs 67-84.
struct Dg5Plan {
    bool ok;
    int splits, nz;
};
template <class G>
static bool dgrad5_shape_ok(const ConvShape& s) {
    return G::kh == 5 && G::kw == 5 && G::s == 2 && G::p == 2 && !knobs().no_igemm2 && !knobs().no_dg5 && s.H == 2 * s.OH &&
           s.W == 2 * s.OW && s.OW % 4 == 0 && 256 % s.OW == 0 && s.K % 16 == 0 && s.K >= 32 && s.C % 32 == 0 &&
           dgrad_tap_major(s.K, 5, 5, 2) && (long long)s.N * s.OH * s.OW >= 256;
}
template <class G>
static Dg5Plan dgrad5_plan(const ConvShape& s) {
    Dg5Plan p{false, 1, 1};
    if (!dgrad5_shape_ok<G>(s)) return p;
    const long long M = (long long)s.N * s.OH * s.OW;
    const long long tiles = ((M + 255) / 256) * (s.C / 32);
    const int chunks = 3 * s.K / 8;          // chunks of 8 LDS rows, 12 k-steps each
    if (tiles * chunks < 1LL * cus() * knobs().dg5_min_units) return p;
    p.ok = true;
    if (tiles >= cus()) return p;                                  // unsplit: every CU has a workgroup
    const long long want = (long long)knobs().dg5_wgs * cus() / 256;
    long long cps = (tiles * chunks + want - 1) / want;
    if (cps < knobs().dg5_min_chunks) cps = knobs().dg5_min_chunks;
    if (cps >= chunks) return p;
    p.splits = (int)((chunks + cps - 1) / cps);
    const int per = (chunks + p.splits - 1) / p.splits;           // (= launch_igemm2's chunks_per_split)
    p.nz = (chunks + per - 1) / per;
    if (p.nz <= 1) p.splits = p.nz = 1;
    return p;
}
template <class G>
static size_t dgrad5_ws_bytes(const ConvShape& s) {
    const Dg5Plan p = dgrad5_plan<G>(s);
    if (!p.ok || p.nz <= 1) return 0;
    return (size_t)p.nz * (size_t)s.N * s.OH * s.OW * (size_t)(4 * s.C) * 4;
}
using CfgQuad = TileCfg2<4, 1, 4, 2, 2>;       // four wavefronts of 64 pixels x (4 phases x 32 channels)
static int run_dgrad5(const float* y, const float* wp, float* x, const ConvShape& s, hipStream_t st, const Dg5Plan& plan,
                      float* slab) {
    using Cfg = CfgQuad;
    using AL = ConvDg5A2<Cfg::BM>;
    using BL = DgQuadB2;
    using Epi = EpiPhaseQuadB;
    const int AH = s.OH, AW = s.OW;
    typename AL::Params pa{y, s, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW)};
    typename BL::Params pb{wp, s.K, s.C, round4(s.C)};
    const int M = s.N * AH * AW;
    typename Epi::Params pe{x, M, s.C, s.H, s.W, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW), nullptr, ACT_NONE, 0.f,
                            nullptr, 0};
    return launch_igemm2<Cfg, AL, BL, Epi>(pa, pb, pe, M, 4 * s.C, 6 * s.K, 1, plan.splits, st, plan.nz > 1 ? slab : nullptr);
}

template <class G>
static bool dgrad2_ok(const ConvShape& s) {
    // ConvDgA2: 16-byte pieces of whole pixel quads; a tile's first pixel starts an image row (256 % AW == 0)
    // (the 512-pixel tile: 512 % AW == 0 follows)
    return G::kh == 4 && G::kw == 4 && G::s == 2 && G::p == 1 && s.H == 2 * s.OH && s.W == 2 * s.OW && s.OW % 4 == 0 &&
           256 % s.OW == 0 && s.K % 4 == 0;
}

template <class G>
static bool dgrad_direct(const float* x, const ConvShape& s) {
    return G::kh == 4 && G::kw == 4 && G::s == 2 && G::p == 1 && s.C <= 4 && s.H == 2 * s.OH && s.W == 2 * s.OW &&
           (((uintptr_t)x & 7) == 0) && !knobs().no_smallc;
}

template <class G>
static SplitPlan dgrad_plan(const ConvShape& s) {
    constexpr int TAPS = ((G::kh + G::s - 1) / G::s) * ((G::kw + G::s - 1) / G::s);
    long long M = (long long)s.N * (s.H / G::s) * (s.W / G::s);
    const int kk = dgrad_tap_major(s.K, G::kh, G::kw, G::s) ? round_bk(s.K) : s.K;
    if (dgrad2_ok<G>(s)) {
        const TileId t2 = pick_tile2(M, s.C, 4, s.K * 4);
        if (is_tile2(t2)) return SplitPlan{t2, 1};
        // 64-255 tiles of 256x128 (small batches, deep layers): cut the reduction so that >= 256 workgroups exist,
        // each with >= 32 chunks (as the forward convolution does)
        const bool off = knobs().no_igemm2;
        const bool t64 = !knobs().no_tile64 && (s.C & 63) == 0;
        const long long tiles = ((M + 255) / 256) * (t64 ? (s.C + 63) / 64 : (s.C + 127) / 128) * 4;
        const int chunks = s.K / 4;
        const int min_tiles = knobs().dg2_min_tiles * (t64 ? 2 : 1);
        if (!off && s.C >= 128 && tiles >= min_tiles && tiles < cus() && chunks >= 64) {
            int splits = (int)((cus() + tiles - 1) / tiles);
            while (splits > 1 && chunks / splits < 32) --splits;
            if (splits > 1 && tiles * splits >= cus()) return SplitPlan{t64 ? T256x64 : T256x128, splits};
        }
    }
    {
        const SplitPlan pt = dgradtap2_plan<G>(s);
        if (pt.tile == T256x128 || pt.tile == T256x64) return pt;
    }
    return plan_split(M, s.C, kk * TAPS, G::s * G::s, pick_tile(M, s.C, G::s * G::s, kk * TAPS));
}

template <class G>
static size_t dgrad_ws_bytes(const ConvShape& s) {
    constexpr int TAPS = ((G::kh + G::s - 1) / G::s) * ((G::kw + G::s - 1) / G::s);
    if (s.H % G::s || s.W % G::s || dgrad_direct<G>(nullptr, s)) return 0;
    const int kk = dgrad_tap_major(s.K, G::kh, G::kw, G::s) ? round_bk(s.K) : s.K;
    const size_t b = split_bytes(dgrad_plan<G>(s), (long long)s.N * (s.H / G::s) * (s.W / G::s), s.C, kk * TAPS, G::s * G::s);
    // (ConvDg5A2 needs aligned tensors, known only at the launch: the workspace serves either path)
    const size_t b5 = dgrad5_ws_bytes<G>(s);
    return b > b5 ? b : b5;
}

template <class G>
static int dispatch_dgrad(const float* y, const float* wp, const float* bias, float* x, const ConvShape& s,
                          int act, float slope, float* ws, size_t ws_bytes, hipStream_t st) {
    if (s.H % G::s || s.W % G::s) return GZ_ERR_UNSUPPORTED;
    if (dgrad_direct<G>(x, s)) {
        switch (s.C) {
            case 1: return run_dgrad_smallc<1>(y, wp, bias, x, s, act, slope, st);
            case 2: return run_dgrad_smallc<2>(y, wp, bias, x, s, act, slope, st);
            case 3: return run_dgrad_smallc<3>(y, wp, bias, x, s, act, slope, st);
            default: return run_dgrad_smallc<4>(y, wp, bias, x, s, act, slope, st);
        }
    }
    if constexpr (G::kh == 5 && G::kw == 5 && G::s == 2 && G::p == 2) {
        if (dgrad_direct5_ok(y, x, s)) {
            switch (s.C) {
                case 1: return run_dgrad_smallc5<1>(y, wp, bias, x, s, act, slope, st);
                case 2: return run_dgrad_smallc5<2>(y, wp, bias, x, s, act, slope, st);
                case 3: return run_dgrad_smallc5<3>(y, wp, bias, x, s, act, slope, st);
                default: return run_dgrad_smallc5<4>(y, wp, bias, x, s, act, slope, st);
            }
        }
    }
    long long M = (long long)s.N * (s.H / G::s) * (s.W / G::s);
    if constexpr (G::kh == 5 && G::kw == 5 && G::s == 2 && G::p == 2) {
        if (!bias && act == ACT_NONE && (((uintptr_t)y | (uintptr_t)x | (uintptr_t)wp) & 15) == 0) {
            const Dg5Plan p5 = dgrad5_plan<G>(s);
            if (p5.ok && (p5.nz <= 1 || (ws && ws_bytes >= dgrad5_ws_bytes<G>(s))))
                return run_dgrad5(y, wp, x, s, st, p5, ws);
        }
    }
    SplitPlan sp = dgrad_plan<G>(s);
    if (sp.splits > 1 && (!ws || ws_bytes < dgrad_ws_bytes<G>(s))) sp = SplitPlan{pick_tile(M, s.C, G::s * G::s), 1};
    float* slab = sp.splits > 1 ? ws : nullptr;
    if (is_tile2(sp.tile) && dgrad2_ok<G>(s) && (((uintptr_t)y) & 15) != 0)
        sp = SplitPlan{pick_tile(M, s.C, G::s * G::s), 1};        // unaligned tensor: the element-wise loaders
    switch (sp.tile) {
        case T256x256: return run_dgrad2<Cfg256x256>(y, wp, bias, x, s, act, slope, st);
        case T256x128:
            if (!dgrad2_ok<G>(s)) return run_dgradtap2<G, Cfg256x128>(y, wp, bias, x, s, act, slope, st, sp.splits, slab);
            return run_dgrad2<Cfg256x128>(y, wp, bias, x, s, act, slope, st, nullptr, sp.splits, slab);
        case T512x64: return run_dgrad2<Cfg512x64>(y, wp, bias, x, s, act, slope, st);
        case T256x64:
            if (!dgrad2_ok<G>(s)) return run_dgradtap2<G, Cfg256x64>(y, wp, bias, x, s, act, slope, st, sp.splits, slab);
            return run_dgrad2<Cfg256x64>(y, wp, bias, x, s, act, slope, st, nullptr, sp.splits, slab);
        case T128x128: return run_dgrad<G, Cfg128x128>(y, wp, bias, x, s, act, slope, st, sp.splits, slab);
        case T128x64: return run_dgrad<G, Cfg128x64>(y, wp, bias, x, s, act, slope, st, sp.splits, slab);
        case T128x32: return run_dgrad<G, Cfg128x32>(y, wp, bias, x, s, act, slope, st, sp.splits, slab);
        default: return run_dgrad<G, Cfg64x64>(y, wp, bias, x, s, act, slope, st, sp.splits, slab);
    }
}

// ---------------------------------------------------------------------------
// F / Dg of 3x3 s1 p1 layers with <= 32 channels on both sides (same layers as wgrad_smallch below).  In the
// 128x32 implicit-GEMM tile half of the MFMA columns are padding when K = 16 and the three kx taps of a row are
// fetched three times.  Here a wavefront owns 16 consecutive pixels of one output row: B = x[4 channels][16
// pixels] comes from ONE dword load per lane and row, the kx = -1 / +1 operands are the same registers shifted
// by one lane (plus one predicated edge load), A = w[16 output channels][4 input channels] of the tap is read
// from an LDS copy of the packed weights staged once per workgroup, and v_mfma_f32_16x16x4_f32 accumulates
// D[output channel][pixel] -- stores are 64-byte runs per channel.  Dg is the same kernel on gy with the taps
// mirrored (tap' = 8 - tap) and the roles of K and C exchanged; both read the weight images the implicit-GEMM
// path uses (tap-major when the input side has >= 16 channels, (c, tap)-major otherwise).
// ---------------------------------------------------------------------------
template <int OT, int IT>     // 16-channel blocks on the output / input side
__global__ __launch_bounds__(256) void conv3x3_smallch_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int N, int CI, int CO, int H, int W, int groups,
                                                              FastDiv div_seg, FastDiv div_h, int tap_major, int inpad,
                                                              int ld, int flip, int act, float slope) {
    constexpr int CB = IT * 4;                       // input-channel quads
    __shared__ float Ws[9 * CB * OT * 64];           // [tap][cb][ot][q][i]
    for (int e = threadIdx.x; e < 9 * CB * OT * 64; e += 256) {
        const int i = e & 15, q = (e >> 4) & 3;
        int rest = e >> 6;
        const int ot = rest % OT;
        rest /= OT;
        const int cb = rest % CB, tap = rest / CB;
        const int co = ot * 16 + i, ci = cb * 4 + q;
        const int t = flip ? 8 - tap : tap;
        float v = 0.f;
        if (co < CO && ci < CI) v = wp[(long long)(tap_major ? t * inpad + ci : ci * 9 + t) * ld + co];
        Ws[e] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int HW = H * W;
    // descriptor moved back by one image row: the per-lane voffset addresses row h - 1 + 1 = h of channel q, the
    // (row, channel-quad) step is a wave-uniform scalar offset, so the inner loop has no per-lane address math
    const __amdgpu_buffer_rsrc_t rin = make_rsrc(reinterpret_cast<const char*>(in) - (size_t)W * 4,
                                                 (uint32_t)N * CI * HW * 4u + (uint32_t)W * 4u);
    const int segs = W >> 4;
    const int nwaves = gridDim.x * 4;
    for (int g = blockIdx.x * 4 + wave; g < groups; g += nwaves) {
        const uint32_t rowid = fdiv((uint32_t)g, div_seg);               // n * H + h
        const int w0 = (g - (int)rowid * segs) << 4;
        const uint32_t n = fdiv(rowid, div_h);
        const int h = (int)(rowid - n * (uint32_t)H);
        const uint32_t vmain = ((n * (uint32_t)CI + q) * (uint32_t)HW + (uint32_t)(h * W + w0 + i)) * 4u;
        uint32_t vedge = OOB;
        if (i == 0 && w0 > 0) vedge = vmain - 4u;
        if (i == 15 && w0 + 16 < W) vedge = vmain + 4u;
        f32x4 acc[OT];
#pragma unroll
        for (int a = 0; a < OT; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int hh = h + dy - 1;
            if ((unsigned)hh >= (unsigned)H) continue;                   // wave-uniform
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const bool cok = cb * 4 + q < CI;
                const uint32_t soff = (uint32_t)(cb * 4 * HW + dy * W) * 4u;
                const float v0 = bload(rin, cok ? vmain : OOB, soff);
                // lanes 0 / 15 of each 16-lane row fetch the pixel left / right of the segment (zero in the padding)
                const float ev = bload(rin, cok ? vedge : OOB, soff);
                float left = __shfl_up(v0, 1, 16), right = __shfl_down(v0, 1, 16);
                if (i == 0) left = ev;
                if (i == 15) right = ev;
                const float* wrow = Ws + ((dy * 3) * CB + cb) * OT * 64 + lane;
#pragma unroll
                for (int a = 0; a < OT; ++a) {
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[(0 * CB) * OT * 64 + a * 64], left, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[(1 * CB) * OT * 64 + a * 64], v0, acc[a], 0, 0, 0);
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[(2 * CB) * OT * 64 + a * 64], right, acc[a], 0, 0, 0);
                }
            }
        }
        // D[row = output channel 4q + r][column = pixel i]
#pragma unroll
        for (int a = 0; a < OT; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = a * 16 + 4 * q + r;
                if (co < CO) {
                    const float bv = bias ? bias[co] : 0.f;
                    out[((long long)(n * (uint32_t)CO + co) * H + h) * W + w0 + i] = act_fwd(acc[a][r] + bv, act, slope);
                }
            }
    }
}

// ... with at most 4 output channels on 64-pixel-wide maps (HoloGAN's last layer, 64 -> 3 + tanh): the MFMA tile above
// multiplies 13 padding rows of 16 (81 us for one read of a 67 MB activation).  Plain FMAs: a workgroup owns an
// 8-row x 64-pixel block of one sample, lane = (row, 8-pixel segment) as in wgrad_k3_fewk_kernel; its four wavefronts
// take a quarter of the input channels each and meet in LDS in a fixed order.  Same weight images, `flip` as above.
template <int KK>
__global__ __launch_bounds__(256) void conv3x3_fewk_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                           const float* __restrict__ bias, float* __restrict__ out, int N,
                                                           int CI, int CO, int H, int tap_major, int inpad, int ld,
                                                           int flip, int act, float slope) {
    __shared__ float Ws[64 * 9 * KK];                // [ci][tap][k]
    __shared__ float red[3][KK * 8][64];
    for (int e = threadIdx.x; e < CI * 9 * KK; e += 256) {
        const int k = e % KK, tap = (e / KK) % 9, ci = e / (9 * KK);
        const int t = flip ? 8 - tap : tap;
        Ws[e] = k < CO ? wp[(long long)(tap_major ? t * inpad + ci : ci * 9 + t) * ld + k] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rbs = H >> 3, HW = H * 64;
    const int n = blockIdx.x / rbs, rb = blockIdx.x - n * rbs;
    const int r = lane >> 3, sg = lane & 7, h = rb * 8 + r, w0 = sg * 8;
    const int cq = (CI + 3) >> 2, c0 = wave * cq, c1 = min(CI, c0 + cq);
    float acc[KK][8];
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    const float* xn = in + (long long)n * CI * HW + w0;
    // the next channel's rows are requested before this channel's FMAs (two wavefronts per SIMD: nothing else covers
    // the load latency)
    auto load_rows = [&](int ci, f32x4 (&rws)[6]) {
        const float* xc = xn + (long long)ci * HW;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int hh = h + d - 1;
            const bool ok = (unsigned)hh < (unsigned)H;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            rws[2 * d] = ok ? *reinterpret_cast<const f32x4*>(xc + hh * 64) : z;
            rws[2 * d + 1] = ok ? *reinterpret_cast<const f32x4*>(xc + hh * 64 + 4) : z;
        }
    };
    f32x4 nxt[6];
    if (c0 < c1) load_rows(c0, nxt);
    for (int ci = c0; ci < c1; ++ci) {
        f32x4 cur[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) cur[e] = nxt[e];
        if (ci + 1 < c1) load_rows(ci + 1, nxt);
        float xr[3][10];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const f32x4 v0 = cur[2 * d], v1 = cur[2 * d + 1];
            const float left = __shfl_up(v1[3], 1, 64), right = __shfl_down(v0[0], 1, 64);
            xr[d][0] = sg == 0 ? 0.f : left;
            xr[d][1] = v0[0]; xr[d][2] = v0[1]; xr[d][3] = v0[2]; xr[d][4] = v0[3];
            xr[d][5] = v1[0]; xr[d][6] = v1[1]; xr[d][7] = v1[2]; xr[d][8] = v1[3];
            xr[d][9] = sg == 7 ? 0.f : right;
        }
        const float* wc = Ws + ci * 9 * KK;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    const float wv = wc[(d * 3 + tx) * KK + k];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(wv, xr[d][j + tx], acc[k][j]);
                }
    }
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < KK; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wave - 1][k * 8 + j][lane] = acc[k][j];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            if (k >= CO) continue;
            const float bv = bias ? bias[k] : 0.f;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = ((acc[k][j] + red[0][k * 8 + j][lane]) + red[1][k * 8 + j][lane]) + red[2][k * 8 + j][lane];
                o[j] = act_fwd(v + bv, act, slope);
            }
            float* dst = out + ((long long)(n * CO + k) * H + h) * 64 + w0;
            *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
        }
    }
}

static bool conv3_fewk_ok(int CI, int CO, int H, int W, const void* in, const void* out) {
    return !knobs().no_fewk_conv && CO <= 4 && CI <= 64 && W == 64 && (H & 7) == 0 &&
           !(((uintptr_t)in | (uintptr_t)out) & 15);
}

static bool conv3_smallch_ok(int N, int CI, int CO, int H, int W) {
    const bool off = knobs().no_smallch_conv;
    // measured against the implicit-GEMM path (tools/resnet_bench.py): wins when the output side fits one 16-row
    // MFMA tile and the input side fills at least half a 16-channel block (16->16 @ 128x128: 123 -> 72 us);
    // loses for 3 input channels (K dimension mostly padding) and for 32 output channels
    return !off && CO <= 16 && CI > 8 && CI <= 64 && (W & 15) == 0 && (long long)N * H * W >= 65536;
}

static int run_conv3_smallch(const float* in, const float* wp, const float* bias, float* out, int N, int CI, int CO,
                             int H, int W, int tap_major, int flip, int act, float slope, hipStream_t st) {
    if (conv3_fewk_ok(CI, CO, H, W, in, out)) {
        const dim3 grid((unsigned)(N * (H >> 3)));
#define GZ_FEWK(KK_)                                                                                                  \
    hipLaunchKernelGGL((conv3x3_fewk_kernel<KK_>), grid, dim3(256), 0, st, in, wp, bias, out, N, CI, CO, H, tap_major, \
                       round_bk(CI), round4(CO), flip, act, slope)
        if (CO == 1) GZ_FEWK(1);
        else if (CO == 2) GZ_FEWK(2);
        else if (CO == 3) GZ_FEWK(3);
        else GZ_FEWK(4);
#undef GZ_FEWK
        return launch_status();
    }
    const int groups = N * H * (W >> 4);
    const int gpw = knobs().c3_gpw;
    long long blocks = (groups + 4 * gpw - 1) / (4 * gpw);   // >= gpw pixel groups per wavefront: the weights are staged per workgroup
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    const FastDiv dseg = make_fastdiv(W >> 4), dh = make_fastdiv(H);
    const int ot = (CO + 15) / 16, it = (CI + 15) / 16;
#define GZ_C3(OT_, IT_)                                                                                              \
    hipLaunchKernelGGL((conv3x3_smallch_kernel<OT_, IT_>), dim3((unsigned)blocks), dim3(256), 0, st, in, wp, bias, out, \
                       N, CI, CO, H, W, groups, dseg, dh, tap_major, round_bk(CI), round4(CO), flip, act, slope)
    (void)ot;                                        // conv3_smallch_ok admits one output block only
    if (it == 1) GZ_C3(1, 1);
    else if (it == 2) GZ_C3(1, 2);
    else GZ_C3(1, 4);
#undef GZ_C3
    return launch_status();
}

// ---------------------------------------------------------------------------
// Wg of 3x3 s1 p1 layers with few channels (ceil(K/16) * ceil(C/16) <= 4: the 128x128 / 64x64 stages of the R1
// ResNets, the image-side convolutions 64 -> 3 / 3 -> 16).  As an implicit GEMM this is M = K <= 32 rows of a 64-row tile: three quarters of
// the MFMA work multiplies padding.  Here one wavefront owns 16 consecutive pixels of one image row and issues
// v_mfma_f32_16x16x4_f32 with A = y[ko][4 pixels], B = x[c][the same 4 pixels shifted by the tap]: 9 * KT * CT
// exact 16x16 tiles, nothing padded.  Lane (i = l & 15, q = l >> 4) loads ONE aligned float4 per operand row i
// (pixels 4q..4q+3); the dx = -1 / +1 taps are assembled from the neighbouring lanes' vectors (lane +-16) plus
// one edge dword, rows in the vertical padding are skipped wave-uniformly.  Each workgroup sums its four
// wavefronts through LDS in a fixed order and writes one slab; reduce_slabs_kernel adds the slabs.
// ---------------------------------------------------------------------------
template <int KT, int CT>
__global__ __launch_bounds__(256) void wgrad_smallch_k3_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               float* __restrict__ slab, ConvShape s, int groups,
                                                               FastDiv div_seg, FastDiv div_h) {
    constexpr int NACC = KT * CT * 9;
    __shared__ float red[KT * CT * 9 * 256];
    __shared__ float bsum[4][KT * 16];
    float ysum[KT];                 // the bias gradient sum_pixels y[ko] comes for free: y is read here anyway
#pragma unroll
    for (int a = 0; a < KT; ++a) ysum[a] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int H = s.H, W = s.W, HW = s.H * s.W;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (uint32_t)s.N * s.C * HW * 4u);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y, (uint32_t)s.N * s.K * HW * 4u);
    f32x4 acc[KT][CT][9];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[a][b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int segs = W >> 4;
    const int nwaves = gridDim.x * 4;
    for (int g = blockIdx.x * 4 + wave; g < groups; g += nwaves) {
        const uint32_t rowid = fdiv((uint32_t)g, div_seg);               // n * H + h
        const int w0 = (g - (int)rowid * segs) << 4;
        const uint32_t n = fdiv(rowid, div_h);
        const int h = (int)(rowid - n * (uint32_t)H);
        const int wq = w0 + 4 * q;
        f32x4 yv[KT];
#pragma unroll
        for (int a = 0; a < KT; ++a) {
            const int ko = a * 16 + i;
            yv[a] = bload4(ry, ko < s.K ? ((n * (uint32_t)s.K + ko) * (uint32_t)HW + (uint32_t)(h * W + wq)) * 4u : OOB, 0);
            ysum[a] += (yv[a][0] + yv[a][1]) + (yv[a][2] + yv[a][3]);
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int hh = h + dy - 1;
            if ((unsigned)hh >= (unsigned)H) continue;                   // wave-uniform: a padding row contributes nothing
#pragma unroll
            for (int b = 0; b < CT; ++b) {
                const int c = b * 16 + i;
                const uint32_t base = ((n * (uint32_t)s.C + c) * (uint32_t)HW + (uint32_t)(hh * W + wq)) * 4u;
                const bool cok = c < s.C;
                const f32x4 cv = bload4(rx, cok ? base : OOB, 0);
                // the pixel left of this lane's vector: lane - 16 holds it, except for q == 0 (previous segment / padding)
                float left = __shfl_up(cv[3], 16, 64);
                float right = __shfl_down(cv[0], 16, 64);
                const float el = bload(rx, (cok && q == 0 && wq > 0) ? base - 4u : OOB, 0);
                const float er = bload(rx, (cok && q == 3 && wq + 4 < W) ? base + 16u : OOB, 0);
                if (q == 0) left = el;
                if (q == 3) right = er;
                const f32x4 lv = {left, cv[0], cv[1], cv[2]};
                const f32x4 rv = {cv[1], cv[2], cv[3], right};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int a = 0; a < KT; ++a) {
                        acc[a][b][dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a][j], lv[j], acc[a][b][dy * 3 + 0], 0, 0, 0);
                        acc[a][b][dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a][j], cv[j], acc[a][b][dy * 3 + 1], 0, 0, 0);
                        acc[a][b][dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(yv[a][j], rv[j], acc[a][b][dy * 3 + 2], 0, 0, 0);
                    }
                }
            }
        }
    }
    // workgroup sum in a fixed order: wave 0 stores, waves 1..3 add in turn
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int a = 0; a < KT; ++a)
#pragma unroll
                for (int b = 0; b < CT; ++b)
#pragma unroll
                    for (int t = 0; t < 9; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* dst = &red[(((a * CT + b) * 9 + t) * 4 + r) * 64 + lane];
                            *dst = (w == 0 ? 0.f : *dst) + acc[a][b][t][r];
                        }
        }
        __syncthreads();
    }
    // bias gradient: lanes i, i+16, i+32, i+48 hold the same channel; then the four wavefronts in order
#pragma unroll
    for (int a = 0; a < KT; ++a) {
        float v = ysum[a];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16) bsum[wave][a * 16 + lane] = v;
    }
    __syncthreads();
    // slab row of this workgroup: K*C*9 weight-gradient entries followed by K bias-gradient entries
    float* out = slab + (long long)blockIdx.x * ((long long)s.K * s.C * 9 + s.K);
    if (threadIdx.x < KT * 16 && (int)threadIdx.x < s.K)
        out[(long long)s.K * s.C * 9 + threadIdx.x] =
            ((bsum[0][threadIdx.x] + bsum[1][threadIdx.x]) + bsum[2][threadIdx.x]) + bsum[3][threadIdx.x];
    // D layout of the 16x16 tile: register r of lane l is (row 4 * (l >> 4) + r, column l & 15) = (ko, c)
    for (int e = threadIdx.x; e < NACC * 256; e += 256) {
        const int l = e & 63, r = (e >> 6) & 3, rest = e >> 8;
        const int t = rest % 9, ab = rest / 9;
        const int ko = (ab / CT) * 16 + 4 * (l >> 4) + r, c = (ab % CT) * 16 + (l & 15);
        if (ko < s.K && c < s.C) out[((long long)ko * s.C + c) * 9 + t] = red[e];
    }
}

// ---------------------------------------------------------------------------
// ... and with at most 4 OUTPUT channels on 64-pixel-wide maps (HoloGAN's last layer, Conv2d(64, 3, k3, p1) at 64x64,
// core/models/hologan_generator.py:65): the 16x16x4 MFMA above multiplies 13 padding rows out of 16 and ran at
// 6 TFLOP/s (150 us for 0.9 GFLOP; the layer is one read of a 67 MB activation).  Plain FMAs instead: a wavefront owns
// one input channel and an 8-row x 64-pixel block -- lane = (row, 8-pixel segment) -- keeps the 9 * K sums of its
// channel in registers while it walks the samples of its slice, and adds the 64 lanes once at the end.  The image rows
// come as aligned float4 loads, the two halo pixels of a segment from the neighbouring lanes.  One slab row per
// (row block, sample slice); reduce_slabs_kernel adds them (and the bias gradient in the row's tail).
// ---------------------------------------------------------------------------
template <int KK>
__global__ __launch_bounds__(256) void wgrad_k3_fewk_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            float* __restrict__ slab, ConvShape s, int nslices,
                                                            int n_per_slice) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int H = s.H, HW = s.H * 64, rbs = s.H >> 3;
    int b = blockIdx.x;
    const int ns = b % nslices;
    b /= nslices;
    const int rb = b % rbs, c = (b / rbs) * 4 + wave;
    if (c >= s.C) return;                                     // (wave-uniform)
    const int r = lane >> 3, sg = lane & 7, h = rb * 8 + r, w0 = sg * 8;
    float acc[KK][9], ysum[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) {
        ysum[k] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[k][t] = 0.f;
    }
    const int n0 = ns * n_per_slice, n1 = min(s.N, n0 + n_per_slice);
    // the next sample's rows are requested before this sample's FMAs (see conv3x3_fewk_kernel)
    auto load_rows = [&](int n, f32x4 (&rws)[6], f32x4 (&yws)[2 * KK]) {
        const float* xc = x + ((long long)n * s.C + c) * HW + w0;
        const float* yn = y + (long long)n * s.K * HW + h * 64 + w0;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int hh = h + d - 1;
            const bool ok = (unsigned)hh < (unsigned)H;
            rws[2 * d] = ok ? *reinterpret_cast<const f32x4*>(xc + hh * 64) : z;
            rws[2 * d + 1] = ok ? *reinterpret_cast<const f32x4*>(xc + hh * 64 + 4) : z;
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            yws[2 * k] = k < s.K ? *reinterpret_cast<const f32x4*>(yn + (long long)k * HW) : z;
            yws[2 * k + 1] = k < s.K ? *reinterpret_cast<const f32x4*>(yn + (long long)k * HW + 4) : z;
        }
    };
    f32x4 nxt[6], ynx[2 * KK];
    if (n0 < n1) load_rows(n0, nxt, ynx);
    for (int n = n0; n < n1; ++n) {
        f32x4 cur[6], ycur[2 * KK];
#pragma unroll
        for (int e = 0; e < 6; ++e) cur[e] = nxt[e];
#pragma unroll
        for (int e = 0; e < 2 * KK; ++e) ycur[e] = ynx[e];
        if (n + 1 < n1) load_rows(n + 1, nxt, ynx);
        float xr[3][10];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const f32x4 v0 = cur[2 * d], v1 = cur[2 * d + 1];
            const float left = __shfl_up(v1[3], 1, 64), right = __shfl_down(v0[0], 1, 64);
            xr[d][0] = sg == 0 ? 0.f : left;
            xr[d][1] = v0[0]; xr[d][2] = v0[1]; xr[d][3] = v0[2]; xr[d][4] = v0[3];
            xr[d][5] = v1[0]; xr[d][6] = v1[1]; xr[d][7] = v1[2]; xr[d][8] = v1[3];
            xr[d][9] = sg == 7 ? 0.f : right;
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            float yv[8];
            {
                const f32x4 a = ycur[2 * k], bq = ycur[2 * k + 1];
                yv[0] = a[0]; yv[1] = a[1]; yv[2] = a[2]; yv[3] = a[3];
                yv[4] = bq[0]; yv[5] = bq[1]; yv[6] = bq[2]; yv[7] = bq[3];
            }
            if (c == 0) ysum[k] += ((yv[0] + yv[1]) + (yv[2] + yv[3])) + ((yv[4] + yv[5]) + (yv[6] + yv[7]));
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    float a = acc[k][d * 3 + tx];
#pragma unroll
                    for (int j = 0; j < 8; ++j) a = fmaf(yv[j], xr[d][j + tx], a);
                    acc[k][d * 3 + tx] = a;
                }
        }
    }
    const long long count = (long long)s.K * s.C * 9;
    float* out = slab + (long long)(rb * nslices + ns) * (count + s.K);
#pragma unroll
    for (int k = 0; k < KK; ++k) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = wave_sum(acc[k][t]);
            if (lane == 0 && k < s.K) out[((long long)k * s.C + c) * 9 + t] = v;
        }
        if (c == 0) {
            const float v = wave_sum(ysum[k]);
            if (lane == 0 && k < s.K) out[count + k] = v;
        }
    }
}

static bool wgrad_fewk_ok(const ConvShape& s) {
    return !knobs().no_fewk_wg && s.K <= 4 && s.W == 64 && (s.H & 7) == 0 && s.OH == s.H && s.OW == s.W;
}

static int wgrad_fewk_slices(const ConvShape& s) {       // sample slices: ~4 workgroups per CU in all
    const long long base = (long long)((s.C + 3) / 4) * (s.H >> 3);
    long long want = (4LL * cus() + base - 1) / base;
    if (want > s.N) want = s.N;
    if (want < 1) want = 1;
    const int per = (int)((s.N + want - 1) / want);
    return (s.N + per - 1) / per;
}

static bool wgrad_smallch_ok(const ConvShape& s, int KH, int KW, int S, int P) {
    const bool off = knobs().no_smallch_wg;
    const int kt = (s.K + 15) / 16, ct = (s.C + 15) / 16;      // 16x16 tiles per tap: at most 4 (36 accumulators)
    return !off && KH == 3 && KW == 3 && S == 1 && P == 1 && kt * ct <= 4 && (s.W & 15) == 0 &&
           (long long)s.N * s.H * s.W >= 65536;
}

static int wgrad_smallch_blocks(const ConvShape& s) {
    if (wgrad_fewk_ok(s)) return (s.H >> 3) * wgrad_fewk_slices(s);       // slab rows
    long long groups = (long long)s.N * s.H * (s.W >> 4);
    long long blocks = (groups + 15) / 16;          // >= 4 pixel groups per wavefront
    return (int)(blocks > 1024 ? 1024 : (blocks < 1 ? 1 : blocks));
}

static int run_wgrad_smallch(const float* x, const float* y, float* dw, float* dbias, float* ws, size_t ws_bytes,
                             const ConvShape& s, hipStream_t st) {
    const int blocks = wgrad_smallch_blocks(s);
    const long long count = (long long)s.K * s.C * 9;
    if (!ws || ws_bytes < (size_t)blocks * (count + s.K) * 4) return GZ_ERR_WORKSPACE;
    const long long row = count + s.K;
    const long long outs = dbias ? row : count;
    if (wgrad_fewk_ok(s)) {
        const int nsl = wgrad_fewk_slices(s), per = (s.N + nsl - 1) / nsl;
        const dim3 grid((unsigned)(((s.C + 3) / 4) * (s.H >> 3) * nsl));
        switch (s.K) {
            case 1: hipLaunchKernelGGL((wgrad_k3_fewk_kernel<1>), grid, dim3(256), 0, st, x, y, ws, s, nsl, per); break;
            case 2: hipLaunchKernelGGL((wgrad_k3_fewk_kernel<2>), grid, dim3(256), 0, st, x, y, ws, s, nsl, per); break;
            case 3: hipLaunchKernelGGL((wgrad_k3_fewk_kernel<3>), grid, dim3(256), 0, st, x, y, ws, s, nsl, per); break;
            default: hipLaunchKernelGGL((wgrad_k3_fewk_kernel<4>), grid, dim3(256), 0, st, x, y, ws, s, nsl, per);
        }
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((outs + 63) / 64)), dim3(64 * RS_WAVES), 0, st, ws, dw, blocks,
                           outs, row, dbias, count);
        return launch_status();
    }
    const int groups = s.N * s.H * (s.W >> 4);
    const FastDiv dseg = make_fastdiv(s.W >> 4), dh = make_fastdiv(s.H);
    const int kt = (s.K + 15) / 16, ct = (s.C + 15) / 16;
#define GZ_SMALLCH(KT_, CT_) \
    hipLaunchKernelGGL((wgrad_smallch_k3_kernel<KT_, CT_>), dim3(blocks), dim3(256), 0, st, x, y, ws, s, groups, dseg, dh)
    if (kt == 1 && ct == 1) GZ_SMALLCH(1, 1);
    else if (kt == 1 && ct == 2) GZ_SMALLCH(1, 2);
    else if (kt == 2 && ct == 1) GZ_SMALLCH(2, 1);
    else if (kt == 2 && ct == 2) GZ_SMALLCH(2, 2);
    else if (kt == 1 && ct <= 4) GZ_SMALLCH(1, 4);
    else GZ_SMALLCH(4, 1);
#undef GZ_SMALLCH
    // slab rows are count + K long; without a dbias pointer the K-long tails are simply not reduced
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((outs + 63) / 64)), dim3(64 * RS_WAVES), 0, st, ws, dw, blocks,
                       outs, row, dbias, count);
    return launch_status();
}

// ---------------------------------------------------------------------------
// Wg
// ---------------------------------------------------------------------------
static int wg_target() {
    const int t = knobs().wg_target;     // 1024: 4 workgroups of the 128x128 shape per CU (round 2; was 1536 at 2 per CU)
    return t < 1 ? 1 : t;
}

// Split-K so that ~4 workgroups per CU are in flight: the wgrad loaders are gather-heavy and only
// reach the MFMA rate when several workgroups per SIMD overlap their load and MFMA phases.
static int wgrad_splits(long long tiles, int chunks, bool big_tile = true) {
    long long target = (big_tile ? wg_target() : 512) * cus() / 256;   // narrow tiles (3-channel layers) are slab-traffic bound
    if (tiles * 4 >= target * 3) return 1;
    long long want = (target + tiles - 1) / tiles;
    if (big_tile) {
        // ... but a 128x128 workgroup that reduces fewer than 32 chunks spends its time on the prologue and its 64 KB
        // slab: keep >= 32 chunks per split as long as >= 512 workgroups remain (bs 128, D.block1 / block2: 128 x 16
        // chunks -> 64 x 32, 0.111 -> 0.105 and 0.116 -> 0.110 ms; every bs 512 layer keeps its plan)
        const long long by_len = chunks / 32 > 0 ? chunks / 32 : 1, want_min = (2 * cus() + tiles - 1) / tiles;
        const long long floor_ = by_len > want_min ? by_len : want_min;
        if (floor_ < want) want = floor_;
    }
    long long cap = chunks / 8 > 0 ? chunks / 8 : 1;  // keep >= 8 chunks (128 pixels) per split
    long long s = want < cap ? want : cap;
    return (int)(s < 1 ? 1 : s);
}

// can a 16-pixel K chunk be taken as whole row segments of one image? (see WgBLoaderRow)
template <class G>
static bool wg_row_geom(const ConvShape& s, WgRowGeom* g) {
    int CW = s.OW < 16 ? s.OW : 16;
    if (CW <= 0 || 16 % CW) return false;
    int R = 16 / CW;
    if (s.OW % CW || s.OH % R) return false;
    auto mx = [](int a, int b) { return a > b ? a : b; };
    if (G::s * R < mx(G::p, G::kh - 1 - G::p) || G::s * CW < mx(G::p, G::kw - 1 - G::p)) return false;
    // a negative offset must never survive masking: the only negative cases are the flagged ones
    g->CW = CW;
    g->R = R;
    g->div_ohw = make_fastdiv(s.OH * s.OW);
    g->div_ow = make_fastdiv(s.OW);
    return true;
}

template <class G, class Cfg, class AL, class BL>
static int launch_wgrad(const typename AL::Params& pa, const typename BL::Params& pb, float* dw, float* ws,
                        size_t ws_bytes, const ConvShape& s, int KTOT, int NTOT, hipStream_t st) {
    long long tiles = (long long)((s.K + Cfg::BM - 1) / Cfg::BM) * ((NTOT + Cfg::BN - 1) / Cfg::BN);
    int chunks = (KTOT + BK - 1) / BK;
    int splits = wgrad_splits(tiles, chunks, Cfg::BM * Cfg::BN >= 128 * 128);
    long long count = (long long)s.K * NTOT;
    if (splits > 1) {
        long long max_splits = (long long)(ws_bytes / 4) / count;
        if (max_splits < 2) splits = 1;
        else if (splits > max_splits) splits = (int)max_splits;
    }
    // recompute the real number of z-slices launch_igemm will use
    int cps = (chunks + splits - 1) / splits;
    int nz = (chunks + cps - 1) / cps;
    float* out = nz > 1 ? ws : dw;
    EpiRowMajorB::Params pe{out, s.K, NTOT, NTOT, count, nullptr, ACT_NONE, 0.f};
    int rc = launch_igemm<Cfg, AL, BL, EpiRowMajorB>(pa, pb, pe, s.K, NTOT, KTOT, 1, splits, st);
    if (rc != GZ_OK) return rc;
    if (nz > 1) {
        if (defer_reduce(nz, count)) return rc;
        if (nz <= 8)
            hipLaunchKernelGGL(reduce_few_slabs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ws, dw,
                               nz, count);
        else
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64 * RS_WAVES), 0, st, ws,
                               dw, nz, count, count, (float*)nullptr, 0ll);
        rc = launch_status();
    }
    return rc;
}

// ---------------------------------------------------------------------------
// Wg of k4 s2 p1 layers with <= 4 channels on the image side (the critics' first convolution and -- as the adjoint --
// G's last transposed convolution): dw[k][c][ky][kx] = sum over (n, oy, ox) of y[n][k][oy][ox] *
// x[n][c][2 oy - 1 + ky][2 ox - 1 + kx], a K x (C * 16) result from a reduction over all N * OH * OW pixels.  On the
// implicit-GEMM skeleton that is ONE 64 x 64 or 128 x 64 tile split 512 ways: prologue, epilogue and a BK-chunk per
// workgroup (36 / 28 us at bs 128 for 5 / 9 us of HBM time).  Here v_mfma_f32_16x16x4_f32 runs with k = 4 pixels:
// A[channel k][pixel] comes from ONE 16-byte load per lane, 16-channel block and 16-pixel segment (component j of the
// vector feeds MFMA j: pixel ox0 + 4 q + j -- any assignment of pixels to k slots is as good as another, as long as B
// uses the same); B[pixel][(ky, kx)] = x[c][2 oy - 1 + ky][2 (ox0 + 4 q + j) - 1 + kx] is one dword load per lane, c
// and j (padding columns are out-of-range voffsets, padding rows wave-uniform).  A wavefront walks its share of the
// (n, oy, segment) items with the next item's loads in flight; the four wavefronts of a workgroup meet in an LDS tree
// and wavefront 0 writes the workgroup's slab (the same slabs gz_reduce_multi / the optimizer read).  In the step at
// bs 128 (inputs cold): D.conv_in 36 -> 27 us.  (Issuing the loads of four items at once measured 35 us.)
// ---------------------------------------------------------------------------
// FUSE (round 5, the first-order backward of `LeakyReLU(conv(x) + bias)`): the operand is the gradient with respect to
// the ACTIVATION's output, masked on load with the saved forward output (g * (out > 0 ? 1 : slope) -- the act_bwd
// launch and its write + re-read of the 34 MB gradient disappear), and a (C+1)-th column block multiplies it with a
// column of ones: D[k][0] = sum over the pixels = the bias gradient, which lands behind the K * C * 16 weight-gradient
// values of the workgroup's slab (a channel_sum launch and its second read of the gradient disappear).
template <int C, int KT, bool FUSE>      // image channels; 16-channel blocks on the feature side
__global__ __launch_bounds__(256) void wgrad_k4s2p1_fewc_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                const float* __restrict__ fwd_out, float* __restrict__ slab,
                                                                int N, int K, int H, int W, int OH, int OW, int items,
                                                                FastDiv div_seg, FastDiv div_oh, int act, float slope,
                                                                long long slab_stride) {
    constexpr int CB = FUSE ? C + 1 : C;
    __shared__ float part[2][KT * CB * 4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;
    const int ky = i >> 2, kx = i & 3;
    const int HW = H * W, OHW = OH * OW;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (uint32_t)N * C * HW * 4u);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(y, (uint32_t)N * K * OHW * 4u);
    const __amdgpu_buffer_rsrc_t ro = make_rsrc(FUSE ? fwd_out : y, (uint32_t)N * K * OHW * 4u);
    const float neg = act == ACT_RELU ? 0.f : slope;
    const float ones = i == 0 ? 1.f : 0.f;
    const int segs = OW >> 4;
    const int nwaves = gridDim.x * 4;
    auto fetch = [&](int g, f32x4 (&ya)[KT], f32x4 (&oa)[FUSE ? KT : 1], float (&xb)[C][4]) {
        const uint32_t rowid = fdiv((uint32_t)g, div_seg);               // n * OH + oy
        const int ox0 = (g - (int)rowid * segs) << 4;
        const uint32_t n = fdiv(rowid, div_oh);
        const int oy = (int)(rowid - n * (uint32_t)OH);
        const uint32_t vy = ((n * (uint32_t)K + i) * (uint32_t)OHW + (uint32_t)(oy * OW + ox0 + 4 * q)) * 4u;
#pragma unroll
        for (int a = 0; a < KT; ++a) {
            ya[a] = bload4(ry, vy, (uint32_t)(a * 16 * OHW) * 4u);
            if constexpr (FUSE) oa[a] = bload4(ro, vy, (uint32_t)(a * 16 * OHW) * 4u);
        }
        const int row = 2 * oy - 1 + ky;
        const bool rok = (unsigned)row < (unsigned)H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 2 * (ox0 + 4 * q + j) - 1 + kx;
            const uint32_t vx = (rok && (unsigned)col < (unsigned)W)
                                    ? (n * (uint32_t)(C * HW) + (uint32_t)(row * W + col)) * 4u : OOB;
#pragma unroll
            for (int c = 0; c < C; ++c) xb[c][j] = bload(rx, vx, (uint32_t)(c * HW) * 4u);
        }
    };
    f32x4 acc[KT][CB];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    int g = blockIdx.x * 4 + wave;
    if (g < items) {
        f32x4 yc[KT], yn[KT], oc[FUSE ? KT : 1], on[FUSE ? KT : 1];
        float xc[C][4], xn[C][4];
        fetch(g, yc, oc, xc);
        for (; g < items; g += nwaves) {
            const bool more = g + nwaves < items;
            if (more) fetch(g + nwaves, yn, on, xn);      // in flight during this item's 4 * KT * CB MFMAs
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < KT; ++a) {
                    float av = yc[a][j];
                    if constexpr (FUSE) av = oc[a][j] > 0.f ? av : av * neg;
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xc[c][j], acc[a][c], 0, 0, 0);
                    if constexpr (FUSE) acc[a][C] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, ones, acc[a][C], 0, 0, 0);
                }
            if (more) {
#pragma unroll
                for (int a = 0; a < KT; ++a) {
                    yc[a] = yn[a];
                    if constexpr (FUSE) oc[a] = on[a];
                }
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) xc[c][j] = xn[c][j];
            }
        }
    }
    // binary tree over the four wavefronts, fixed order
#pragma unroll
    for (int h = 2; h >= 1; h >>= 1) {
        if (wave >= h && wave < 2 * h) {
#pragma unroll
            for (int a = 0; a < KT; ++a)
#pragma unroll
                for (int c = 0; c < CB; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part[wave - h][(a * CB + c) * 4 + r][lane] = acc[a][c][r];
        }
        __syncthreads();
        if (wave < h) {
#pragma unroll
            for (int a = 0; a < KT; ++a)
#pragma unroll
                for (int c = 0; c < CB; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[a][c][r] += part[wave][(a * CB + c) * 4 + r][lane];
        }
        __syncthreads();
    }
    if (wave > 0) return;
    // D[row = channel 16 a + 4 q + r][column = (ky, kx)]  ->  dw[k][c][ky][kx]; the ones column -> dbias[k]
    float* out = slab + (long long)blockIdx.x * slab_stride;
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < C; ++c) out[((a * 16 + 4 * q + r) * C + c) * 16 + i] = acc[a][c][r];
            if constexpr (FUSE) {
                if (i == 0) out[K * C * 16 + a * 16 + 4 * q + r] = acc[a][C][r];
            }
        }
}

static bool wgrad_k4s2p1_fewc_ok(const ConvShape& s) {
    // (K = 128, G's last layer: 29 us against the tile path's 28 in the step at bs 128 -- not taken)
    return !knobs().no_fewc_wg && s.C <= 4 && (s.K == 16 || s.K == 32 || s.K == 64) &&
           s.H == 2 * s.OH && s.W == 2 * s.OW && s.OW % 16 == 0 && (long long)s.N * s.K * s.OH * s.OW < (1ll << 29) &&
           (long long)s.N * s.OH * (s.OW >> 4) >= 1024;
}

static int wgrad_k4s2p1_fewc_blocks(const ConvShape& s) {
    const long long items = (long long)s.N * s.OH * (s.OW >> 4);
    long long blocks = items / (4 * 4);                       // >= 4 items per wavefront
    const long long cap = knobs().fewc_wg_blocks;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

// fwd_out != nullptr: the fused form (y = gradient w.r.t. the activation output, masked with fwd_out; slab rows carry
// the K bias-gradient values behind the K * C * 16 weight-gradient values).  The fused form is always left unreduced
// (gz_conv2d_wgrad_act_partial).
static int run_wgrad_k4s2p1_fewc(const float* x, const float* y, const float* fwd_out, int act, float slope, float* dw,
                                 float* ws, size_t ws_bytes, const ConvShape& s, hipStream_t st) {
    const bool fuse = fwd_out != nullptr;
    const long long count = (long long)s.K * s.C * 16;
    const long long stride = count + (fuse ? s.K : 0);
    int blocks = wgrad_k4s2p1_fewc_blocks(s);
    const long long room = (long long)(ws_bytes / 4) / stride;
    if (room < 1 || !ws) return GZ_ERR_WORKSPACE;
    if (blocks > room) blocks = (int)room;
    const int items = s.N * s.OH * (s.OW >> 4);
    const FastDiv dseg = make_fastdiv(s.OW >> 4), doh = make_fastdiv(s.OH);
#define GZ_FEWC(C_, KT_, F_)                                                                                         \
    hipLaunchKernelGGL((wgrad_k4s2p1_fewc_kernel<C_, KT_, F_>), dim3((unsigned)blocks), dim3(256), 0, st, x, y, fwd_out, \
                       ws, s.N, s.K, s.H, s.W, s.OH, s.OW, items, dseg, doh, act, slope, stride)
#define GZ_FEWC_C(KT_, F_)                                                                                           \
    switch (s.C) {                                                                                                   \
        case 1: GZ_FEWC(1, KT_, F_); break;                                                                          \
        case 2: GZ_FEWC(2, KT_, F_); break;                                                                          \
        case 3: GZ_FEWC(3, KT_, F_); break;                                                                          \
        default: GZ_FEWC(4, KT_, F_);                                                                                \
    }
#define GZ_FEWC_K(F_)                                                                                                \
    switch (s.K / 16) {                                                                                              \
        case 1: GZ_FEWC_C(1, F_); break;                                                                             \
        case 2: GZ_FEWC_C(2, F_); break;                                                                             \
        default: GZ_FEWC_C(4, F_);                                                                                   \
    }
    if (fuse) { GZ_FEWC_K(true) } else { GZ_FEWC_K(false) }
#undef GZ_FEWC_K
#undef GZ_FEWC_C
#undef GZ_FEWC
    int rc = launch_status();
    if (rc != GZ_OK) return rc;
    if (defer_reduce(blocks, stride)) return rc;
    if (fuse) return GZ_ERR_UNSUPPORTED;
    if (blocks <= 8)
        hipLaunchKernelGGL(reduce_few_slabs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ws, dw, blocks,
                           count);
    else
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64 * RS_WAVES), 0, st, ws, dw,
                           blocks, count, count, (float*)nullptr, 0ll);
    return launch_status();
}

// Weight gradient on the igemm2 skeleton (register-staged transposing loaders, two LDS stages).  Tile 256 (output
// channels) x 128 ((c, ky, kx) columns); the reduction over the pixels is split so that >= 512 workgroups exist.
using Cfg2Wg = TileCfg2<2, 2, 2, 2>;

// K in [128, 256): the 128 x 256 tile (one row of wavefronts, all four along the columns)
static bool wgrad2_narrow(const ConvShape& s) { return s.K < 256; }

template <class G>
static constexpr bool wgrad2w_geom() {
    return (G::kh == 4 && G::kw == 4 && G::s == 2 && G::p == 1) || (G::kh == 5 && G::kw == 5 && G::s == 2 && G::p == 2) ||
           (G::kh == 3 && G::kw == 3 && G::s == 1 && G::p == 1);
}

template <class G>
static int wgrad2_splits(const ConvShape& s) {
    const bool off = knobs().no_igemm2 || knobs().no_igemm2_wg;
    WgRowGeom rg;
    const int NTOT = s.C * G::kh * G::kw;
    if (off || s.K < 128 || s.K % 32 || NTOT < 256 - 128 * !wgrad2_narrow(s) || !wg_row_geom<G>(s, &rg)) return 0;
    const long long tiles = wgrad2_narrow(s) ? (long long)((s.K + 127) / 128) * ((NTOT + 255) / 256)
                                             : (long long)((s.K + 255) / 256) * ((NTOT + 127) / 128);
    const int chunks = (s.N * s.OH * s.OW + BK - 1) / BK;
    // at most ONE round of the chip's 512 workgroup slots (two per CU): 18 tiles x 29 splits = 522 workgroups took two
    // rounds (3x3 s1 p1 256@32: 96 TFLOP/s; 28 splits: one round)
    int splits = (int)(2 * cus() / tiles);
    if (splits < 1) splits = 1;
    // >= 64 chunks per workgroup for the register-staged kernel; the LDS-DMA kernel's chunks cost nothing but their
    // MFMAs, so 32 are enough there (bs 128: D.block1-3's weight gradients move from the 128x128 kernel onto it)
    const int min_chunks_env = knobs().wg2_min_chunks;
    const bool dma = wgrad2w_geom<G>() && s.H == G::s * s.OH && s.W == G::s * s.OW &&
                     (s.OW == 4 || s.OW == 8 || s.OW % 16 == 0) && !knobs().no_igemm2w;
    const int min_chunks = min_chunks_env > 0 ? min_chunks_env : (dma ? 32 : 64);
    while (splits > 1 && chunks / splits < min_chunks) --splits;
    return tiles * splits >= cus() ? splits : 0;
}

template <class G, class Cfg>
static int run_wgrad2(const float* x, const float* y, float* dw, float* ws, size_t ws_bytes, const ConvShape& s,
                      int splits, hipStream_t st) {
    using AL = WgALoaderRow<Cfg::BM>;
    using BL = WgBLoaderRow<Cfg::BN, G::kh, G::kw, G::s, G::p>;
    const int KTOT = s.N * s.OH * s.OW;
    const int NTOT = s.C * G::kh * G::kw;
    WgRowGeom rg;
    wg_row_geom<G>(s, &rg);
    typename AL::Params pa{y, s, rg, KTOT};
    typename BL::Params pb{x, s, rg, KTOT, NTOT};
    const long long count = (long long)s.K * NTOT;
    if (splits > 1) {
        long long max_splits = (long long)(ws_bytes / 4) / count;
        if (max_splits < 2) splits = 1;
        else if (splits > max_splits) splits = (int)max_splits;
    }
    const int chunks = (KTOT + BK - 1) / BK;
    const int cps = (chunks + splits - 1) / splits;
    const int nz = (chunks + cps - 1) / cps;
    float* out = nz > 1 ? ws : dw;
    EpiRowMajorB::Params pe{out, s.K, NTOT, NTOT, count, nullptr, ACT_NONE, 0.f};
    int rc = launch_igemm2r<Cfg, AL, BL, EpiRowMajorB>(pa, pb, pe, s.K, NTOT, KTOT, splits, st);
    if (rc != GZ_OK) return rc;
    if (nz > 1) {
        if (defer_reduce(nz, count)) return rc;
        if (nz <= 8)
            hipLaunchKernelGGL(reduce_few_slabs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ws, dw,
                               nz, count);
        else
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64 * RS_WAVES), 0, st, ws,
                               dw, nz, count, count, (float*)nullptr, 0ll);
        rc = launch_status();
    }
    return rc;
}

// ... and with both operands by LDS-DMA (igemm2w_kernel: k4 s2 p1 only, pixel rows of 4, 8 or a multiple of 16)
template <class G>
static int wgrad2w_cw_shape(const ConvShape& s) {        // pixel-chunk width of the LDS-DMA weight gradient, 0 = not applicable
    const bool off = knobs().no_igemm2w;
    const bool off_g = knobs().no_igemm2wg;        // the generic-geometry image only
    if (off || !wgrad2w_geom<G>()) return 0;
    if (off_g && !(G::kh == 4 && G::kw == 4)) return 0;
    if (s.H != G::s * s.OH || s.W != G::s * s.OW || (s.W & 3)) return 0;
    const int cw = s.OW == 4 ? 4 : s.OW == 8 ? 8 : (s.OW % 16 == 0 ? 16 : 0);
    if (!cw || s.OH % (16 / cw)) return 0;
    return cw;
}

template <class G>
static int wgrad2w_cw(const float* x, const float* y, const ConvShape& s) {
    if ((((uintptr_t)x) | ((uintptr_t)y)) & 15) return 0;
    return wgrad2w_cw_shape<G>(s);
}

template <class G, int BN, int CW>
struct wg2w_image {
    using type = std::conditional_t<G::kh == 4 && G::kw == 4, WgImgB2<BN, CW>, WgImgBG<BN, CW, G::kh, G::kw, G::s, G::p>>;
};

template <class G, class Cfg, int CW>
static int run_wgrad2w(const float* x, const float* y, float* dw, float* ws, size_t ws_bytes, const ConvShape& s,
                       int splits, hipStream_t st) {
    using BL = typename wg2w_image<G, Cfg::BN, CW>::type;
    const int KTOT = s.N * s.OH * s.OW;
    const int NTOT = s.C * G::kh * G::kw;
    Wg2Params p{x, y, s, make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW)};
    const long long count = (long long)s.K * NTOT;
    if (splits > 1) {
        long long max_splits = (long long)(ws_bytes / 4) / count;
        if (max_splits < 2) splits = 1;
        else if (splits > max_splits) splits = (int)max_splits;
    }
    const int chunks = (KTOT + BK - 1) / BK;
    const int cps = (chunks + splits - 1) / splits;
    const int nz = (chunks + cps - 1) / cps;
    float* out = nz > 1 ? ws : dw;
    EpiRowMajorB::Params pe{out, s.K, NTOT, NTOT, count, nullptr, ACT_NONE, 0.f};
    int rc = launch_igemm2w<Cfg, BL, EpiRowMajorB>(p, pe, s.K, NTOT, KTOT, splits, st);
    if (rc != GZ_OK) return rc;
    if (nz > 1) {
        if (defer_reduce(nz, count)) return rc;
        if (nz <= 8)
            hipLaunchKernelGGL(reduce_few_slabs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ws, dw,
                               nz, count);
        else
            hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64 * RS_WAVES), 0, st, ws,
                               dw, nz, count, count, (float*)nullptr, 0ll);
        rc = launch_status();
    }
    return rc;
}

template <class G, class Cfg>
static int run_wgrad2w_cw(int cw, const float* x, const float* y, float* dw, float* ws, size_t ws_bytes,
                          const ConvShape& s, int splits, hipStream_t st) {
    if constexpr (wgrad2w_geom<G>()) {
        switch (cw) {
            case 4: return run_wgrad2w<G, Cfg, 4>(x, y, dw, ws, ws_bytes, s, splits, st);
            case 8: return run_wgrad2w<G, Cfg, 8>(x, y, dw, ws, ws_bytes, s, splits, st);
            default: return run_wgrad2w<G, Cfg, 16>(x, y, dw, ws, ws_bytes, s, splits, st);
        }
    }
    return GZ_ERR_UNSUPPORTED;
}

template <class G, class Cfg>
static int run_wgrad(const float* x, const float* y, float* dw, float* ws, size_t ws_bytes, const ConvShape& s,
                     hipStream_t st) {
    const int KTOT = s.N * s.OH * s.OW;
    const int NTOT = s.C * G::kh * G::kw;
    WgRowGeom rg;
    const bool generic_only = knobs().wg_generic;
    if (!generic_only && wg_row_geom<G>(s, &rg)) {
        using AL = WgALoaderRow<Cfg::BM>;
        using BL = WgBLoaderRow<Cfg::BN, G::kh, G::kw, G::s, G::p>;
        typename AL::Params pa{y, s, rg, KTOT};
        typename BL::Params pb{x, s, rg, KTOT, NTOT};
        return launch_wgrad<G, Cfg, AL, BL>(pa, pb, dw, ws, ws_bytes, s, KTOT, NTOT, st);
    }
    using AL = WgALoader<Cfg::BM>;
    using BL = WgBLoader<Cfg::BN, G::kh, G::kw, G::s, G::p>;
    typename AL::Params pa{y, s, make_fastdiv(s.OH * s.OW), KTOT};
    typename BL::Params pb{x, s, make_fastdiv(s.OH * s.OW), make_fastdiv(s.OW), KTOT, NTOT};
    return launch_wgrad<G, Cfg, AL, BL>(pa, pb, dw, ws, ws_bytes, s, KTOT, NTOT, st);
}

extern "C" int gz_conv2d_tile(int op, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S);

template <class G>
static int dispatch_wgrad(const float* x, const float* y, float* dw, float* ws, size_t ws_bytes,
                          const ConvShape& s, hipStream_t st) {
    TileId t = (TileId)gz_conv2d_tile(2, s.N, s.C, s.H, s.W, s.K, s.OH, s.OW, G::kh, G::kw, G::s);
    if (t == T256x128 || t == T128x256) {
        const int splits = wgrad2_splits<G>(s);
        if (splits > 0) {
            const int cw = wgrad2w_cw<G>(x, y, s);
            if (cw)
                return wgrad2_narrow(s) ? run_wgrad2w_cw<G, Cfg128x256>(cw, x, y, dw, ws, ws_bytes, s, splits, st)
                                        : run_wgrad2w_cw<G, Cfg2Wg>(cw, x, y, dw, ws, ws_bytes, s, splits, st);
            if constexpr (G::kh == 4 && G::kw == 4)       // unaligned tensors: the register-staged loaders
                return wgrad2_narrow(s) ? run_wgrad2<G, Cfg128x256>(x, y, dw, ws, ws_bytes, s, splits, st)
                                        : run_wgrad2<G, Cfg2Wg>(x, y, dw, ws, ws_bytes, s, splits, st);
        }
        t = T128x128;
    }
    switch (t) {
        case T128x128: return run_wgrad<G, Cfg128x128>(x, y, dw, ws, ws_bytes, s, st);
        case T128x64: return run_wgrad<G, Cfg128x64>(x, y, dw, ws, ws_bytes, s, st);
        case T128x32: return run_wgrad<G, Cfg128x32>(x, y, dw, ws, ws_bytes, s, st);
        default: return run_wgrad<G, Cfg64x64>(x, y, dw, ws, ws_bytes, s, st);
    }
}

// ---------------------------------------------------------------------------
// plain GEMM  C[M][N] = op(A) . op(B)  (+bias[n], activation)
// ---------------------------------------------------------------------------
template <class Cfg, class AL, class BL>
static int run_gemm(const float* a, const float* b, const float* bias, float* c, int M, int N, int K, int lda,
                    int ldb, int ldc, int act, float slope, hipStream_t st, int splits, float* slab) {
    typename AL::Params pa{a, K, M, lda, 0};
    typename BL::Params pb{b, K, N, ldb, 0};
    EpiRowMajor::Params pe{c, M, N, ldc, 0, bias, act, slope};
    return launch_igemm<Cfg, AL, BL, EpiRowMajor>(pa, pb, pe, M, N, K, 1, splits, st, slab);
}

template <class Cfg>
static int gemm_ops(const float* a, const float* b, const float* bias, float* c, int M, int N, int K, int lda,
                    int ldb, int ldc, int ta, int tb, int act, float slope, hipStream_t st, int splits, float* slab) {
    // ta == 0: A is [M][K] row-major (k contiguous);  ta == 1: A is stored [K][M] (m contiguous)
    // tb == 0: B is [K][N] row-major (n contiguous);  tb == 1: B is stored [N][K] (k contiguous)
    const bool b_vec = !tb && (ldb % 4 == 0) && (N % 4 == 0) && (((uintptr_t)b & 15) == 0);
    if (!ta && !tb) {
        if (b_vec) return run_gemm<Cfg, KContigLoader<Cfg::BM>, MContigLoader4<Cfg::BN>>(a, b, bias, c, M, N, K, lda, ldb, ldc, act, slope, st, splits, slab);
        return run_gemm<Cfg, KContigLoader<Cfg::BM>, MContigLoader<Cfg::BN>>(a, b, bias, c, M, N, K, lda, ldb, ldc, act, slope, st, splits, slab);
    }
    if (ta && !tb) {
        if (b_vec) return run_gemm<Cfg, MContigLoader<Cfg::BM>, MContigLoader4<Cfg::BN>>(a, b, bias, c, M, N, K, lda, ldb, ldc, act, slope, st, splits, slab);
        return run_gemm<Cfg, MContigLoader<Cfg::BM>, MContigLoader<Cfg::BN>>(a, b, bias, c, M, N, K, lda, ldb, ldc, act, slope, st, splits, slab);
    }
    if (!ta && tb) return run_gemm<Cfg, KContigLoader<Cfg::BM>, KContigLoader<Cfg::BN>>(a, b, bias, c, M, N, K, lda, ldb, ldc, act, slope, st, splits, slab);
    return run_gemm<Cfg, MContigLoader<Cfg::BM>, KContigLoader<Cfg::BN>>(a, b, bias, c, M, N, K, lda, ldb, ldc, act, slope, st, splits, slab);
}

}  // namespace gz

using namespace gz;

typedef Geo<4, 4, 2, 1> G4421;
typedef Geo<5, 5, 2, 2> G5522;
typedef Geo<3, 3, 1, 1> G3311;
typedef Geo<1, 1, 1, 0> G1110;

#define GZ_GEOM_DISPATCH_OR(CALL, ELSE)                                 \
    if (KH == 4 && KW == 4 && S == 2 && P == 1) return CALL(G4421);     \
    if (KH == 5 && KW == 5 && S == 2 && P == 2) return CALL(G5522);     \
    if (KH == 3 && KW == 3 && S == 1 && P == 1) return CALL(G3311);     \
    if (KH == 1 && KW == 1 && S == 1 && P == 0) return CALL(G1110);     \
    return ELSE;
#define GZ_GEOM_DISPATCH(CALL) GZ_GEOM_DISPATCH_OR(CALL, GZ_ERR_UNSUPPORTED)

extern "C" {

long long gz_conv2d_pack_fwd_elems(int K, int C, int KH, int KW) {
    return (long long)(fwd_tap_major(C, KH, KW) ? round_bk(C) : C) * KH * KW * round4(K);
}

long long gz_conv2d_pack_dgrad_elems(int K, int C, int KH, int KW, int S) {
    int TY = (KH + S - 1) / S, TX = (KW + S - 1) / S;
    return (long long)S * S * (dgrad_tap_major(K, KH, KW, S) ? round_bk(K) : K) * TY * TX * round4(C);
}

int gz_conv2d_pack_fwd(const float* w, float* wp, int K, int C, int KH, int KW, hipStream_t stream) {
    gz::clear_stale_error();
    if (K <= 0 || C <= 0 || KH <= 0 || KW <= 0) return GZ_ERR_BAD_SHAPE;
    int Kg = C * KH * KW, ld = round4(K);
    if (fwd_tap_major(C, KH, KW)) {
        long long total = (long long)KH * KW * round_bk(C) * ld;
        hipLaunchKernelGGL(pack_fwd_tap_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)),
                           dim3(256), 0, stream, w, wp, K, C, KH * KW, round_bk(C), ld);
        return launch_status();
    }
    dim3 grid((Kg + 31) / 32, (ld + 31) / 32);
    hipLaunchKernelGGL(transpose_pad_kernel, grid, dim3(256), 0, stream, w, wp, K, Kg, ld);
    return launch_status();
}

int gz_conv2d_pack_dgrad(const float* w, float* wp, int K, int C, int KH, int KW, int S, int P,
                         hipStream_t stream) {
    gz::clear_stale_error();
    if (K <= 0 || C <= 0 || KH <= 0 || KW <= 0 || S <= 0) return GZ_ERR_BAD_SHAPE;
    int TY = (KH + S - 1) / S, TX = (KW + S - 1) / S;
    if (dgrad_tap_major(K, KH, KW, S)) {
        long long total = (long long)TY * TX * round_bk(K) * round4(C);
        unsigned bx = (unsigned)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
        hipLaunchKernelGGL(pack_dgrad_tap_kernel, dim3(bx, S * S), dim3(256), 0, stream, w, wp, K, C, KH, KW, S, P, TY,
                           TX, round_bk(K), round4(C));
        return launch_status();
    }
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(K, S * S), dim3(256), 0, stream, w, wp, K, C, KH, KW, S, P, TY, TX,
                       round4(C));
    return launch_status();
}

size_t gz_conv2d_pack_job_bytes(void) { return sizeof(PackJob); }

int gz_conv2d_pack_job(void* job_out, const float* w, float* wp, int is_dgrad, int K, int C, int KH, int KW, int S, int P,
                       int block0) {
    if (!job_out || K <= 0 || C <= 0 || KH <= 0 || KW <= 0 || S <= 0) return GZ_ERR_BAD_SHAPE;
    PackJob jb{w, wp, 0, K, C, KH, KW, S, P, 1, 1, block0};
    const int TY = (KH + S - 1) / S, TX = (KW + S - 1) / S;
    if (!is_dgrad) {
        if (fwd_tap_major(C, KH, KW)) {
            const long long total = (long long)KH * KW * round_bk(C) * round4(K);
            jb.kind = 1;
            jb.gx = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        } else {
            jb.kind = 0;
            jb.gx = (C * KH * KW + 31) / 32;
            jb.gy = (round4(K) + 31) / 32;
        }
    } else if (dgrad_tap_major(K, KH, KW, S)) {
        const long long total = (long long)TY * TX * round_bk(K) * round4(C);
        jb.kind = 3;
        jb.gx = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
        jb.gy = S * S;
    } else if (KH == 4 && KW == 4 && S == 2 && P == 1 && !knobs().no_pack_k4 && (((uintptr_t)w | (uintptr_t)wp) & 15) == 0) {
        jb.kind = 5;          // the whole ko in one workgroup, contiguous reads (pack_dgrad_k4_body)
        jb.gx = K;
        jb.gy = 1;
    } else {
        jb.kind = 2;
        jb.gx = K;
        jb.gy = S * S;
    }
    *reinterpret_cast<PackJob*>(job_out) = jb;
    return jb.gx * jb.gy;
}

int gz_conv2d_pack_multi(const void* jobs_dev, int njobs, int total_blocks, hipStream_t stream) {
    gz::clear_stale_error();
    if (!jobs_dev || njobs <= 0 || total_blocks <= 0) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(pack_multi_kernel, dim3(total_blocks), dim3(256), 0, stream,
                       reinterpret_cast<const PackJob*>(jobs_dev), njobs);
    return launch_status();
}

int gz_conv2d_pack_table_max_jobs(void) { return PACK_TABLE_MAX; }
size_t gz_conv2d_pack_table_bytes(void) { return sizeof(PackTable); }

/* what: 0 the forward image, 1 the dgrad image, 2 a plain copy (w * scale in w's own layout) */
int gz_conv2d_pack_table_add(void* table_host, const float* w, float* wp, const float* sigma, int what, int K, int C, int KH,
                             int KW, int S, int P) {
    PackTable* t = reinterpret_cast<PackTable*>(table_host);
    if (!t || !w || !wp || what < 0 || what > 2) return GZ_ERR_BAD_SHAPE;
    if (t->njobs < 0 || t->njobs >= PACK_TABLE_MAX) return GZ_ERR_UNSUPPORTED;
    PackJob jb;
    if (what == 2) {
        if (K <= 0 || C <= 0 || KH <= 0 || KW <= 0) return GZ_ERR_BAD_SHAPE;
        const long long total = (long long)K * C * KH * KW;
        jb = PackJob{w, wp, 4, K, C, KH, KW, 1, 0, (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256), 1, 0};
    } else {
        const int n = gz_conv2d_pack_job(&jb, w, wp, what, K, C, KH, KW, S, P, 0);
        if (n < 0) return n;
    }
    t->jobs[t->njobs] = jb;
    t->sigma[t->njobs] = sigma;
    ++t->njobs;
    return GZ_OK;
}

int gz_conv2d_pack_table_launch(void* table_host, hipStream_t stream) {
    gz::clear_stale_error();
    PackTable* t = reinterpret_cast<PackTable*>(table_host);
    if (!t || t->njobs <= 0 || t->njobs > PACK_TABLE_MAX) return GZ_ERR_BAD_SHAPE;
    long long blocks = 0;
    for (int j = 0; j < t->njobs; ++j) {
        t->jobs[j].block0 = (int)blocks;
        blocks += (long long)t->jobs[j].gx * t->jobs[j].gy;
    }
    if (blocks <= 0 || blocks >= (1ll << 31)) return GZ_ERR_TOO_LARGE;
    hipLaunchKernelGGL(pack_table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, *t);
    return launch_status();
}

size_t gz_conv2d_fwd_workspace_bytes(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P) {
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P)) return 0;
#define CALL(G) fwd_ws_bytes<G>(s)
    GZ_GEOM_DISPATCH_OR(CALL, 0)
#undef CALL
}

size_t gz_conv2d_dgrad_workspace_bytes(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S,
                                       int P) {
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P)) return 0;
#define CALL(G) dgrad_ws_bytes<G>(s)
    GZ_GEOM_DISPATCH_OR(CALL, 0)
#undef CALL
}

int gz_conv2d_fwd(const float* x, const float* wpack, const float* bias, float* y, float* workspace, size_t ws_bytes,
                  int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, int act,
                  float slope, hipStream_t stream) {
    gz::clear_stale_error();
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P)) return GZ_ERR_BAD_SHAPE;
    if (too_large((long long)N * C * H * W) || too_large((long long)N * K * OH * OW)) return GZ_ERR_TOO_LARGE;
    if (((uintptr_t)wpack & 15) || ((uintptr_t)y & 15)) return GZ_ERR_BAD_SHAPE;
    if (KH == 3 && KW == 3 && S == 1 && P == 1 && conv3_smallch_ok(N, C, K, H, W))
        return run_conv3_smallch(x, wpack, bias, y, N, C, K, H, W, fwd_tap_major(C, 3, 3), 0, act, slope, stream);
#define CALL(G) dispatch_fwd<G>(x, wpack, bias, y, s, act, slope, workspace, ws_bytes, stream)
    GZ_GEOM_DISPATCH(CALL)
#undef CALL
}

int gz_conv2d_dgrad(const float* y, const float* wpack, const float* bias, float* x, float* workspace,
                    size_t ws_bytes, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P,
                    int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P)) return GZ_ERR_BAD_SHAPE;
    if (too_large((long long)N * C * H * W) || too_large((long long)N * K * OH * OW)) return GZ_ERR_TOO_LARGE;
    if ((uintptr_t)wpack & 15) return GZ_ERR_BAD_SHAPE;
    if (KH == 3 && KW == 3 && S == 1 && P == 1 && conv3_smallch_ok(N, K, C, H, W))
        return run_conv3_smallch(y, wpack, bias, x, N, K, C, H, W, dgrad_tap_major(K, 3, 3, 1), 1, act, slope, stream);
#define CALL(G) dispatch_dgrad<G>(y, wpack, bias, x, s, act, slope, workspace, ws_bytes, stream)
    GZ_GEOM_DISPATCH(CALL)
#undef CALL
}

size_t gz_conv2d_wgrad_workspace_bytes(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW) {
    long long count = (long long)K * C * KH * KW;
    ConvShape s{N, C, H, W, K, OH, OW};
    if (OH == H && OW == W && wgrad_smallch_ok(s, KH, KW, 1, 1)) return (size_t)wgrad_smallch_blocks(s) * (count + K) * 4;
    size_t fewc = 0;
    if (KH == 4 && KW == 4 && wgrad_k4s2p1_fewc_ok(s)) fewc = (size_t)wgrad_k4s2p1_fewc_blocks(s) * (count + K) * 4;
    int chunks = (N * OH * OW + BK - 1) / BK;
    // upper bound over the tile choices: smallest tile count is with 128x128 tiles
    long long tiles = (long long)((K + 127) / 128) * ((C * KH * KW + 127) / 128);
    int splits = wgrad_splits(tiles, chunks);
    {                                    // the igemm2 plans may split further
        const int s2 = (KH == 4 && KW == 4) ? wgrad2_splits<G4421>(s)
                       : (KH == 5 && KW == 5 && OH * 2 == H) ? wgrad2_splits<G5522>(s)
                       : (KH == 3 && KW == 3 && OH == H) ? wgrad2_splits<G3311>(s) : 0;
        if (s2 > splits) splits = s2;
    }
    const size_t generic = splits > 1 ? (size_t)splits * count * 4 : 0;
    return generic > fewc ? generic : fewc;
}

int gz_conv2d_wgrad_fuses_bias(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P) {
    ConvShape s{N, C, H, W, K, OH, OW};
    return shape_ok(s, KH, KW, S, P) && wgrad_smallch_ok(s, KH, KW, S, P) ? 1 : 0;
}

int gz_conv2d_wgrad(const float* x, const float* y, float* dw, float* dbias, float* workspace, size_t ws_bytes, int N,
                    int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, hipStream_t stream) {
    gz::clear_stale_error();
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P)) return GZ_ERR_BAD_SHAPE;
    if (too_large((long long)N * C * H * W) || too_large((long long)N * K * OH * OW)) return GZ_ERR_TOO_LARGE;
    if (wgrad_smallch_ok(s, KH, KW, S, P)) return run_wgrad_smallch(x, y, dw, dbias, workspace, ws_bytes, s, stream);
    if (dbias) return GZ_ERR_UNSUPPORTED;       // ask gz_conv2d_wgrad_fuses_bias first
    if (KH == 4 && KW == 4 && S == 2 && P == 1 && wgrad_k4s2p1_fewc_ok(s) && workspace &&
        ws_bytes >= (size_t)K * C * 16 * 4)
        return run_wgrad_k4s2p1_fewc(x, y, nullptr, ACT_NONE, 0.f, dw, workspace, ws_bytes, s, stream);
#define CALL(G) dispatch_wgrad<G>(x, y, dw, workspace, ws_bytes, s, stream)
    GZ_GEOM_DISPATCH(CALL)
#undef CALL
}

int gz_conv2d_wgrad_partial(const float* x, const float* y, float* dw, float* workspace, size_t ws_bytes, int N, int C,
                            int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, int* nz_out,
                            long long* stride_out, hipStream_t stream) {
    gz::clear_stale_error();
    if (!nz_out || !stride_out) return GZ_ERR_BAD_SHAPE;
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P)) return GZ_ERR_BAD_SHAPE;
    if (too_large((long long)N * C * H * W) || too_large((long long)N * K * OH * OW)) return GZ_ERR_TOO_LARGE;
    *nz_out = 1;
    *stride_out = (long long)K * C * KH * KW;
    if (wgrad_smallch_ok(s, KH, KW, S, P)) return run_wgrad_smallch(x, y, dw, nullptr, workspace, ws_bytes, s, stream);
    WgDefer d{1, *stride_out};
    tl_wg_defer = &d;
    if (KH == 4 && KW == 4 && S == 2 && P == 1 && wgrad_k4s2p1_fewc_ok(s) && workspace &&
        ws_bytes >= (size_t)K * C * 16 * 4) {
        const int rc = run_wgrad_k4s2p1_fewc(x, y, nullptr, ACT_NONE, 0.f, dw, workspace, ws_bytes, s, stream);
        tl_wg_defer = nullptr;
        *nz_out = d.nz;
        *stride_out = d.stride;
        return rc;
    }
#define CALL(G) dispatch_wgrad<G>(x, y, dw, workspace, ws_bytes, s, stream)
    const int rc = [&]() -> int { GZ_GEOM_DISPATCH(CALL) }();
#undef CALL
    tl_wg_defer = nullptr;
    *nz_out = d.nz;
    *stride_out = d.stride;
    return rc;
}

static bool dgrad_act_direct(const ConvShape& s, int KH, int KW, int S, int P, int act) {
    return KH == 4 && KW == 4 && S == 2 && P == 1 && (act == ACT_RELU || act == ACT_LRELU) && !knobs().smallc_one_pos &&
           !knobs().no_act_fuse && dgrad_direct<G4421>(nullptr, s) && s.OW % 4 == 0 && 64 % (s.OW / 4) == 0;
}

int gz_conv2d_dgrad_act_fuses(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, int act) {
    ConvShape s{N, C, H, W, K, OH, OW};
    return shape_ok(s, KH, KW, S, P) && dgrad_act_direct(s, KH, KW, S, P, act) ? 1 : 0;
}

int gz_conv2d_dgrad_act(const float* gy, const float* fwd_out, int act, float slope, const float* wpack, float* x, int N,
                        int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, hipStream_t stream) {
    gz::clear_stale_error();
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!gy || !fwd_out || !wpack || !x || !shape_ok(s, KH, KW, S, P)) return GZ_ERR_BAD_SHAPE;
    if (too_large((long long)N * C * H * W) || too_large((long long)N * K * OH * OW)) return GZ_ERR_TOO_LARGE;
    if (!dgrad_act_direct(s, KH, KW, S, P, act) || (((uintptr_t)gy | (uintptr_t)x | (uintptr_t)fwd_out | (uintptr_t)wpack) & 15))
        return GZ_ERR_UNSUPPORTED;
    const float neg = act == ACT_RELU ? 0.f : slope;
    switch (C) {
        case 1: return run_dgrad_smallc<1>(gy, wpack, nullptr, x, s, ACT_NONE, 0.f, stream, fwd_out, neg);
        case 2: return run_dgrad_smallc<2>(gy, wpack, nullptr, x, s, ACT_NONE, 0.f, stream, fwd_out, neg);
        case 3: return run_dgrad_smallc<3>(gy, wpack, nullptr, x, s, ACT_NONE, 0.f, stream, fwd_out, neg);
        default: return run_dgrad_smallc<4>(gy, wpack, nullptr, x, s, ACT_NONE, 0.f, stream, fwd_out, neg);
    }
}

int gz_conv2d_wgrad_act_fuses(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, int act) {
    ConvShape s{N, C, H, W, K, OH, OW};
    return shape_ok(s, KH, KW, S, P) && KH == 4 && KW == 4 && S == 2 && P == 1 && (act == ACT_RELU || act == ACT_LRELU) &&
                   !knobs().no_act_fuse && wgrad_k4s2p1_fewc_ok(s) ? 1 : 0;
}

int gz_conv2d_wgrad_act_partial(const float* x, const float* gy, const float* fwd_out, int act, float slope,
                                float* workspace, size_t ws_bytes, int N, int C, int H, int W, int K, int OH, int OW,
                                int KH, int KW, int S, int P, int* nz_out, long long* stride_out,
                                long long* bias_offset_out, hipStream_t stream) {
    gz::clear_stale_error();
    if (!nz_out || !stride_out || !bias_offset_out || !x || !gy || !fwd_out) return GZ_ERR_BAD_SHAPE;
    if (!gz_conv2d_wgrad_act_fuses(N, C, H, W, K, OH, OW, KH, KW, S, P, act)) return GZ_ERR_UNSUPPORTED;
    ConvShape s{N, C, H, W, K, OH, OW};
    if (too_large((long long)N * C * H * W) || too_large((long long)N * K * OH * OW)) return GZ_ERR_TOO_LARGE;
    WgDefer d{1, 0};
    tl_wg_defer = &d;
    const int rc = run_wgrad_k4s2p1_fewc(x, gy, fwd_out, act, slope, nullptr, workspace, ws_bytes, s, stream);
    tl_wg_defer = nullptr;
    *nz_out = d.nz;
    *stride_out = d.stride;
    *bias_offset_out = (long long)K * C * 16;
    return rc;
}

int gz_set_cu_budget(int cu_count) {
    if (cu_count <= 0) cu_count = 256;
    if (cu_count < 64) cu_count = 64;
    if (cu_count > 256) cu_count = 256;
    gz::cu_budget_ref().store(cu_count, std::memory_order_relaxed);
    return cu_count;
}

int gz_get_cu_budget(void) { return gz::cus(); }

int gz_reduce_multi_max_jobs(void) { return REDUCE_MAX_JOBS; }
int gz_reduce_multi_max_sources(void) { return REDUCE_MAX_SRC; }
size_t gz_reduce_multi_table_bytes(void) { return sizeof(ReduceTable); }

/* table_host: a ReduceTable filled through gz_reduce_multi_add (host memory; copied into the kernel argument) */
int gz_reduce_multi_add(void* table_host, float* out, long long count, int beta, const float* slabs, int nz,
                        long long stride) {
    ReduceTable* t = reinterpret_cast<ReduceTable*>(table_host);
    // slabs == NULL, nz == 0: a contribution that is exactly ZERO (a bias in front of a normalisation over its own
    // plane): the job exists -- with beta = 0 the gradient is written as zeros by the same launch that sums the others,
    // instead of a fill launch per such parameter -- but reads nothing
    const bool zero_src = !slabs && nz == 0;
    if (!t || !out || (!slabs && !zero_src) || count <= 0 || (count & 3) || (stride & 3) || (!zero_src && nz < 1) ||
        (((uintptr_t)out | (uintptr_t)slabs) & 15))
        return GZ_ERR_BAD_SHAPE;
    for (int j = 0; j < t->njobs; ++j)
        if (t->jobs[j].out == out) {            // another contribution to the same gradient
            ReduceJob& jb = t->jobs[j];
            if (zero_src) return jb.count == count ? GZ_OK : GZ_ERR_UNSUPPORTED;
            if (jb.count != count || jb.nsrc >= REDUCE_MAX_SRC) return GZ_ERR_UNSUPPORTED;
            jb.src[jb.nsrc++] = ReduceSrc{slabs, stride, nz, 0};
            return GZ_OK;
        }
    if (t->njobs >= REDUCE_MAX_JOBS) return GZ_ERR_UNSUPPORTED;
    ReduceJob& jb = t->jobs[t->njobs++];
    jb.out = out;
    jb.count = count;
    jb.beta = beta ? 1 : 0;
    jb.nsrc = zero_src ? 0 : 1;
    jb.block0 = 0;
    jb.src[0] = ReduceSrc{slabs, stride, nz, 0};
    return GZ_OK;
}

int gz_reduce_multi(void* table_host, hipStream_t stream) {
    gz::clear_stale_error();
    ReduceTable* t = reinterpret_cast<ReduceTable*>(table_host);
    if (!t || t->njobs <= 0 || t->njobs > REDUCE_MAX_JOBS) return GZ_ERR_BAD_SHAPE;
    long long blocks = 0;
    for (int j = 0; j < t->njobs; ++j) {
        t->jobs[j].block0 = (int)blocks;
        blocks += (t->jobs[j].count + 255) / 256;
    }
    if (blocks <= 0 || blocks >= (1ll << 31)) return GZ_ERR_TOO_LARGE;
    hipLaunchKernelGGL(reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, *t);
    return launch_status();
}

// ---- convolution + BatchNorm statistics in one launch --------------------------------------------------------
static int stats_wm(TileId t) { return t == T128x32 ? 4 : 2; }
static int stats_tm_rows(TileId t, long long M) {       // partial rows per phase: tiles_m * WM
    if (t == T256x256 || t == T256x128) return (int)((M + 255) / 256) * 2;
    if (t == T512x64) return (int)((M + 511) / 512) * 4;
    if (t == T256x64) return (int)((M + 255) / 256) * 2;
    const int bm = t == T64x64 ? 64 : 128;
    return (int)((M + bm - 1) / bm) * stats_wm(t);
}

int gz_conv2d_fwd_stats_rows(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P) {
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P)) return 0;
#define CALL(G) fwd_plan<G>(s)
    SplitPlan sp = [&]() -> SplitPlan { GZ_GEOM_DISPATCH_OR(CALL, (SplitPlan{T64x64, 2})) }();
#undef CALL
    // split launches (round 4): splitk_finish_kernel runs the same epilogue per 32 x 32 output block and writes the
    // statistics there -- one partial row per 32 pixels
    if (sp.splits > 1) return (int)(((long long)N * OH * OW + 31) / 32);
    return stats_tm_rows(sp.tile, (long long)N * OH * OW);      // (the tap-major gather launches do carry them)
}

int gz_conv2d_fwd_stats_ws(const float* x, const float* wpack, float* y, float* stats, float* workspace, size_t ws_bytes,
                           int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P,
                           hipStream_t stream) {
    gz::clear_stale_error();
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P) || !stats) return GZ_ERR_BAD_SHAPE;
    if (gz_conv2d_fwd_stats_rows(N, C, H, W, K, OH, OW, KH, KW, S, P) <= 0) return GZ_ERR_UNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) != 0) return GZ_ERR_BAD_SHAPE;      // 16-byte LDS-DMA pieces / row stores
    if (too_large((long long)N * C * H * W) || too_large((long long)N * K * OH * OW)) return GZ_ERR_TOO_LARGE;
#define CALL(G)                                                                                                      \
    [&]() -> int {                                                                                                   \
        const SplitPlan sp = fwd_plan<G>(s);                                                                         \
        if (sp.splits > 1 && (!workspace || ws_bytes < fwd_ws_bytes<G>(s))) return GZ_ERR_WORKSPACE;                 \
        float* slab = sp.splits > 1 ? workspace : nullptr;                                                           \
        switch (sp.tile) {                                                                                           \
            case T256x256: return run_fwd2_ow<Cfg256x256>(x, wpack, nullptr, y, s, 0, 0.f, stream, sp.splits, slab, stats); \
            case T256x64:                                                                                            \
                if (!fwd2_ok<G>(s)) return run_fwdtap2<G, Cfg256x64>(x, wpack, nullptr, y, s, 0, 0.f, stream, sp.splits, slab, stats); \
                return run_fwd2_ow<Cfg256x64>(x, wpack, nullptr, y, s, 0, 0.f, stream, sp.splits, slab, stats);      \
            case T256x128:                                                                                           \
                if (!fwd2_ok<G>(s)) return run_fwdtap2<G, Cfg256x128>(x, wpack, nullptr, y, s, 0, 0.f, stream, sp.splits, slab, stats); \
                return run_fwd2_ow<Cfg256x128>(x, wpack, nullptr, y, s, 0, 0.f, stream, sp.splits, slab, stats);     \
            case T128x128: return run_fwd<G, Cfg128x128>(x, wpack, nullptr, y, s, 0, 0.f, stream, sp.splits, slab, stats); \
            case T128x64: return run_fwd<G, Cfg128x64>(x, wpack, nullptr, y, s, 0, 0.f, stream, sp.splits, slab, stats);   \
            case T128x32: return run_fwd<G, Cfg128x32>(x, wpack, nullptr, y, s, 0, 0.f, stream, sp.splits, slab, stats);   \
            default: return run_fwd<G, Cfg64x64>(x, wpack, nullptr, y, s, 0, 0.f, stream, sp.splits, slab, stats);         \
        }                                                                                                            \
    }()
    GZ_GEOM_DISPATCH(CALL)
#undef CALL
}

int gz_conv2d_fwd_stats(const float* x, const float* wpack, float* y, float* stats, int N, int C, int H, int W, int K,
                        int OH, int OW, int KH, int KW, int S, int P, hipStream_t stream) {
    return gz_conv2d_fwd_stats_ws(x, wpack, y, stats, nullptr, 0, N, C, H, W, K, OH, OW, KH, KW, S, P, stream);
}

int gz_conv2d_dgrad_stats_rows(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P) {
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P) || H % S || W % S) return 0;
#define CALL(G) (dgrad_direct<G>(nullptr, s) ? SplitPlan{T64x64, 2} : dgrad_plan<G>(s))
    SplitPlan sp = [&]() -> SplitPlan { GZ_GEOM_DISPATCH_OR(CALL, (SplitPlan{T64x64, 2})) }();
#undef CALL
    if (is_tile2(sp.tile) && !(KH == 4 && KW == 4 && S == 2 && P == 1)) return 0;      // gather-loader launches: not fused
    if (sp.splits > 1) {          // the finish kernel writes them, one partial row per 32 pixels of a phase (round 4)
        if (!(KH == 4 && KW == 4 && S == 2 && P == 1)) return 0;       // phases of unequal length: own slab map, not fused
        return S * S * (int)(((long long)N * (H / S) * (W / S) + 31) / 32);
    }
    return S * S * stats_tm_rows(sp.tile, (long long)N * (H / S) * (W / S));
}

int gz_conv2d_dgrad_stats_ws(const float* y, const float* wpack, float* x, float* stats, float* workspace,
                             size_t ws_bytes, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S,
                             int P, hipStream_t stream) {
    gz::clear_stale_error();
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P) || !stats) return GZ_ERR_BAD_SHAPE;
    if (gz_conv2d_dgrad_stats_rows(N, C, H, W, K, OH, OW, KH, KW, S, P) <= 0) return GZ_ERR_UNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) != 0) return GZ_ERR_BAD_SHAPE;      // 16-byte LDS-DMA pieces / row stores
    if (too_large((long long)N * C * H * W) || too_large((long long)N * K * OH * OW)) return GZ_ERR_TOO_LARGE;
#define CALL(G)                                                                                                        \
    [&]() -> int {                                                                                                     \
        const SplitPlan sp = dgrad_plan<G>(s);                                                                         \
        if (sp.splits > 1 && (!workspace || ws_bytes < dgrad_ws_bytes<G>(s))) return GZ_ERR_WORKSPACE;                 \
        float* slab = sp.splits > 1 ? workspace : nullptr;                                                             \
        switch (sp.tile) {                                                                                             \
            case T256x256: return run_dgrad2<Cfg256x256>(y, wpack, nullptr, x, s, 0, 0.f, stream, stats);               \
            case T256x128: return run_dgrad2<Cfg256x128>(y, wpack, nullptr, x, s, 0, 0.f, stream, stats, sp.splits, slab); \
            case T512x64: return run_dgrad2<Cfg512x64>(y, wpack, nullptr, x, s, 0, 0.f, stream, stats);                 \
            case T256x64: return run_dgrad2<Cfg256x64>(y, wpack, nullptr, x, s, 0, 0.f, stream, stats, sp.splits, slab); \
            case T128x128: return run_dgrad<G, Cfg128x128>(y, wpack, nullptr, x, s, 0, 0.f, stream, sp.splits, slab, stats); \
            case T128x64: return run_dgrad<G, Cfg128x64>(y, wpack, nullptr, x, s, 0, 0.f, stream, sp.splits, slab, stats);   \
            case T128x32: return run_dgrad<G, Cfg128x32>(y, wpack, nullptr, x, s, 0, 0.f, stream, sp.splits, slab, stats);   \
            default: return run_dgrad<G, Cfg64x64>(y, wpack, nullptr, x, s, 0, 0.f, stream, sp.splits, slab, stats);         \
        }                                                                                                              \
    }()
    GZ_GEOM_DISPATCH(CALL)
#undef CALL
}

int gz_conv2d_dgrad_stats(const float* y, const float* wpack, float* x, float* stats, int N, int C, int H, int W, int K,
                          int OH, int OW, int KH, int KW, int S, int P, hipStream_t stream) {
    return gz_conv2d_dgrad_stats_ws(y, wpack, x, stats, nullptr, 0, N, C, H, W, K, OH, OW, KH, KW, S, P, stream);
}

long long gz_conv2d_pack_fwd_any_elems(int K, int C, int KH, int KW) {
    return (long long)round_bk(C) * KH * KW * round4(K);
}

int gz_conv2d_pack_fwd_any(const float* w, float* wp, int K, int C, int KH, int KW, hipStream_t stream) {
    gz::clear_stale_error();
    if (K <= 0 || C <= 0 || KH <= 0 || KW <= 0) return GZ_ERR_BAD_SHAPE;
    const int ld = round4(K);
    long long total = (long long)KH * KW * round_bk(C) * ld;
    hipLaunchKernelGGL(pack_fwd_tap_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)),
                       dim3(256), 0, stream, w, wp, K, C, KH * KW, round_bk(C), ld);
    return launch_status();
}

static bool any_shape_ok(const ConvShape& s, const AnyGeom& g) {
    if (s.N <= 0 || s.C <= 0 || s.K <= 0 || s.H <= 0 || s.W <= 0 || g.KH <= 0 || g.KW <= 0 || g.SH <= 0 || g.SW <= 0 ||
        g.PH < 0 || g.PW < 0)
        return false;
    return s.OH == (s.H + 2 * g.PH - g.KH) / g.SH + 1 && s.OW == (s.W + 2 * g.PW - g.KW) / g.SW + 1 && s.OH > 0 &&
           s.OW > 0;
}

size_t gz_conv2d_fwd_any_workspace_bytes(int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int SH,
                                         int SW, int PH, int PW) {
    ConvShape s{N, C, H, W, K, OH, OW};
    AnyGeom g{KH, KW, SH, SW, PH, PW};
    if (!any_shape_ok(s, g)) return 0;
    return split_bytes(fwd_any_plan(s, g), (long long)N * OH * OW, K, KH * KW * round_bk(C), 1);
}

int gz_conv2d_fwd_any(const float* x, const float* wpack, const float* bias, float* y, float* workspace,
                      size_t ws_bytes, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int SH, int SW,
                      int PH, int PW, int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    ConvShape s{N, C, H, W, K, OH, OW};
    AnyGeom g{KH, KW, SH, SW, PH, PW};
    if (!any_shape_ok(s, g)) return GZ_ERR_BAD_SHAPE;
    if ((long long)N * C * H * W * 4 >= (1ll << 31) || (long long)N * K * OH * OW * 4 >= (1ll << 31)) return GZ_ERR_TOO_LARGE;
    long long M = (long long)N * OH * OW;
    if (fwd_any2_cols(s, g, bias, act) == 64) return run_fwd_any2<Cfg256x64>(x, wpack, bias, y, s, g, act, slope, stream, K);
    SplitPlan sp = fwd_any_plan(s, g);
    if (sp.splits > 1 && (!workspace || ws_bytes < split_bytes(sp, M, K, KH * KW * round_bk(C), 1)))
        sp = SplitPlan{pick_tile(M, K, 1), 1};
    float* slab = sp.splits > 1 ? workspace : nullptr;
    switch (sp.tile) {
        case T128x128: return run_fwd_any<Cfg128x128>(x, wpack, bias, y, s, g, act, slope, stream, sp.splits, slab);
        case T128x64: return run_fwd_any<Cfg128x64>(x, wpack, bias, y, s, g, act, slope, stream, sp.splits, slab);
        case T128x32: return run_fwd_any<Cfg128x32>(x, wpack, bias, y, s, g, act, slope, stream, sp.splits, slab);
        default: return run_fwd_any<Cfg64x64>(x, wpack, bias, y, s, g, act, slope, stream, sp.splits, slab);
    }
}

int gz_conv2d_fwd_any_into(const float* x, const float* wpack, const float* bias, float* y, int y_image_channels, int N,
                           int C, int H, int W, int K, int OH, int OW, int KH, int KW, int SH, int SW, int PH, int PW,
                           int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    ConvShape s{N, C, H, W, K, OH, OW};
    AnyGeom g{KH, KW, SH, SW, PH, PW};
    if (!any_shape_ok(s, g) || y_image_channels < K) return GZ_ERR_BAD_SHAPE;
    if ((long long)N * C * H * W * 4 >= (1ll << 31) || (long long)N * y_image_channels * OH * OW * 4 >= (1ll << 31))
        return GZ_ERR_TOO_LARGE;
    if (fwd_any2_cols(s, g, bias, act) != 64) return GZ_ERR_UNSUPPORTED;
    return run_fwd_any2<Cfg256x64>(x, wpack, bias, y, s, g, act, slope, stream, y_image_channels);
}

#ifdef GZ2_STAMPS
/* diagnostic builds only: copies the per-workgroup stamps of the last igemm2 launch (n x 8 x u64) */
int gz_debug_read_stamps(void* out, int nwg) {
    return hip_status(hipMemcpyFromSymbol(out, HIP_SYMBOL(gz::gz2_stamps), (size_t)nwg * 64, 0, hipMemcpyDeviceToHost));
}
#endif

/* which tile configuration a launch of op (0 F, 1 Dg, 2 Wg) would use: 0 128x128, 1 128x64, 2 128x32, 3 64x64 */
int gz_conv2d_tile(int op, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S) {
    // (the reduction lengths below ignore the channel padding of the tap-major loaders: labels only)
    if (op == 0) {
        if (KH == 4 && KW == 4 && S == 2) {
            ConvShape s{N, C, H, W, K, OH, OW};
            const SplitPlan p2 = fwd2_plan<G4421>(s);
            if (p2.tile == T256x256 || p2.tile == T256x128 || p2.tile == T256x64) return p2.tile;
        }
        if ((KH == 5 && KW == 5 && S == 2) || (KH == 3 && KW == 3 && S == 1) || (KH == 1 && KW == 1 && S == 1)) {
            ConvShape s{N, C, H, W, K, OH, OW};
            const SplitPlan p2 = KH == 5 ? fwdtap2_plan<G5522>(s) : KH == 3 ? fwdtap2_plan<G3311>(s) : fwdtap2_plan<G1110>(s);
            if (p2.tile == T256x128 || p2.tile == T256x64) return p2.tile;
        }
        return pick_tile_fwd((long long)N * OH * OW, K, OW, KH, KW, S, C * KH * KW);
    }
    if (op == 1) {
        if (KH == 4 && KW == 4 && S == 2) {
            ConvShape s{N, C, H, W, K, OH, OW};
            if (dgrad2_ok<G4421>(s) && !dgrad_direct<G4421>(nullptr, s)) {
                const SplitPlan sp = dgrad_plan<G4421>(s);
                if (is_tile2(sp.tile)) return sp.tile;
            }
        }
        if (KH == 5 && KW == 5 && S == 2) {
            ConvShape s{N, C, H, W, K, OH, OW};
            if (dgrad5_plan<G5522>(s).ok) return T256x128P;
        }
        if ((KH == 5 && KW == 5 && S == 2) || (KH == 3 && KW == 3 && S == 1) || (KH == 1 && KW == 1 && S == 1)) {
            ConvShape s{N, C, H, W, K, OH, OW};
            const SplitPlan pt = KH == 5 ? dgradtap2_plan<G5522>(s) : KH == 3 ? dgradtap2_plan<G3311>(s) : dgradtap2_plan<G1110>(s);
            if (pt.tile == T256x128 || pt.tile == T256x64) return pt.tile;
        }
        return pick_tile((long long)N * (H / S) * (W / S), C, S * S, K * ((KH + S - 1) / S) * ((KW + S - 1) / S));
    }
    long long NTOT = (long long)C * KH * KW;
    int t;
    if (KH == 4 && KW == 4 && S == 2 && forced_tile() < 0) {
        ConvShape s{N, C, H, W, K, OH, OW};
        if (wgrad2_splits<G4421>(s) > 0) return wgrad2_narrow(s) ? T128x256 : T256x128;
    }
    if (((KH == 5 && KW == 5 && S == 2) || (KH == 3 && KW == 3 && S == 1)) && forced_tile() < 0) {
        // the LDS-DMA weight gradient with the generic raw-row image (needs H = S * OH, rows of 4 / 8 / 16k pixels)
        ConvShape s{N, C, H, W, K, OH, OW};
        const bool shape_ok = H == S * OH && W == S * OW && (W & 3) == 0 && (OW == 4 || OW == 8 || OW % 16 == 0) &&
                              !knobs().no_igemm2w && !knobs().no_igemm2wg;
        const int sp = !shape_ok ? 0 : KH == 5 ? wgrad2_splits<G5522>(s) : wgrad2_splits<G3311>(s);
        if (sp > 0) return wgrad2_narrow(s) ? T128x256 : T256x128;
    }
    if (NTOT <= 32) t = T128x32;
    else if (NTOT <= 64 || K <= 64) t = (K <= 64 ? T64x64 : T128x64);
    else {
        // split-K supplies the parallelism; with few pixels per split (small batches) the narrower
        // tile keeps more workgroups busy per slab byte
        long long pixels = (long long)N * OH * OW;
        t = pixels >= 8192 ? T128x128 : T128x64;     // round 2: 128x128 now holds 4 workgroups per CU (was 65536)
    }
    int f = forced_tile();
    if (f >= 0 && f <= 3 && !(f == T128x128 && NTOT <= 64)) t = f;
    return t;
}

/* ---- which kernel a launch takes, as text (round 4: the dispatch is pinned by tests/test_dispatch_plan.py) ---------
 * Runs on the CPU (no HIP call).  Describes the launch for 16-byte aligned tensors and a workspace of the advertised
 * size -- an unaligned view or a missing workspace falls back to the element-wise loaders / an unsplit plan. */
}  // extern "C" (the describe_* templates need C++ linkage)

static const char* tile_text(TileId t) {
    switch (t) {
        case T128x128: return "128x128";
        case T128x64: return "128x64";
        case T128x32: return "128x32";
        case T64x64: return "64x64";
        case T256x256: return "256x256";
        case T256x128: return "256x128";
        case T512x64: return "512x64";
        case T256x64: return "256x64";
        case T256x128P: return "256x(4x32)";
        default: return "128x256";
    }
}

// does this igemm2 launch run two wave groups per workgroup (gz_igemm.h: igemm2_use_kg2)?  The launcher's own arithmetic.
static bool kg2_applies(TileId t, long long M, long long N, int ny, int kdim, int splits, bool rows_loader) {
    if (!rows_loader || !(t == T256x128 || t == T256x64)) return false;
    const int chunks = (kdim + BK - 1) / BK;
    const int cps = (chunks + (splits < 1 ? 1 : splits) - 1) / (splits < 1 ? 1 : splits);
    const int nz = (chunks + cps - 1) / cps;
    return igemm2_use_kg2(tile_count(t, M, N, ny) * nz, cps);
}

template <class G>
static int describe_fwd(const ConvShape& s, char* b, size_t n) {
    if (G::kh == 3 && G::kw == 3 && G::s == 1 && G::p == 1 && conv3_smallch_ok(s.N, s.C, s.K, s.H, s.W))
        return snprintf(b, n, "F direct %s", (!knobs().no_fewk_conv && s.K <= 4 && s.C <= 64 && s.W == 64 && (s.H & 7) == 0)
                                                 ? "conv3x3_fewk<fma>" : "conv3x3_smallch<mfma16x16x4>");
    const SplitPlan sp = fwd_plan<G>(s);
    const int rows = gz_conv2d_fwd_stats_rows(s.N, s.C, s.H, s.W, s.K, s.OH, s.OW, G::kh, G::kw, G::s, G::p);
    if (is_tile2(sp.tile)) {
        const bool rowsA = fwd2_ok<G>(s);
        const int kdim = rowsA ? s.C * 16 : G::kh * G::kw * round_bk(s.C);
        // (the 1x1 plane loader has no two-group form; every other igemm2 forward loader does)
        const bool plane = !rowsA && G::kh * G::kw == 1 && G::s == 1 && G::p == 0 && !knobs().no_plane_a && ((s.H * s.W) & 3) == 0;
        const bool kg2 = kg2_applies(sp.tile, (long long)s.N * s.OH * s.OW, s.K, 1, kdim, sp.splits, !plane);
        return snprintf(b, n, "F igemm2<%s> %s slabs=%d bn_stats_rows=%d%s", tile_text(sp.tile),
                        rowsA ? "ConvFwdA2(raw rows, LDS-DMA 16B)" : "ConvTapA2(gather, LDS-DMA 4B)",
                        split_nz(kdim, sp.splits), rows, kg2 ? " wave_groups=2" : "");
    }
    const char* loader = "ConvFwdALoader";
    const int bm = sp.tile == T64x64 ? 64 : 128;
    if (BK % (G::kh * G::kw) != 0 && fwd_tap_major(s.C, G::kh, G::kw)) loader = "ConvFwdALoaderTap";
    else if (G::kh == 4 && G::kw == 4 && G::s == 2 && G::p == 1)
        loader = (!knobs().no_row4 && s.W == 2 * s.OW && s.H == 2 * s.OH && s.OW >= 16 && s.OW <= bm && bm % s.OW == 0)
                     ? "ConvFwdALoaderRow4" : "ConvFwdALoaderK4V";
    return snprintf(b, n, "F igemm<%s> %s slabs=%d bn_stats_rows=%d", tile_text(sp.tile), loader,
                    split_nz(fwd_kdim<G>(s), sp.splits), rows);
}

template <class G>
static int describe_dgrad(const ConvShape& s, char* b, size_t n) {
    if (s.H % G::s || s.W % G::s) return snprintf(b, n, "Dg unsupported (H, W not multiples of the stride)");
    if (G::kh == 3 && G::kw == 3 && G::s == 1 && G::p == 1 && conv3_smallch_ok(s.N, s.K, s.C, s.H, s.W))
        return snprintf(b, n, "Dg direct conv3x3_smallch<mfma16x16x4>");
    const long long M = (long long)s.N * s.OH * s.OW;
    if (dgrad_direct<G>(nullptr, s)) {
        if (!knobs().smallc_one_pos && s.OW % 4 == 0 && 64 % (s.OW / 4) == 0)
            return snprintf(b, n, "Dg direct dgrad_smallc4_k4s2p1<C=%d,KS=%d>", s.C, smallc_split(M / 4, s.K));
        return snprintf(b, n, "Dg direct dgrad_smallc_k4s2p1<C=%d>", s.C);
    }
    if (G::kh == 5 && G::kw == 5 && G::s == 2 && G::p == 2 && dgrad_direct5_ok(nullptr, nullptr, s))
        return snprintf(b, n, "Dg direct dgrad_smallc4_k5s2p2<C=%d,KS=%d>", s.C, smallc_split(M / 4, s.K));
    if constexpr (G::kh == 5 && G::kw == 5 && G::s == 2 && G::p == 2) {
        const Dg5Plan p5 = dgrad5_plan<G>(s);
        if (p5.ok)
            return snprintf(b, n, "Dg igemm2<256x(4 phases x 32)> ConvDg5A2(row-shared, LDS-DMA 16B, 12 k-steps) slabs=%d "
                                  "(no bias / activation, aligned tensors; else the gather loader)", p5.nz);
    }
    constexpr int TAPS = ((G::kh + G::s - 1) / G::s) * ((G::kw + G::s - 1) / G::s);
    const SplitPlan sp = dgrad_plan<G>(s);
    const int rows = gz_conv2d_dgrad_stats_rows(s.N, s.C, s.H, s.W, s.K, s.OH, s.OW, G::kh, G::kw, G::s, G::p);
    const bool tapm = dgrad_tap_major(s.K, G::kh, G::kw, G::s);
    const int kk = tapm ? round_bk(s.K) : s.K;
    if (is_tile2(sp.tile)) {
        const bool rowsA = dgrad2_ok<G>(s);
        constexpr int TYX = ((G::kh + G::s - 1) / G::s) * ((G::kw + G::s - 1) / G::s);
        const bool plane = !rowsA && G::kh * G::kw == 1 && G::s == 1 && G::p == 0 && !knobs().no_plane_a && ((s.OH * s.OW) & 3) == 0;
        const bool kg2 = kg2_applies(sp.tile, (long long)s.N * (s.H / G::s) * (s.W / G::s), s.C, G::s * G::s,
                                     rowsA ? 4 * s.K : TYX * round_bk(s.K), sp.splits, !plane);
        return snprintf(b, n, "Dg igemm2<%s> %s splits=%d bn_stats_rows=%d%s", tile_text(sp.tile),
                        rowsA ? "ConvDgA2(row-shared, LDS-DMA 16B)" : "ConvDgTapA2(gather, LDS-DMA 4B)", sp.splits, rows,
                        kg2 ? " wave_groups=2" : "");
    }
    const char* loader = "ConvDgALoader";
    if (tapm && !(G::kh % G::s == 0 && G::kw % G::s == 0 && BK % TAPS == 0)) loader = "ConvDgALoaderTap";
    else if (G::kh == 4 && G::kw == 4 && G::s == 2 && G::p == 1 && !knobs().no_row4 && (s.W / 2) % 4 == 0)
        loader = "ConvDgALoaderRow4";
    return snprintf(b, n, "Dg igemm<%s> %s splits=%d bn_stats_rows=%d", tile_text(sp.tile), loader,
                    sp.splits > 1 ? sp.splits : 1, rows);
    (void)kk;
}

template <class G>
static int describe_wgrad(const ConvShape& s, char* b, size_t n) {
    if (wgrad_smallch_ok(s, G::kh, G::kw, G::s, G::p))
        return snprintf(b, n, "Wg direct %s slabs=%d", wgrad_fewk_ok(s) ? "wgrad_k3_fewk<fma>" : "wgrad_smallch_k3<mfma16x16x4>",
                        wgrad_smallch_blocks(s));
    if (G::kh == 4 && G::kw == 4 && G::s == 2 && G::p == 1 && wgrad_k4s2p1_fewc_ok(s))
        return snprintf(b, n, "Wg direct wgrad_k4s2p1_fewc<mfma16x16x4,C=%d,KT=%d> slabs=%d", s.C, s.K / 16,
                        wgrad_k4s2p1_fewc_blocks(s));
    TileId t = (TileId)gz_conv2d_tile(2, s.N, s.C, s.H, s.W, s.K, s.OH, s.OW, G::kh, G::kw, G::s);
    const int chunks = (s.N * s.OH * s.OW + BK - 1) / BK;
    const int NTOT = s.C * G::kh * G::kw;
    if (t == T256x128 || t == T128x256) {
        const int splits = wgrad2_splits<G>(s);
        if (splits > 0) {
            const int cw = wgrad2w_cw_shape<G>(s);
            const int cps = (chunks + splits - 1) / splits, nz = (chunks + cps - 1) / cps;
            if (cw)
                return snprintf(b, n, "Wg igemm2w<%s> %s<CW=%d>(both operands LDS-DMA) slabs=%d", tile_text(t),
                                (G::kh == 4 && G::kw == 4) ? "WgImgB2" : "WgImgBG", cw, nz);
            if (G::kh == 4 && G::kw == 4)
                return snprintf(b, n, "Wg igemm2r<%s> WgALoaderRow+WgBLoaderRow(register-staged) slabs=%d", tile_text(t), nz);
        }
        t = T128x128;
    }
    const int bm = t == T64x64 ? 64 : 128, bn = t == T128x128 ? 128 : (t == T128x32 ? 32 : 64);
    const long long tiles = (long long)((s.K + bm - 1) / bm) * ((NTOT + bn - 1) / bn);
    const int splits = wgrad_splits(tiles, chunks, bm * bn >= 128 * 128);
    const int cps = (chunks + splits - 1) / splits, nz = (chunks + cps - 1) / cps;
    WgRowGeom rg;
    const bool row = !knobs().wg_generic && wg_row_geom<G>(s, &rg);
    return snprintf(b, n, "Wg igemm<%s> %s slabs=%d", tile_text(t), row ? "WgALoaderRow+WgBLoaderRow" : "WgALoader+WgBLoader", nz);
}

extern "C" {

int gz_conv2d_plan(int op, int N, int C, int H, int W, int K, int OH, int OW, int KH, int KW, int S, int P, char* buf,
                   int buflen) {
    if (!buf || buflen <= 0) return GZ_ERR_BAD_SHAPE;
    buf[0] = 0;
    ConvShape s{N, C, H, W, K, OH, OW};
    if (!shape_ok(s, KH, KW, S, P) || op < 0 || op > 2) return GZ_ERR_BAD_SHAPE;
    const size_t n = (size_t)buflen;
#define CALL(G) (op == 0 ? describe_fwd<G>(s, buf, n) : op == 1 ? describe_dgrad<G>(s, buf, n) : describe_wgrad<G>(s, buf, n))
    GZ_GEOM_DISPATCH(CALL)
#undef CALL
}

size_t gz_gemm_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return split_bytes(plan_split(M, N, K, 1, pick_tile(M, N, 1)), M, N, K, 1);
}

int gz_gemm(const float* a, const float* b, const float* bias, float* c, float* workspace, size_t ws_bytes, int M,
            int N, int K, int lda, int ldb, int ldc, int trans_a, int trans_b, int act, float slope,
            hipStream_t stream) {
    gz::clear_stale_error();
    if (M <= 0 || N <= 0 || K <= 0) return GZ_ERR_BAD_SHAPE;
    if (too_large((long long)M * K) || too_large((long long)K * N) || too_large((long long)M * N)) return GZ_ERR_TOO_LARGE;
    SplitPlan sp = plan_split(M, N, K, 1, pick_tile(M, N, 1));
    if (sp.splits > 1 && (!workspace || ws_bytes < split_bytes(sp, M, N, K, 1))) sp = SplitPlan{pick_tile(M, N, 1), 1};
    float* slab = sp.splits > 1 ? workspace : nullptr;
    switch (sp.tile) {
        case T128x128: return gemm_ops<Cfg128x128>(a, b, bias, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, slope, stream, sp.splits, slab);
        case T128x64: return gemm_ops<Cfg128x64>(a, b, bias, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, slope, stream, sp.splits, slab);
        case T128x32: return gemm_ops<Cfg128x32>(a, b, bias, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, slope, stream, sp.splits, slab);
        default: return gemm_ops<Cfg64x64>(a, b, bias, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, slope, stream, sp.splits, slab);
    }
}

}  // extern "C"
