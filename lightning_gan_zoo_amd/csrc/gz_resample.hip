// HoloGAN's 3-D rigid-body resampling (reference core/models/hologan_generator.py:198-321):
// every output voxel (z, y, x) of the S^3 grid is mapped through the per-sample inverse transform,
// its 8 CLAMPED neighbours are gathered with weights computed from the clamped corners (so weights
// outside the volume can be negative / not sum to one -- reference behaviour, preserved), and the
// result is written directly in the layout the "learned projection" consumes:
//     out2d[n][c*S + (S-1-y)][z][x]      (= permute(0,1,3,2,4) -> flip(dim 2) -> reshape, :130-133)
// HBM-bound gather (forward) / float-atomic scatter-add (backward; 8 adds of 4 B per output element).
#include <cstdlib>
#include "gz_common.h"
#include "gz_knobs.h"
#include "../../include/gz_ops.h"

namespace gz {

constexpr int RS_THREADS = 256;
constexpr int RS_CB = 8;   // channels per thread

struct Corner8 {
    int off[8];     // offsets inside one channel volume (z*S*S + y*S + x), reference order a..h
    float w[8];
};

// LDS placement of voxel e = (a * S + b) * S + c of a volume staged as [S^3][4 floats] (round 5).  A 16-byte element's
// bank group is e & 7 -- the low bits of the INNERMOST coordinate alone -- so a wavefront whose gathered voxels differ
// mostly in the outer coordinates (a view rotated by ~90 degrees maps an output row onto a source column) hits one bank
// group 16 times over.  XOR-ing those three bits with the low bits of a ^ b spreads any axis-aligned run over all
// eight groups; sh = log2(S) for power-of-two S >= 8, 0 = plain placement.
__device__ __forceinline__ int lds_slot(int e, int sh) {
    return sh ? e ^ (((e >> (2 * sh)) ^ (e >> sh)) & 7) : e;
}

static int swizzle_shift(int S) {
    if (knobs().no_resample_swizzle || S < 8 || (S & (S - 1))) return 0;
    int sh = 0;
    while ((1 << sh) < S) ++sh;
    return sh;
}

// minv: row-major 4x4 (already inverted on the host side exactly as the reference does)
__device__ __forceinline__ Corner8 corners(const float* __restrict__ m, int x, int y, int z, int S) {
    const float fx = (float)x, fy = (float)y, fz = (float)z;
    // k-ordered multiply-add chain, as a GEMM micro-kernel does for [4x4] x [4 x S^3]
    float sx = fmaf(m[3], 1.f, fmaf(m[2], fz, fmaf(m[1], fy, m[0] * fx)));
    float sy = fmaf(m[7], 1.f, fmaf(m[6], fz, fmaf(m[5], fy, m[4] * fx)));
    float sz = fmaf(m[11], 1.f, fmaf(m[10], fz, fmaf(m[9], fy, m[8] * fx)));
    int x0 = (int)floorf(sx), y0 = (int)floorf(sy), z0 = (int)floorf(sz);
    int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    const int hi = S - 1;
    x0 = min(max(x0, 0), hi); x1 = min(max(x1, 0), hi);
    y0 = min(max(y0, 0), hi); y1 = min(max(y1, 0), hi);
    z0 = min(max(z0, 0), hi); z1 = min(max(z1, 0), hi);
    const float wx[2] = {(float)x1 - sx, sx - (float)x0};
    const float wy[2] = {(float)y1 - sy, sy - (float)y0};
    const float wz[2] = {(float)z1 - sz, sz - (float)z0};
    const int xs[2] = {x0, x1}, ys[2] = {y0, y1}, zs[2] = {z0, z1};
    Corner8 c;
    int k = 0;
#pragma unroll
    for (int kz = 0; kz < 2; ++kz)
#pragma unroll
        for (int kx = 0; kx < 2; ++kx)
#pragma unroll
            for (int ky = 0; ky < 2; ++ky) {
                c.off[k] = (zs[kz] * S + ys[ky]) * S + xs[kx];
                c.w[k] = (wx[kx] * wy[ky]) * wz[kz];
                ++k;
            }
    return c;
}

__global__ __launch_bounds__(RS_THREADS) void resample_fwd_kernel(const float* __restrict__ vox,
                                                                  const float* __restrict__ minv,
                                                                  float* __restrict__ out, long long* idx_out, int N,
                                                                  int C, int S) {
    const int S3 = S * S * S;
    long long v = (long long)blockIdx.x * RS_THREADS + threadIdx.x;
    if (v >= (long long)N * S3) return;
    const int n = (int)(v / S3);
    const int r = (int)(v - (long long)n * S3);
    const int z = r / (S * S), y = (r / S) % S, x = r % S;
    Corner8 cn = corners(minv + n * 16, x, y, z, S);
    if (idx_out && blockIdx.y == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) idx_out[(long long)k * N * S3 + v] = (long long)n * S3 + cn.off[k];
    }
    const int c0 = blockIdx.y * RS_CB;
#pragma unroll
    for (int j = 0; j < RS_CB; ++j) {
        int c = c0 + j;
        if (c >= C) break;
        const float* src = vox + ((long long)n * C + c) * S3;
        float acc = cn.w[0] * src[cn.off[0]];
#pragma unroll
        for (int k = 1; k < 8; ++k) acc += cn.w[k] * src[cn.off[k]];
        out[(((long long)n * C * S + (long long)c * S + (S - 1 - y)) * S + z) * S + x] = acc;
    }
}

// Forward with the source volumes staged through LDS (round 2): the lanes of a wavefront are neighbours in the OUTPUT
// grid, their corners are neighbours in a rotated lattice of the source -- from global memory every 4-byte corner
// read pulls its own 64-byte line (170 us for 2 x 134 MB).  A workgroup stages FW_LC channel volumes of one sample
// with coalesced 16-byte loads and gathers the corners from LDS.
constexpr int FW_LC = 4;

__global__ __launch_bounds__(RS_THREADS) void resample_fwd_staged_kernel(const float* __restrict__ vox,
                                                                         const float* __restrict__ minv,
                                                                         float* __restrict__ out, int N, int C, int S,
                                                                         int sh) {
    // [S^3][FW_LC], channels interleaved (round 4): a corner's FW_LC channel values are ONE 16-byte LDS read instead of
    // FW_LC scattered 4-byte ones -- the gathers' bank conflicts, not HBM, set this kernel's pace (111 us for 134 MB)
    static_assert(FW_LC == 4, "one f32x4 per corner");
    extern __shared__ __attribute__((aligned(16))) float vl[];
    const int S3 = S * S * S;
    const int n = blockIdx.x;
    const int c0 = blockIdx.y * FW_LC;
    const int lc = min(FW_LC, C - c0);
    const float* src = vox + ((long long)n * C + c0) * S3;
    for (int i = threadIdx.x * 4; i < FW_LC * S3; i += RS_THREADS * 4) {
        const int j = i / S3, u = i - j * S3;            // (S^3 is a multiple of 4: a float4 stays inside one channel)
        const f32x4 v = j < lc ? *reinterpret_cast<const f32x4*>(src + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        vl[lds_slot(u + 0, sh) * FW_LC + j] = v[0];
        vl[lds_slot(u + 1, sh) * FW_LC + j] = v[1];
        vl[lds_slot(u + 2, sh) * FW_LC + j] = v[2];
        vl[lds_slot(u + 3, sh) * FW_LC + j] = v[3];
    }
    __syncthreads();
    const float* m = minv + n * 16;
    const f32x4* vq = reinterpret_cast<const f32x4*>(vl);
    for (int r = threadIdx.x; r < S3; r += RS_THREADS) {
        const int z = r / (S * S), y = (r / S) % S, x = r % S;
        Corner8 cn = corners(m, x, y, z, S);
        f32x4 acc = vq[lds_slot(cn.off[0], sh)] * cn.w[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) acc = acc + vq[lds_slot(cn.off[k], sh)] * cn.w[k];   // same order as resample_fwd_kernel
#pragma unroll
        for (int j = 0; j < FW_LC; ++j)
            if (j < lc) out[(((long long)n * C * S + (long long)(c0 + j) * S + (S - 1 - y)) * S + z) * S + x] = acc[j];
    }
}

// Adjoint of the gather.  One workgroup owns the S^3 gradient volumes of BW_CB channels of one sample
// in LDS (BW_CB * 16 KiB at S = 16), walks all S^3 output voxels, adds their 8 weighted contributions with
// LDS float atomics, and finally streams the volumes out with coalesced stores -- no global atomics
// (scattered global float atomics ran at 77 GB/s here: one lane per 64-byte segment).
constexpr int BW_CB = 2;

__global__ __launch_bounds__(RS_THREADS) void resample_bwd_kernel(const float* __restrict__ gout,
                                                                  const float* __restrict__ minv,
                                                                  float* __restrict__ gvox, int N, int C, int S) {
    extern __shared__ __attribute__((aligned(16))) float vol[];      // [BW_CB][S^3]
    const int S3 = S * S * S;
    const int n = blockIdx.x;
    const int c0 = blockIdx.y * BW_CB;
    for (int i = threadIdx.x; i < BW_CB * S3; i += RS_THREADS) vol[i] = 0.f;
    __syncthreads();
    const float* m = minv + n * 16;
    for (int r = threadIdx.x; r < S3; r += RS_THREADS) {
        const int z = r / (S * S), y = (r / S) % S, x = r % S;
        Corner8 cn = corners(m, x, y, z, S);
#pragma unroll
        for (int j = 0; j < BW_CB; ++j) {
            const int c = c0 + j;
            if (c >= C) break;
            const float g = gout[(((long long)n * C * S + (long long)c * S + (S - 1 - y)) * S + z) * S + x];
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(&vol[j * S3 + cn.off[k]], cn.w[k] * g);
        }
    }
    __syncthreads();
    for (int j = 0; j < BW_CB; ++j) {
        const int c = c0 + j;
        if (c >= C) break;
        float* dst = gvox + ((long long)n * C + c) * S3;
        for (int i = threadIdx.x; i < S3; i += RS_THREADS) dst[i] = vol[j * S3 + i];
    }
}

// Adjoint in GATHER form (round 2).  The scatter kernel above spends its time in LDS float atomics (ds_add_f32
// retires one lane at a time: ~700 cycles per wave-instruction measured, 682 us for 64 x 64 x 16^3 = 0.2 TB/s).
// Here a lane owns one SOURCE voxel u and collects  gvox[u] = sum_v w(u, v) * g[v]  over the output voxels v whose
// (clamped) corners include u.  The map v -> source position p = A v + t is affine, so those v lie in the image of
// the cube (u-1, u+1)^3 under A^-1: a parallelepiped whose bounding box (half extents sum_j |A^-1_ij|, a few lattice
// points per axis) is enumerated; every candidate's position is recomputed with the forward kernel's own arithmetic
// and its weight onto u is the product over the axes of the corner weights whose clamped index equals u -- which
// reproduces the reference's clamped-corner semantics exactly (out-of-volume corners that collapse onto one index
// carry weights +a and -a and cancel, as in the forward gather).  No atomics, every output written once, the g
// reads are plain cached loads.  Channels are split over blockIdx.z.
constexpr int BG_CH = 32;      // channels per lane (accumulators in registers)
constexpr int BG_HITS = 12;    // list entries per lane between flushes

__device__ __forceinline__ float axis_weight(float s, int S, int u) {
    int i0 = (int)floorf(s), i1 = i0 + 1;
    const int hi = S - 1;
    i0 = min(max(i0, 0), hi);
    i1 = min(max(i1, 0), hi);
    float w = 0.f;
    if (i0 == u) w += (float)i1 - s;
    if (i1 == u) w += s - (float)i0;
    return w;
}

__global__ __launch_bounds__(RS_THREADS) void resample_bwd_gather_kernel(const float* __restrict__ gout,
                                                                         const float* __restrict__ minv,
                                                                         float* __restrict__ gvox, int N, int C, int S,
                                                                         const int* __restrict__ only_if) {
    if (only_if && !*only_if) return;      // fallback of the staged form: runs only when a hit list overflowed
    const int S3 = S * S * S;
    const int n = blockIdx.y;
    const int c0 = blockIdx.z * BG_CH;
    const int u = blockIdx.x * RS_THREADS + threadIdx.x;
    if (u >= S3) return;
    const int uz = u / (S * S), uy = (u / S) % S, ux = u % S;
    const float* m = minv + n * 16;
    // p = A v + t with v = (x, y, z):  A = m[0..2][0..2], t = m[.][3]
    const float a00 = m[0], a01 = m[1], a02 = m[2], t0 = m[3];
    const float a10 = m[4], a11 = m[5], a12 = m[6], t1 = m[7];
    const float a20 = m[8], a21 = m[9], a22 = m[10], t2 = m[11];
    // A^-1 by cofactors (only steers the candidate search; a margin absorbs its rounding)
    const float c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const float c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
    const float c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
    const float det = a00 * c00 + a01 * c10 + a02 * c20;
    const float id = 1.f / det;
    const float px = (float)ux - t0, py = (float)uy - t1, pz = (float)uz - t2;
    const float vcx = (c00 * px + c01 * py + c02 * pz) * id;      // centre of the candidate box, v = (x, y, z)
    const float vcy = (c10 * px + c11 * py + c12 * pz) * id;
    const float vcz = (c20 * px + c21 * py + c22 * pz) * id;
    const float aid = fabsf(id);
    const float ex = (fabsf(c00) + fabsf(c01) + fabsf(c02)) * aid + 1e-3f;
    const float ey = (fabsf(c10) + fabsf(c11) + fabsf(c12)) * aid + 1e-3f;
    const float ez = (fabsf(c20) + fabsf(c21) + fabsf(c22)) * aid + 1e-3f;
    // boundary voxels also receive the (cancelling) clamped corners of positions up to one cell outside: the box of
    // a boundary u is taken around the cube (u-1, u+1)^3 as for interior ones, which covers them
    const int x_lo = max((int)ceilf(vcx - ex), 0), x_hi = min((int)floorf(vcx + ex), S - 1);
    const int y_lo = max((int)ceilf(vcy - ey), 0), y_hi = min((int)floorf(vcy + ey), S - 1);
    const int z_lo = max((int)ceilf(vcz - ez), 0), z_hi = min((int)floorf(vcz + ez), S - 1);
    float acc[BG_CH];
#pragma unroll
    for (int j = 0; j < BG_CH; ++j) acc[j] = 0.f;
    const float* gbase = gout + ((long long)n * C + c0) * S3;
    // Two phases, so that the wavefront does not walk the channel loop once per CANDIDATE (the lanes hit different
    // candidates: 64 iterations x 16 loads, mostly with a few lanes active -- that form ran as slowly as the atomics):
    // first every lane compacts its own hits (offset, weight) into a short list in LDS, then the lanes walk their
    // lists in step, ~8 entries each.  A full list is flushed in between (views that zoom out have more hits).
    __shared__ int hit_off[BG_HITS][RS_THREADS];
    __shared__ float hit_w[BG_HITS][RS_THREADS];
    int cnt = 0;
    auto flush = [&]() {
        for (int k = 0; k < cnt; ++k) {
            const float* gp = gbase + hit_off[k][threadIdx.x];
            const float w = hit_w[k][threadIdx.x];
#pragma unroll
            for (int j = 0; j < BG_CH; ++j)
                if (c0 + j < C) acc[j] = fmaf(w, gp[(long long)j * S3], acc[j]);
        }
        cnt = 0;
    };
    for (int z = z_lo; z <= z_hi; ++z)
        for (int y = y_lo; y <= y_hi; ++y)
            for (int x = x_lo; x <= x_hi; ++x) {
                const float fx = (float)x, fy = (float)y, fz = (float)z;
                // the forward kernel's own chain (corners())
                const float sx = fmaf(t0, 1.f, fmaf(a02, fz, fmaf(a01, fy, a00 * fx)));
                const float sy = fmaf(t1, 1.f, fmaf(a12, fz, fmaf(a11, fy, a10 * fx)));
                const float sz = fmaf(t2, 1.f, fmaf(a22, fz, fmaf(a21, fy, a20 * fx)));
                const float w = (axis_weight(sx, S, ux) * axis_weight(sy, S, uy)) * axis_weight(sz, S, uz);
                if (w != 0.f) {
                    if (cnt == BG_HITS) flush();
                    hit_off[cnt][threadIdx.x] = ((S - 1 - y) * S + z) * S + x;     // out2d[n][c*S + (S-1-y)][z][x]
                    hit_w[cnt][threadIdx.x] = w;
                    ++cnt;
                }
            }
    flush();
#pragma unroll
    for (int j = 0; j < BG_CH; ++j)
        if (c0 + j < C) gvox[((long long)n * C + c0 + j) * S3 + u] = acc[j];
}

// ---- the form that runs: hit lists + LDS-staged gradient volumes -------------------------------------------
// The gather above still runs at 0.3 TB/s: the lanes of a wavefront are neighbours in the SOURCE volume, their
// hits are neighbours in a rotated lattice, so every 4-byte load pulls its own 64-byte line through L2.  Split it:
//   (1) resample_hits_kernel: once per (sample, source voxel) -- not per channel -- the <= BH (offset, weight)
//       pairs, written hit-major so that the consumer reads them coalesced (N * S^3 * BH * 8 B of workspace);
//   (2) resample_bwd_staged_kernel: a workgroup stages the gradient volumes of LC channels of one sample into LDS
//       with coalesced 16-byte loads (each byte of gout leaves HBM once), then every lane walks the lists of its
//       source voxels and gathers from LDS, where a scattered 4-byte read costs a bank conflict, not a cache line.
// A voxel with more than BH hits (views that zoom out) sets a flag and the launch falls back to the gather kernel.
constexpr int BH = 24;         // a rotated open cube of side 2 holds 8 lattice points on average, rarely more than 16
constexpr int LC = 4;          // channels staged per workgroup: LC * 16 KiB of LDS at S = 16

__global__ __launch_bounds__(RS_THREADS) void resample_hits_kernel(const float* __restrict__ minv,
                                                                   int* __restrict__ hoff, float* __restrict__ hw,
                                                                   int* __restrict__ overflow, int N, int S) {
    const int S3 = S * S * S;
    const int n = blockIdx.y;
    const int u = blockIdx.x * RS_THREADS + threadIdx.x;
    if (u >= S3) return;
    const int uz = u / (S * S), uy = (u / S) % S, ux = u % S;
    const float* m = minv + n * 16;
    const float a00 = m[0], a01 = m[1], a02 = m[2], t0 = m[3];
    const float a10 = m[4], a11 = m[5], a12 = m[6], t1 = m[7];
    const float a20 = m[8], a21 = m[9], a22 = m[10], t2 = m[11];
    const float c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const float c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
    const float c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
    const float id = 1.f / (a00 * c00 + a01 * c10 + a02 * c20);
    const float px = (float)ux - t0, py = (float)uy - t1, pz = (float)uz - t2;
    const float vcx = (c00 * px + c01 * py + c02 * pz) * id;
    const float vcy = (c10 * px + c11 * py + c12 * pz) * id;
    const float vcz = (c20 * px + c21 * py + c22 * pz) * id;
    const float aid = fabsf(id);
    const float ex = (fabsf(c00) + fabsf(c01) + fabsf(c02)) * aid + 1e-3f;
    const float ey = (fabsf(c10) + fabsf(c11) + fabsf(c12)) * aid + 1e-3f;
    const float ez = (fabsf(c20) + fabsf(c21) + fabsf(c22)) * aid + 1e-3f;
    const int x_lo = max((int)ceilf(vcx - ex), 0), x_hi = min((int)floorf(vcx + ex), S - 1);
    const int y_lo = max((int)ceilf(vcy - ey), 0), y_hi = min((int)floorf(vcy + ey), S - 1);
    const int z_lo = max((int)ceilf(vcz - ez), 0), z_hi = min((int)floorf(vcz + ez), S - 1);
    const long long NS3 = (long long)N * S3;
    const long long slot = (long long)n * S3 + u;
    int cnt = 0;
    for (int z = z_lo; z <= z_hi; ++z)
        for (int y = y_lo; y <= y_hi; ++y)
            for (int x = x_lo; x <= x_hi; ++x) {
                const float fx = (float)x, fy = (float)y, fz = (float)z;
                const float sx = fmaf(t0, 1.f, fmaf(a02, fz, fmaf(a01, fy, a00 * fx)));
                const float sy = fmaf(t1, 1.f, fmaf(a12, fz, fmaf(a11, fy, a10 * fx)));
                const float sz = fmaf(t2, 1.f, fmaf(a22, fz, fmaf(a21, fy, a20 * fx)));
                const float w = (axis_weight(sx, S, ux) * axis_weight(sy, S, uy)) * axis_weight(sz, S, uz);
                if (w != 0.f) {
                    if (cnt < BH) {
                        hoff[cnt * NS3 + slot] = ((S - 1 - y) * S + z) * S + x;
                        hw[cnt * NS3 + slot] = w;
                    }
                    ++cnt;
                }
            }
    hoff[BH * NS3 + slot] = min(cnt, BH);      // the list's length follows the BH offset planes
    if (cnt > BH) *overflow = 1;
}

__global__ __launch_bounds__(RS_THREADS) void resample_bwd_staged_kernel(const float* __restrict__ gout,
                                                                         const int* __restrict__ hoff,
                                                                         const float* __restrict__ hw,
                                                                         const int* __restrict__ overflow,
                                                                         float* __restrict__ gvox, int N, int C, int S,
                                                                         int sh) {
    static_assert(LC == 4, "one f32x4 per hit");
    extern __shared__ __attribute__((aligned(16))) float gl[];      // [S^3][LC], channels interleaved (see the forward)
    if (*overflow) return;                 // the launcher re-runs this gradient with the gather kernel
    const int S3 = S * S * S;
    // (Tried, round 5: an XCD-aware block order that sends all sixteen channel groups of a sample to one XCD so that the
    // sample's 278 KB of hit lists stay in that L2 -- 100 -> 105 us: the lists already come out of the last-level cache.)
    const int n = blockIdx.x;
    const int c0 = blockIdx.y * LC;
    const int lc = min(LC, C - c0);
    const float* src = gout + ((long long)n * C + c0) * S3;
    for (int i = threadIdx.x * 4; i < LC * S3; i += RS_THREADS * 4) {
        const int j = i / S3, u = i - j * S3;
        const f32x4 v = j < lc ? *reinterpret_cast<const f32x4*>(src + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        gl[lds_slot(u + 0, sh) * LC + j] = v[0];
        gl[lds_slot(u + 1, sh) * LC + j] = v[1];
        gl[lds_slot(u + 2, sh) * LC + j] = v[2];
        gl[lds_slot(u + 3, sh) * LC + j] = v[3];
    }
    __syncthreads();
    const long long NS3 = (long long)N * S3;
    const f32x4* gq = reinterpret_cast<const f32x4*>(gl);
    // Two source voxels per lane and pass, their hit lists read in batches of HB with all 4 * HB list loads in flight
    // (round 5; one dependent pair of loads per hit before -- a lane-dependent trip count kept the compiler from hoisting
    // them: ~12 serial round trips per voxel, 16 voxels per lane): 99 -> 88 us; four voxels per pass: 86 us.  Same
    // summation order per voxel.  Planes past a voxel's count are unwritten memory and never enter a sum.
    constexpr int HB = 8, UV = 2;
    static_assert(BH % HB == 0, "whole batches");
    for (int u0 = threadIdx.x; u0 < S3; u0 += UV * RS_THREADS) {
        long long slot[UV];
        int cnt[UV];
        f32x4 acc[UV];
#pragma unroll
        for (int v = 0; v < UV; ++v) {
            const int u = u0 + v * RS_THREADS;
            slot[v] = (long long)n * S3 + (u < S3 ? u : u0);
            cnt[v] = u < S3 ? hoff[BH * NS3 + slot[v]] : 0;
            acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int k0 = 0; __any(k0 < max(cnt[0], cnt[1])); k0 += HB) {
            static_assert(UV == 2, "the loop bound spells out the two counts");
            int offs[UV][HB];
            float ws[UV][HB];
#pragma unroll
            for (int v = 0; v < UV; ++v)
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    offs[v][k] = hoff[(long long)(k0 + k) * NS3 + slot[v]];
                    ws[v][k] = hw[(long long)(k0 + k) * NS3 + slot[v]];
                }
#pragma unroll
            for (int v = 0; v < UV; ++v)
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    const bool live = k0 + k < cnt[v];
                    const f32x4 g4 = gq[lds_slot(live ? offs[v][k] : 0, sh)];
                    if (live) {
#pragma unroll
                        for (int j = 0; j < LC; ++j) acc[v][j] = fmaf(ws[v][k], g4[j], acc[v][j]);
                    }
                }
        }
#pragma unroll
        for (int v = 0; v < UV; ++v) {
            const int u = u0 + v * RS_THREADS;
            if (u < S3) {
#pragma unroll
                for (int j = 0; j < LC; ++j)
                    if (j < lc) gvox[((long long)n * C + c0 + j) * S3 + u] = acc[v][j];
            }
        }
    }
}

}  // namespace gz

using namespace gz;

extern "C" {

int gz_rigid_resample_fwd(const float* vox, const float* minv, float* out2d, long long* idx_out, int N, int C, int S,
                          hipStream_t stream) {
    gz::clear_stale_error();
    if (N <= 0 || C <= 0 || S <= 0) return GZ_ERR_BAD_SHAPE;
    const int S3 = S * S * S;
    const size_t lds = (size_t)FW_LC * S3 * sizeof(float);
    const bool direct = knobs().resample_fwd_direct;      // experiment: the round-1 kernel
    if (!idx_out && !direct && S3 % 4 == 0 && lds <= 64 * 1024 && (((uintptr_t)vox) & 15) == 0) {
        hipLaunchKernelGGL(resample_fwd_staged_kernel, dim3(N, (C + FW_LC - 1) / FW_LC), dim3(RS_THREADS), lds, stream,
                           vox, minv, out2d, N, C, S, swizzle_shift(S));
        return launch_status();
    }
    long long vox_n = (long long)N * S * S * S;
    dim3 grid((unsigned)((vox_n + RS_THREADS - 1) / RS_THREADS), (C + RS_CB - 1) / RS_CB);
    hipLaunchKernelGGL(resample_fwd_kernel, grid, dim3(RS_THREADS), 0, stream, vox, minv, out2d, idx_out, N, C, S);
    return launch_status();
}

size_t gz_rigid_resample_bwd_workspace_bytes(int N, int S) {
    if (N <= 0 || S <= 0) return 0;
    return (size_t)N * S * S * S * (BH * 8 + 4) + 16;      // hit offsets + lengths, weights, the overflow flag
}

int gz_rigid_resample_bwd(const float* gout2d, const float* minv, float* gvox, float* workspace, size_t ws_bytes,
                          int N, int C, int S, hipStream_t stream) {
    gz::clear_stale_error();
    if (N <= 0 || C <= 0 || S <= 0) return GZ_ERR_BAD_SHAPE;
    const int mode = knobs().resample_bwd_mode;      // experiments: 1 "scatter" (round 1), 2 "gather"
    const int S3 = S * S * S;
    const size_t lds_staged = (size_t)LC * S3 * sizeof(float);
    if (!mode && workspace && ws_bytes >= gz_rigid_resample_bwd_workspace_bytes(N, S) && S3 % 4 == 0 &&
        lds_staged <= 64 * 1024 && (((uintptr_t)gout2d | (uintptr_t)workspace) & 15) == 0) {
        const long long NS3 = (long long)N * S3;
        int* hoff = reinterpret_cast<int*>(workspace);                    // [BH + 1][N * S^3]
        float* hw = workspace + NS3 * (BH + 1);                           // [BH][N * S^3]
        int* flag = reinterpret_cast<int*>(workspace + NS3 * (2 * BH + 1));
        hipMemsetAsync(flag, 0, sizeof(int), stream);
        hipLaunchKernelGGL(resample_hits_kernel, dim3((S3 + RS_THREADS - 1) / RS_THREADS, N), dim3(RS_THREADS), 0, stream,
                           minv, hoff, hw, flag, N, S);
        hipLaunchKernelGGL(resample_bwd_staged_kernel, dim3(N, (C + LC - 1) / LC), dim3(RS_THREADS), lds_staged, stream,
                           gout2d, hoff, hw, flag, gvox, N, C, S, swizzle_shift(S));
        // more than BH hits somewhere (a view that zooms out): the staged kernel did nothing; no host sync --
        // the gather kernel reads the same flag and runs only then
        dim3 grid((S3 + RS_THREADS - 1) / RS_THREADS, N, (C + BG_CH - 1) / BG_CH);
        hipLaunchKernelGGL(resample_bwd_gather_kernel, grid, dim3(RS_THREADS), 0, stream, gout2d, minv, gvox, N, C, S,
                           flag);
        return launch_status();
    }
    if (!mode || mode == 2) {
        dim3 grid((S3 + RS_THREADS - 1) / RS_THREADS, N, (C + BG_CH - 1) / BG_CH);
        hipLaunchKernelGGL(resample_bwd_gather_kernel, grid, dim3(RS_THREADS), 0, stream, gout2d, minv, gvox, N, C, S,
                           (const int*)nullptr);
        return launch_status();
    }
    size_t lds = (size_t)BW_CB * S * S * S * sizeof(float);
    if (lds > 64 * 1024) return GZ_ERR_UNSUPPORTED;
    dim3 grid(N, (C + BW_CB - 1) / BW_CB);
    hipLaunchKernelGGL(resample_bwd_kernel, grid, dim3(RS_THREADS), lds, stream, gout2d, minv, gvox, N, C, S);
    return launch_status();
}

}  // extern "C"
