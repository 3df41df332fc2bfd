// HoloGAN's 3-D rigid-body resampling (reference core/models/hologan_generator.py:198-321):
// every output voxel (z, y, x) of the S^3 grid is mapped through the per-sample inverse transform,
// its 8 CLAMPED neighbours are gathered with weights computed from the clamped corners (so weights
// outside the volume can be negative / not sum to one -- reference behaviour, preserved), and the
// result is written directly in the layout the "learned projection" consumes:
//     out2d[n][c*S + (S-1-y)][z][x]      (= permute(0,1,3,2,4) -> flip(dim 2) -> reshape, :130-133)
// HBM-bound gather (forward) / float-atomic scatter-add (backward; 8 adds of 4 B per output element).
#include "gz_common.h"
#include "../../include/gz_ops.h"

namespace gz {

constexpr int RS_THREADS = 256;
constexpr int RS_CB = 8;   // channels per thread

struct Corner8 {
    int off[8];     // offsets inside one channel volume (z*S*S + y*S + x), reference order a..h
    float w[8];
};

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

}  // namespace gz

using namespace gz;

extern "C" {

int gz_rigid_resample_fwd(const float* vox, const float* minv, float* out2d, long long* idx_out, int N, int C, int S,
                          hipStream_t stream) {
    gz::clear_stale_error();
    if (N <= 0 || C <= 0 || S <= 0) return GZ_ERR_BAD_SHAPE;
    long long vox_n = (long long)N * S * S * S;
    dim3 grid((unsigned)((vox_n + RS_THREADS - 1) / RS_THREADS), (C + RS_CB - 1) / RS_CB);
    hipLaunchKernelGGL(resample_fwd_kernel, grid, dim3(RS_THREADS), 0, stream, vox, minv, out2d, idx_out, N, C, S);
    return launch_status();
}

int gz_rigid_resample_bwd(const float* gout2d, const float* minv, float* gvox, int N, int C, int S,
                          hipStream_t stream) {
    gz::clear_stale_error();
    if (N <= 0 || C <= 0 || S <= 0) return GZ_ERR_BAD_SHAPE;
    size_t lds = (size_t)BW_CB * S * S * S * sizeof(float);
    if (lds > 64 * 1024) return GZ_ERR_UNSUPPORTED;
    dim3 grid(N, (C + BW_CB - 1) / BW_CB);
    hipLaunchKernelGGL(resample_bwd_kernel, grid, dim3(RS_THREADS), lds, stream, gout2d, minv, gvox, N, C, S);
    return launch_status();
}

}  // extern "C"
