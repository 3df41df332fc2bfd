// HBM-bound helpers around the convolution stack:
//   * the discriminator's last layer (Conv2d(C,1,k4,s2,p0) on a 4x4 map = one 16*C-long dot per
//     sample, reference core/models/standard_networks.py:27-30) as row-dot / outer / column-dot,
//     which again form a closed {F, Dg, Wg} triple for the double backward;
//   * the gradient-penalty tail (core/utils/utils.py:41-42,55-57): per-sample lerp, per-sample
//     sum of squares, per-sample scaling;
//   * WGAN weight clipping (core/lightning_module.py:160-162).
#include "gz_common.h"
#include "../../include/gz_ops.h"

namespace gz {

constexpr int MT = 256;

// y[r] = sum_l a[r][l] * b[r or 0][l]   (b_rows == 1 broadcasts b)
__global__ __launch_bounds__(MT) void rowdot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ y, int R, int L4, int b_bcast) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * MT + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * MT) >> 6;
    for (int r = wave; r < R; r += nwaves) {
        const f32x4* pa = reinterpret_cast<const f32x4*>(a) + (long long)r * L4;
        const f32x4* pb = reinterpret_cast<const f32x4*>(b) + (b_bcast ? 0 : (long long)r * L4);
        float s0 = 0.f, s1 = 0.f;
        int q = lane;
        for (; q + 64 < L4; q += 128) {
            f32x4 u = pa[q], w = pb[q], u2 = pa[q + 64], w2 = pb[q + 64];
            s0 += (u.x * w.x + u.y * w.y) + (u.z * w.z + u.w * w.w);
            s1 += (u2.x * w2.x + u2.y * w2.y) + (u2.z * w2.z + u2.w * w2.w);
        }
        for (; q < L4; q += 64) {
            f32x4 u = pa[q], w = pb[q];
            s0 += (u.x * w.x + u.y * w.y) + (u.z * w.z + u.w * w.w);
        }
        float s = wave_sum(s0 + s1);
        if (lane == 0) y[r] = s;
    }
}

// out[r][l] = s[r] * x[r or 0][l] (+ s2[r] * x2[r][l])
__global__ __launch_bounds__(MT) void rowscale_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                      const float* __restrict__ x2, const float* __restrict__ s2,
                                                      float* __restrict__ out, long long total4, FastDiv div_l4,
                                                      int L4, int x_bcast, int one_minus_s2) {
    const long long stride = (long long)gridDim.x * MT;
    for (long long i = (long long)blockIdx.x * MT + threadIdx.x; i < total4; i += stride) {
        uint32_t r = fdiv((uint32_t)i, div_l4);
        uint32_t q = (uint32_t)i - r * (uint32_t)L4;
        float sv = s[r];
        f32x4 v = reinterpret_cast<const f32x4*>(x)[x_bcast ? (long long)q : i];
        f32x4 o = v * sv;
        if (x2) {
            float tv = one_minus_s2 ? 1.f - sv : s2[r];
            f32x4 w = reinterpret_cast<const f32x4*>(x2)[i];
            o = o + w * tv;
        }
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

// slab[z][l] = sum_{r in slice z} g[r] * x[r][l]
__global__ __launch_bounds__(MT) void coldot_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                    float* __restrict__ slab, int R, int L4, int rows_per_slice) {
    int q = blockIdx.x * MT + threadIdx.x;
    if (q >= L4) return;
    int r0 = blockIdx.y * rows_per_slice;
    int r1 = min(R, r0 + rows_per_slice);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[(long long)r * L4 + q];
        acc = acc + v * g[r];
    }
    reinterpret_cast<f32x4*>(slab)[(long long)blockIdx.y * L4 + q] = acc;
}

__global__ __launch_bounds__(MT) void slab_sum_kernel(const float* __restrict__ slab, float* __restrict__ out, int S,
                                                      int L) {
    int i = blockIdx.x * MT + threadIdx.x;
    if (i >= L) return;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += slab[(long long)s * L + i];
    out[i] = a;
}

__global__ __launch_bounds__(MT) void clamp_kernel(float* __restrict__ p, long long n, float lo, float hi) {
    const long long stride = (long long)gridDim.x * MT;
    for (long long i = (long long)blockIdx.x * MT + threadIdx.x; i < n; i += stride) {
        float v = p[i];
        p[i] = v < lo ? lo : (v > hi ? hi : v);
    }
}

static int grid_for(long long items) {
    long long b = (items + MT - 1) / MT;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

// out[l] = sum_r x[r][l]: the bias gradient of an nn.Linear (R = batch rows, at most a few hundred).  A lane owns a
// column; the 16 lane-groups of a workgroup take every 16th row and meet in LDS in a fixed order.
__global__ __launch_bounds__(MT) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int L) {
    __shared__ float part[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s = 0.f;
    if (c < L)
        for (int r = rg; r < R; r += MT / 16) s += x[(long long)r * L + c];
    part[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < L) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][cl];
        out[c] = t;
    }
}

}  // namespace gz

using namespace gz;

extern "C" {

int gz_colsum(const float* x, float* out, int R, int L, hipStream_t stream) {
    gz::clear_stale_error();
    if (R <= 0 || L <= 0) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(colsum_kernel, dim3((L + 15) / 16), dim3(MT), 0, stream, x, out, R, L);
    return launch_status();
}

int gz_rowdot(const float* a, const float* b, float* y, int R, int L, int b_broadcast, hipStream_t stream) {
    gz::clear_stale_error();
    if (R <= 0 || L <= 0 || (L & 3)) return GZ_ERR_BAD_SHAPE;
    int blocks = (R + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rowdot_kernel, dim3(blocks), dim3(MT), 0, stream, a, b, y, R, L / 4, b_broadcast);
    return launch_status();
}

int gz_rowscale(const float* x, const float* s, const float* x2, const float* s2, float* out, int R, int L,
                int x_broadcast, int one_minus_s, hipStream_t stream) {
    gz::clear_stale_error();
    if (R <= 0 || L <= 0 || (L & 3)) return GZ_ERR_BAD_SHAPE;
    long long total4 = (long long)R * (L / 4);
    if (total4 >= (1ll << 31)) return GZ_ERR_TOO_LARGE;
    hipLaunchKernelGGL(rowscale_kernel, dim3(grid_for(total4)), dim3(MT), 0, stream, x, s, x2, s2, out, total4,
                       make_fastdiv(L / 4), L / 4, x_broadcast, one_minus_s);
    return launch_status();
}

size_t gz_coldot_workspace_bytes(int R, int L) {
    int slices = R >= 64 ? 32 : 1;
    return (size_t)slices * L * 4;
}

int gz_coldot(const float* g, const float* x, float* out, float* workspace, size_t ws_bytes, int R, int L,
              hipStream_t stream) {
    gz::clear_stale_error();
    if (R <= 0 || L <= 0 || (L & 3)) return GZ_ERR_BAD_SHAPE;
    int slices = R >= 64 ? 32 : 1;
    if (slices > 1 && ws_bytes < (size_t)slices * L * 4) return GZ_ERR_WORKSPACE;
    int rps = (R + slices - 1) / slices;
    int L4 = L / 4;
    float* dst = slices > 1 ? workspace : out;
    hipLaunchKernelGGL(coldot_kernel, dim3((L4 + MT - 1) / MT, slices), dim3(MT), 0, stream, g, x, dst, R, L4, rps);
    if (slices > 1)
        hipLaunchKernelGGL(slab_sum_kernel, dim3((L + MT - 1) / MT), dim3(MT), 0, stream, workspace, out, slices, L);
    return launch_status();
}

int gz_clamp_(float* p, long long count, float lo, float hi, hipStream_t stream) {
    gz::clear_stale_error();
    if (count <= 0) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(clamp_kernel, dim3(grid_for(count)), dim3(MT), 0, stream, p, count, lo, hi);
    return launch_status();
}

}  // extern "C"

extern "C" const char* gz_last_error(void) { return gz::last_error_slot(); }

extern "C" const char* gz_build_info(void) { return "gz_hip gfx950 fp32-mfma " __DATE__; }
