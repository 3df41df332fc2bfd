// Evaluation-path helpers (the InceptionV3 feature extractor behind FID / KID, reference
// core/callback_inception_metrics.py:183-246, core/submodules/gan_stability/metrics/inception.py): 2-D pooling,
// bilinear resize.  One lane per output element, lanes along the row; forward only.
#include "gz_common.h"
#include "gz_knobs.h"
#include "../../include/gz_ops.h"

namespace gz {

constexpr int IT = 256;

// mode 0: max;  1: average over the window positions INSIDE the image (count_include_pad=False, TensorFlow's);
// 2: average over KS*KS (count_include_pad=True)
__global__ __launch_bounds__(IT) void pool2d_kernel(const float* __restrict__ x, float* __restrict__ y, long long total,
                                                   int H, int W, int OH, int OW, int KS, int S, int P, int mode) {
    const long long stride = (long long)gridDim.x * IT;
    for (long long i = (long long)blockIdx.x * IT + threadIdx.x; i < total; i += stride) {
        const int ox = (int)(i % OW);
        const long long t = i / OW;
        const int oy = (int)(t % OH);
        const long long plane = t / OH;
        const float* src = x + plane * H * W;
        const int y0 = oy * S - P, x0 = ox * S - P;
        float acc = mode == 0 ? -INFINITY : 0.f;
        int cnt = 0;
        for (int dy = 0; dy < KS; ++dy) {
            const int iy = y0 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int dx = 0; dx < KS; ++dx) {
                const int ix = x0 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float v = src[iy * W + ix];
                acc = mode == 0 ? fmaxf(acc, v) : acc + v;
                ++cnt;
            }
        }
        if (mode == 1) acc = acc / (float)cnt;
        if (mode == 2) acc = acc / (float)(KS * KS);
        y[i] = acc;
    }
}

// The same pooling with a PLANE per workgroup pass (round 6): the plane (<= 8192 floats: 35 x 35, 17 x 17, 8 x 8, 73 x 73 of
// InceptionV3) is staged in LDS with coalesced loads, the window is read from there, all index arithmetic is 32-bit (the
// flat kernel above spends three 64-bit divisions per output: 435 us for the 88 M floats of a [250, 288, 35, 35] average
// pool = 1.6 TB/s of read + write).
__global__ __launch_bounds__(IT) void pool2d_plane_kernel(const float* __restrict__ x, float* __restrict__ y, int planes,
                                                         int H, int W, int OH, int OW, int KS, int S, int P, int mode,
                                                         gz::FastDiv div_ow) {
    extern __shared__ float plane[];
    const int hw = H * W, ohw = OH * OW;
    for (int pl = blockIdx.x; pl < planes; pl += gridDim.x) {
        const float* src = x + (long long)pl * hw;
        __syncthreads();                                   // (the previous plane's readers are done)
        for (int i = threadIdx.x; i < hw; i += IT) plane[i] = src[i];
        __syncthreads();
        float* dst = y + (long long)pl * ohw;
        for (int i = threadIdx.x; i < ohw; i += IT) {
            const int oy = (int)gz::fdiv((uint32_t)i, div_ow), ox = i - oy * OW;
            const int y0 = oy * S - P, x0 = ox * S - P;
            float acc = mode == 0 ? -INFINITY : 0.f;
            int cnt = 0;
            for (int dy = 0; dy < KS; ++dy) {
                const int iy = y0 + dy;
                if ((unsigned)iy >= (unsigned)H) continue;
                for (int dx = 0; dx < KS; ++dx) {
                    const int ix = x0 + dx;
                    if ((unsigned)ix >= (unsigned)W) continue;
                    const float v = plane[iy * W + ix];
                    acc = mode == 0 ? fmaxf(acc, v) : acc + v;
                    ++cnt;
                }
            }
            if (mode == 1) acc = acc / (float)cnt;
            if (mode == 2) acc = acc / (float)(KS * KS);
            dst[i] = acc;
        }
    }
}

// The InceptionV3 windows (3 x 3, stride 1 pad 1 or stride 2 pad 0) with everything known at compile time (round 6): the
// run-time-window kernel above spends ~100 instructions per output on loop control and bounds tests and was
// instruction-bound at 2.2 TB/s.  Here a workgroup stages a GROUP of whole planes (~2048 floats: one 35 x 35, seven
// 17 x 17, thirty-two 8 x 8 -- a lone 17 x 17 plane left most of the second sweep idle) into LDS images with a border of
// P elements that holds the window's neutral element (0 / -inf; written once, the interiors are overwritten per group),
// so the nine taps are unconditional reads at constant offsets.  Sums run in the same order as before; padded taps add
// an exact zero.
template <int KS, int S, int P, int MODE>
__global__ __launch_bounds__(IT) void pool2d_group_kernel(const float* __restrict__ x, float* __restrict__ y, int planes,
                                                         int H, int W, int OH, int OW, int G, gz::FastDiv div_hw,
                                                         gz::FastDiv div_w, gz::FastDiv div_ohw, gz::FastDiv div_ow) {
    extern __shared__ float plane[];
    const int hw = H * W, ohw = OH * OW, PW = W + 2 * P, PS = (H + 2 * P) * PW;
    if (P > 0) {
        for (int i = threadIdx.x; i < G * PS; i += IT) plane[i] = MODE == 0 ? -INFINITY : 0.f;
    }
    const int groups = (planes + G - 1) / G;
    for (int g = blockIdx.x; g < groups; g += gridDim.x) {
        const int n_pl = min(G, planes - g * G);
        const float* src = x + (long long)g * G * hw;
        __syncthreads();                                   // (border written / the previous group's readers are done)
        const int n_in = n_pl * hw;
#pragma unroll 4
        for (int i = threadIdx.x; i < n_in; i += IT) {
            const int pl = (int)gz::fdiv((uint32_t)i, div_hw), rem = i - pl * hw;
            const int yy = (int)gz::fdiv((uint32_t)rem, div_w), xx = rem - yy * W;
            plane[pl * PS + (yy + P) * PW + xx + P] = src[i];
        }
        __syncthreads();
        float* dst = y + (long long)g * G * ohw;
        const int n_out = n_pl * ohw;
        for (int i = threadIdx.x; i < n_out; i += IT) {
            const int pl = (int)gz::fdiv((uint32_t)i, div_ohw), rem = i - pl * ohw;
            const int oy = (int)gz::fdiv((uint32_t)rem, div_ow), ox = rem - oy * OW;
            const float* w0 = plane + pl * PS + oy * S * PW + ox * S;
            float acc = MODE == 0 ? -INFINITY : 0.f;
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int dx = 0; dx < KS; ++dx) {
                    const float v = w0[dy * PW + dx];
                    acc = MODE == 0 ? fmaxf(acc, v) : acc + v;
                }
            if (MODE == 1) {                               // window positions inside the image (KS <= H, W)
                const int cy = KS - max(0, P - oy * S) - max(0, oy * S - P + KS - H);
                const int cx = KS - max(0, P - ox * S) - max(0, ox * S - P + KS - W);
                acc = acc / (float)(cy * cx);
            }
            if (MODE == 2) acc = acc / (float)(KS * KS);
            dst[i] = acc;
        }
    }
}

// Planes too large for one LDS image (147 x 147 before the first max-pool): a workgroup takes a BAND of OB output rows of
// one plane; its input rows are one contiguous span of the plane.  Unpadded windows only (P = 0).
template <int KS, int S, int MODE>
__global__ __launch_bounds__(IT) void pool2d_band_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                        int OH, int OW, int OB, int bands, gz::FastDiv div_bands,
                                                        gz::FastDiv div_ow) {
    extern __shared__ float plane[];
    const int pl = (int)gz::fdiv((uint32_t)blockIdx.x, div_bands), band = blockIdx.x - pl * bands;
    const int oy0 = band * OB, rows = min(OB, OH - oy0);
    const float* src = x + ((long long)pl * H + oy0 * S) * W;
    const int n_in = ((rows - 1) * S + KS) * W;
#pragma unroll 4
    for (int i = threadIdx.x; i < n_in; i += IT) plane[i] = src[i];
    __syncthreads();
    float* dst = y + ((long long)pl * OH + oy0) * OW;
    const int n_out = rows * OW;
    for (int i = threadIdx.x; i < n_out; i += IT) {
        const int oy = (int)gz::fdiv((uint32_t)i, div_ow), ox = i - oy * OW;
        const float* w0 = plane + oy * S * W + ox * S;
        float acc = MODE == 0 ? -INFINITY : 0.f;
#pragma unroll
        for (int dy = 0; dy < KS; ++dy)
#pragma unroll
            for (int dx = 0; dx < KS; ++dx) {
                const float v = w0[dy * W + dx];
                acc = MODE == 0 ? fmaxf(acc, v) : acc + v;
            }
        if (MODE != 0) acc = acc / (float)(KS * KS);
        dst[i] = acc;
    }
}

// Global average (the 8 x 8 -> 1 x 1 pool in front of the 2048-d features): a plane of 4 * L floats per L lanes
// (L = 16 for 8 x 8), float4 per lane, butterfly over the L lanes; the 512 k planes of a batch of 250 were one
// workgroup each in the plane kernel (1.19 ms).
template <int L>
__global__ __launch_bounds__(IT) void global_avg_kernel(const float* __restrict__ x, float* __restrict__ y, int planes,
                                                       float inv) {
    const int unit = (blockIdx.x * IT + threadIdx.x) / L, sub = threadIdx.x % L;
    float acc = 0.f;
    if (unit < planes) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((long long)unit * L + sub) * 4);
        acc = (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int m = L / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (unit < planes && sub == 0) y[unit] = acc * inv;
}

template <int KS, int S, int P>
static void launch_pool_group(const float* x, float* y, int planes, int H, int W, int OH, int OW, int mode,
                              hipStream_t stream) {
    const int hw = H * W, PS = (H + 2 * P) * (W + 2 * P);
    int G = 2048 / hw;
    G = G < 1 ? 1 : G;
    const int groups = (planes + G - 1) / G;
    const int grid = groups < 8192 ? groups : 8192;
    const size_t lds = (size_t)G * PS * 4;
    const gz::FastDiv a = gz::make_fastdiv(hw), b = gz::make_fastdiv(W), c = gz::make_fastdiv(OH * OW),
                      d = gz::make_fastdiv(OW);
    if (mode == 0)
        hipLaunchKernelGGL((pool2d_group_kernel<KS, S, P, 0>), dim3(grid), dim3(IT), lds, stream, x, y, planes, H, W, OH, OW,
                           G, a, b, c, d);
    else if (mode == 1)
        hipLaunchKernelGGL((pool2d_group_kernel<KS, S, P, 1>), dim3(grid), dim3(IT), lds, stream, x, y, planes, H, W, OH, OW,
                           G, a, b, c, d);
    else
        hipLaunchKernelGGL((pool2d_group_kernel<KS, S, P, 2>), dim3(grid), dim3(IT), lds, stream, x, y, planes, H, W, OH, OW,
                           G, a, b, c, d);
}

// F.interpolate(mode='bilinear', align_corners=False) followed by  out * mul + add  (inception.py:141-149: resize to
// 299 x 299, then 2x - 1)
__global__ __launch_bounds__(IT) void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            long long total, int H, int W, int OH, int OW, float sh,
                                                            float sw, float mul, float add) {
    const long long stride = (long long)gridDim.x * IT;
    for (long long i = (long long)blockIdx.x * IT + threadIdx.x; i < total; i += stride) {
        const int ox = (int)(i % OW);
        const long long t = i / OW;
        const int oy = (int)(t % OH);
        const long long plane = t / OH;
        const float* src = x + plane * H * W;
        // ATen's area_pixel_compute_source_index (align_corners=False): max(0, (dst + 0.5) * scale - 0.5)
        const float fy = fmaxf((oy + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * sw - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float v = hy * (hx * src[y0 * W + x0] + lx * src[y0 * W + x1]) +
                        ly * (hx * src[y1 * W + x0] + lx * src[y1 * W + x1]);
        y[i] = v * mul + add;
    }
}

static int infer_grid(long long items) {
    long long b = (items + IT - 1) / IT;
    if (b > 256 * 16) b = 256 * 16;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace gz

using namespace gz;

extern "C" {

int gz_pool2d(const float* x, float* y, long long planes, int H, int W, int OH, int OW, int KS, int S, int P, int mode,
              hipStream_t stream) {
    gz::clear_stale_error();
    if (planes <= 0 || H <= 0 || W <= 0 || KS <= 0 || S <= 0 || P < 0 || mode < 0 || mode > 2) return GZ_ERR_BAD_SHAPE;
    if (OH != (H + 2 * P - KS) / S + 1 || OW != (W + 2 * P - KS) / S + 1 || OH <= 0 || OW <= 0) return GZ_ERR_BAD_SHAPE;
    const long long total = planes * OH * OW;
    if (KS == H && KS == W && P == 0 && mode != 0 && H * W == 64 && planes < (1ll << 31) / 16 && !((uintptr_t)x & 15) &&
        !gz::knobs().no_pool_group) {
        const long long threads = planes * 16;
        hipLaunchKernelGGL((global_avg_kernel<16>), dim3((unsigned)((threads + IT - 1) / IT)), dim3(IT), 0, stream, x, y,
                           (int)planes, 1.f / 64.f);
        return launch_status();
    }
    if (KS == 3 && S == 2 && P == 0 && H * W > 8192 && W <= 1024 && planes * 64 < (1ll << 31) && !gz::knobs().no_pool_group) {
        const int OB = 8, bands = (OH + OB - 1) / OB;
        const size_t lds = (size_t)((OB - 1) * S + KS) * W * 4;
        const gz::FastDiv a = gz::make_fastdiv(bands), b = gz::make_fastdiv(OW);
        const unsigned grid = (unsigned)(planes * bands);
        if (mode == 0)
            hipLaunchKernelGGL((pool2d_band_kernel<3, 2, 0>), dim3(grid), dim3(IT), lds, stream, x, y, H, W, OH, OW, OB, bands,
                               a, b);
        else
            hipLaunchKernelGGL((pool2d_band_kernel<3, 2, 2>), dim3(grid), dim3(IT), lds, stream, x, y, H, W, OH, OW, OB, bands,
                               a, b);
        return launch_status();
    }
    if (KS == 3 && H >= 3 && W >= 3 && H * W <= 8192 && planes < (1ll << 31) && !gz::knobs().no_pool_plane &&
        !gz::knobs().no_pool_group && ((S == 1 && P == 1) || (S == 2 && P == 0))) {
        if (S == 1) launch_pool_group<3, 1, 1>(x, y, (int)planes, H, W, OH, OW, mode, stream);
        else launch_pool_group<3, 2, 0>(x, y, (int)planes, H, W, OH, OW, mode, stream);
        return launch_status();
    }
    if (H * W <= 8192 && planes < (1ll << 31) && !gz::knobs().no_pool_plane) {
        const int grid = (int)(planes < 65536 ? planes : 65536);
        hipLaunchKernelGGL(pool2d_plane_kernel, dim3(grid), dim3(IT), (size_t)H * W * 4, stream, x, y, (int)planes, H, W, OH,
                           OW, KS, S, P, mode, gz::make_fastdiv(OW));
        return launch_status();
    }
    hipLaunchKernelGGL(pool2d_kernel, dim3(infer_grid(total)), dim3(IT), 0, stream, x, y, total, H, W, OH, OW, KS, S, P,
                       mode);
    return launch_status();
}

int gz_resize_bilinear(const float* x, float* y, long long planes, int H, int W, int OH, int OW, float mul, float add,
                       hipStream_t stream) {
    gz::clear_stale_error();
    if (planes <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return GZ_ERR_BAD_SHAPE;
    const long long total = planes * OH * OW;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(infer_grid(total)), dim3(IT), 0, stream, x, y, total, H, W, OH, OW,
                       (float)H / (float)OH, (float)W / (float)OW, mul, add);
    return launch_status();
}

}  // extern "C"
