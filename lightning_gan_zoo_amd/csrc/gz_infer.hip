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
