// Evaluation-path helpers (the InceptionV3 feature extractor behind FID / KID, reference
// core/callback_inception_metrics.py:183-246, core/submodules/gan_stability/metrics/inception.py): 2-D pooling,
// bilinear resize.  One lane per output element, lanes along the row; forward only.
#include "gz_common.h"
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
