// HBM-bound pieces of the R1-regularised ResNet path: the pre-activation
// LeakyReLU, the residual combination `x_s + 0.1*dx`, AvgPool2d(3, stride 2, padding 1) and the
// nearest-neighbour 2x Upsample of reference core/submodules/gan_stability/models/resnet.py:30-33,
// 72-75,119-124,131-133.  All four are linear (or piecewise linear) maps, so each forward/backward
// pair below is also its own double-backward pair: the adjoint of the adjoint is the forward.
#include "gz_common.h"
#include "../../include/gz_ops.h"

namespace gz {

constexpr int RT = 256;

static int rn_grid(long long items) {
    long long b = (items + RT - 1) / RT;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

__global__ __launch_bounds__(RT) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                     long long total4, int act, float slope) {
    const long long stride = (long long)gridDim.x * RT;
    for (long long i = (long long)blockIdx.x * RT + threadIdx.x; i < total4; i += stride) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = act_fwd(v[q], act, slope);
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}

// out = alpha*a + beta*b  (b may be null: out = alpha*a); optionally act_out = act(out) in the same pass
__global__ __launch_bounds__(RT) void axpby_kernel(const float* __restrict__ a, float alpha,
                                                   const float* __restrict__ b, float beta,
                                                   float* __restrict__ out, float* __restrict__ act_out,
                                                   long long total4, int act, float slope) {
    const long long stride = (long long)gridDim.x * RT;
    for (long long i = (long long)blockIdx.x * RT + threadIdx.x; i < total4; i += stride) {
        f32x4 o = reinterpret_cast<const f32x4*>(a)[i] * alpha;
        if (b) o = o + reinterpret_cast<const f32x4*>(b)[i] * beta;
        reinterpret_cast<f32x4*>(out)[i] = o;
        if (act_out) {
            f32x4 t;
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = act_fwd(o[q], act, slope);
            reinterpret_cast<f32x4*>(act_out)[i] = t;
        }
    }
}

// AvgPool2d(3, 2, 1), count_include_pad: y[oh][ow] = (1/9) * sum over the 3x3 window centred on (2oh, 2ow)
// with out-of-map taps contributing zero.  One thread per output element; the 3 rows it reads are shared with
// its neighbours through L1/L2.
__global__ __launch_bounds__(RT) void avgpool3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            long long total, int H, int W, int OH, int OW,
                                                            FastDiv div_ow, FastDiv div_oh) {
    const long long stride = (long long)gridDim.x * RT;
    for (long long i = (long long)blockIdx.x * RT + threadIdx.x; i < total; i += stride) {
        uint32_t t = fdiv((uint32_t)i, div_ow);
        int ow = (int)((uint32_t)i - t * (uint32_t)OW);
        uint32_t p = fdiv(t, div_oh);
        int oh = (int)(t - p * (uint32_t)OH);
        const float* src = x + (long long)p * H * W;
        float s = 0.f;
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh) {
            int h = 2 * oh + dh;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                int w = 2 * ow + dw;
                if (w >= 0 && w < W) s += src[h * W + w];
            }
        }
        y[i] = s * (1.f / 9.f);
    }
}

// adjoint of the above: gx[h][w] = (1/9) * sum of gy over the (at most 2x2) windows that cover (h, w)
__global__ __launch_bounds__(RT) void avgpool3s2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                            long long total, int H, int W, int OH, int OW,
                                                            FastDiv div_w, FastDiv div_h) {
    const long long stride = (long long)gridDim.x * RT;
    for (long long i = (long long)blockIdx.x * RT + threadIdx.x; i < total; i += stride) {
        uint32_t t = fdiv((uint32_t)i, div_w);
        int w = (int)((uint32_t)i - t * (uint32_t)W);
        uint32_t p = fdiv(t, div_h);
        int h = (int)(t - p * (uint32_t)H);
        const float* src = gy + (long long)p * OH * OW;
        // windows centred on 2*o cover 2o-1..2o+1: an even coordinate has one covering window, an odd one two
        int oh0 = h >> 1, oh1 = (h & 1) ? oh0 + 1 : oh0;
        int ow0 = w >> 1, ow1 = (w & 1) ? ow0 + 1 : ow0;
        float s = 0.f;
        for (int oh = oh0; oh <= oh1; ++oh) {
            if (oh >= OH) continue;
            for (int ow = ow0; ow <= ow1; ++ow)
                if (ow < OW) s += src[oh * OW + ow];
        }
        gx[i] = s * (1.f / 9.f);
    }
}

// nn.Upsample(scale_factor=2), nearest: y[2h+a][2w+b] = x[h][w].  One thread per input element, two 8-byte stores.
__global__ __launch_bounds__(RT) void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           long long total, int W, FastDiv div_w) {
    const long long stride = (long long)gridDim.x * RT;
    for (long long i = (long long)blockIdx.x * RT + threadIdx.x; i < total; i += stride) {
        uint32_t row = fdiv((uint32_t)i, div_w);          // plane*H + h
        uint32_t w = (uint32_t)i - row * (uint32_t)W;
        float v = x[i];
        f32x2 vv = {v, v};
        float* dst = y + (long long)row * 4 * W + 2 * w;  // row (2*(plane*H+h)) of width 2W
        *reinterpret_cast<f32x2*>(dst) = vv;
        *reinterpret_cast<f32x2*>(dst + 2 * W) = vv;
    }
}

__global__ __launch_bounds__(RT) void upsample2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                           long long total, int W, FastDiv div_w) {
    const long long stride = (long long)gridDim.x * RT;
    for (long long i = (long long)blockIdx.x * RT + threadIdx.x; i < total; i += stride) {
        uint32_t row = fdiv((uint32_t)i, div_w);
        uint32_t w = (uint32_t)i - row * (uint32_t)W;
        const float* src = gy + (long long)row * 4 * W + 2 * w;
        f32x2 a = *reinterpret_cast<const f32x2*>(src);
        f32x2 b = *reinterpret_cast<const f32x2*>(src + 2 * W);
        gx[i] = (a.x + a.y) + (b.x + b.y);
    }
}

}  // namespace gz

using namespace gz;

extern "C" {

int gz_act_fwd(const float* x, float* y, long long count, int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    if (count <= 0 || (count & 3)) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(rn_grid(count / 4)), dim3(RT), 0, stream, x, y, count / 4, act, slope);
    return launch_status();
}

int gz_axpby(const float* a, float alpha, const float* b, float beta, float* out, float* act_out, long long count,
             int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    if (count <= 0 || (count & 3)) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(axpby_kernel, dim3(rn_grid(count / 4)), dim3(RT), 0, stream, a, alpha, b, beta, out, act_out,
                       count / 4, act, slope);
    return launch_status();
}

static int pool_shape_ok(long long planes, int H, int W, int OH, int OW) {
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    if (OH != (H - 1) / 2 + 1 || OW != (W - 1) / 2 + 1) return 0;
    return planes * H * W < (1ll << 31);
}

int gz_avgpool3s2_fwd(const float* x, float* y, long long planes, int H, int W, int OH, int OW,
                      hipStream_t stream) {
    gz::clear_stale_error();
    if (!pool_shape_ok(planes, H, W, OH, OW)) return GZ_ERR_BAD_SHAPE;
    long long total = planes * OH * OW;
    hipLaunchKernelGGL(avgpool3s2_fwd_kernel, dim3(rn_grid(total)), dim3(RT), 0, stream, x, y, total, H, W, OH, OW,
                       make_fastdiv(OW), make_fastdiv(OH));
    return launch_status();
}

int gz_avgpool3s2_bwd(const float* gy, float* gx, long long planes, int H, int W, int OH, int OW,
                      hipStream_t stream) {
    gz::clear_stale_error();
    if (!pool_shape_ok(planes, H, W, OH, OW)) return GZ_ERR_BAD_SHAPE;
    long long total = planes * H * W;
    hipLaunchKernelGGL(avgpool3s2_bwd_kernel, dim3(rn_grid(total)), dim3(RT), 0, stream, gy, gx, total, H, W, OH, OW,
                       make_fastdiv(W), make_fastdiv(H));
    return launch_status();
}

int gz_upsample2_fwd(const float* x, float* y, long long planes, int H, int W, hipStream_t stream) {
    gz::clear_stale_error();
    if (planes <= 0 || H <= 0 || W <= 0 || planes * H * W * 4 >= (1ll << 31)) return GZ_ERR_BAD_SHAPE;
    long long total = planes * H * W;
    hipLaunchKernelGGL(upsample2_fwd_kernel, dim3(rn_grid(total)), dim3(RT), 0, stream, x, y, total, W,
                       make_fastdiv(W));
    return launch_status();
}

int gz_upsample2_bwd(const float* gy, float* gx, long long planes, int H, int W, hipStream_t stream) {
    gz::clear_stale_error();
    if (planes <= 0 || H <= 0 || W <= 0 || planes * H * W * 4 >= (1ll << 31)) return GZ_ERR_BAD_SHAPE;
    long long total = planes * H * W;
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(rn_grid(total)), dim3(RT), 0, stream, gy, gx, total, W,
                       make_fastdiv(W));
    return launch_status();
}

}  // extern "C"

// ---------------------------------------------------------------------------
// input step: decoded uint8 HWC images -> normalised float NCHW on the device, i.e.
// ToTensor() + Normalize(mean, std) of reference core/lightning_module.py:42-47 in one HBM pass (1 B read, 4 B
// written per element) instead of on the host.  One thread per output pixel, all channels.
// ---------------------------------------------------------------------------
namespace gz {
__global__ __launch_bounds__(256) void u8hwc_to_nchw_kernel(const unsigned char* __restrict__ in,
                                                            float* __restrict__ out, long long pixels, int HW, int C,
                                                            float scale, float shift, FastDiv div_hw) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += stride) {
        const uint32_t n = fdiv((uint32_t)p, div_hw);
        const uint32_t pix = (uint32_t)p - n * (uint32_t)HW;
        const unsigned char* src = in + p * C;
        float* dst = out + (long long)n * C * HW + pix;
        for (int c = 0; c < C; ++c) dst[(long long)c * HW] = (float)src[c] * scale + shift;
    }
}
}  // namespace gz

extern "C" int gz_u8hwc_to_nchw(const unsigned char* in, float* out, int N, int H, int W, int C, float mean, float std,
                                hipStream_t stream) {
    gz::clear_stale_error();
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 4 || std == 0.f) return GZ_ERR_BAD_SHAPE;
    const long long pixels = (long long)N * H * W;
    if (pixels >= (1ll << 31)) return GZ_ERR_TOO_LARGE;
    // (x / 255 - mean) / std  =  x * (1 / (255 std)) - mean / std
    hipLaunchKernelGGL(gz::u8hwc_to_nchw_kernel, dim3(gz::rn_grid(pixels)), dim3(256), 0, stream, in, out, pixels, H * W,
                       C, 1.f / (255.f * std), -mean / std, gz::make_fastdiv(H * W));
    return gz::launch_status();
}
