// Small fused tails of the training step that the reference spells as chains of tiny torch operators:
//   * BCE-with-logits against a constant target, mean-reduced (core/lightning_module.py:114-119,126,221-235:
//     criterion(logits, ones_like / zeros_like)) and its gradient;
//   * mean squared error (HoloGAN's q_loss, :226,234) and its gradient;
//   * the scalar side of torch.nn.utils.spectral_norm (core/models/hologan_discriminator.py:15): vector
//     normalisation of the power iteration, sigma, weight / sigma, and the gradient of that quotient.
// All are a handful of values to a few MB; one launch each instead of 4..14 framework launches (HoloGAN's optimizer
// cycle spent 42 % of its launches in such operators).  Scalars stay on the device (no host synchronisation).
#include "gz_common.h"
#include "../../include/gz_ops.h"

namespace gz {

constexpr int LT = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {      // red: >= 4 floats of LDS
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// loss = mean_i( max(x,0) - x*t + log1p(exp(-|x|)) )   (torch's binary_cross_entropy_with_logits formula)
__global__ __launch_bounds__(LT) void bce_logits_mean_kernel(const float* __restrict__ x, float* __restrict__ loss,
                                                            int n, float t) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += LT) {
        const float v = x[i];
        s += (fmaxf(v, 0.f) - v * t) + log1pf(expf(-fabsf(v)));
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) loss[0] = s / (float)n;
}

// dx = (sigmoid(x) - t) * g / n
__global__ __launch_bounds__(LT) void bce_logits_mean_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                float* __restrict__ dx, int n, float t) {
    const float scale = g[0] / (float)n;
    for (int i = blockIdx.x * LT + threadIdx.x; i < n; i += gridDim.x * LT) {
        const float v = x[i];
        dx[i] = (1.f / (1.f + expf(-v)) - t) * scale;
    }
}

// Loss head of a discriminator step on the stacked batch [real; fake] (round 4): logits x[0..n) against t0 and
// x[n..2n) against t1 in ONE launch.
//   mode 0: (mean BCE(x[:n], t0) + mean BCE(x[n:], t1)) / 2        (DCGAN, core/lightning_module.py:114-120)
//   mode 1: t0 * mean(x[:n]) + t1 * mean(x[n:])                     (WGAN critic loss with t0 = -1, t1 = +1, :168)
__global__ __launch_bounds__(LT) void pair_loss_kernel(const float* __restrict__ x, float* __restrict__ loss, int n,
                                                       float t0, float t1, int mode) {
    __shared__ float red[4];
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < n; i += LT) {
        const float a = x[i], b = x[n + i];
        if (mode == 0) {
            s0 += (fmaxf(a, 0.f) - a * t0) + log1pf(expf(-fabsf(a)));
            s1 += (fmaxf(b, 0.f) - b * t1) + log1pf(expf(-fabsf(b)));
        } else {
            s0 += a;
            s1 += b;
        }
    }
    s0 = block_sum(s0, red);
    s1 = block_sum(s1, red);
    if (threadIdx.x == 0)
        loss[0] = mode == 0 ? (s0 / (float)n + s1 / (float)n) / 2.f : t0 * (s0 / (float)n) + t1 * (s1 / (float)n);
}

__global__ __launch_bounds__(LT) void pair_loss_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           float* __restrict__ dx, int n, float t0, float t1, int mode) {
    const float gs = g[0] / (float)n;
    for (int i = blockIdx.x * LT + threadIdx.x; i < 2 * n; i += gridDim.x * LT) {
        const float t = i < n ? t0 : t1;
        dx[i] = mode == 0 ? (1.f / (1.f + expf(-x[i])) - t) * (gs * 0.5f) : t * gs;
    }
}

__global__ __launch_bounds__(LT) void mse_mean_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     float* __restrict__ loss, int n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += LT) {
        const float d = a[i] - b[i];
        s += d * d;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) loss[0] = s / (float)n;
}

// da = 2 (a - b) g / n   (db = -da, taken by the caller when b needs a gradient)
__global__ __launch_bounds__(LT) void mse_mean_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ g, float* __restrict__ da, int n) {
    const float scale = 2.f * g[0] / (float)n;
    for (int i = blockIdx.x * LT + threadIdx.x; i < n; i += gridDim.x * LT) da[i] = (a[i] - b[i]) * scale;
}

// WGAN-GP tail (reference core/utils/utils.py:55-57): gradient_norm = ||g_n||_2 per sample, penalty = mean((norm - 1)^2),
// from the per-sample sums of squares s_n.  torch.norm's subgradient at an exactly-zero vector is 0, so d/ds_n =
// (sqrt(s_n) - 1) / (sqrt(s_n) N) for s_n > 0 and 0 at s_n == 0.  One launch each way instead of ~8 framework operators
// (compare, where, sqrt, mul, sub, pow, mean and their autograd mirrors).
__global__ __launch_bounds__(LT) void gp_penalty_kernel(const float* __restrict__ sumsq, float* __restrict__ out, int n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += LT) {
        const float d = sqrtf(sumsq[i]) - 1.f;
        s += d * d;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

__global__ __launch_bounds__(LT) void gp_penalty_bwd_kernel(const float* __restrict__ sumsq, const float* __restrict__ g,
                                                           float* __restrict__ ds, int n) {
    const float scale = g[0] / (float)n;
    for (int i = blockIdx.x * LT + threadIdx.x; i < n; i += gridDim.x * LT) {
        const float r = sqrtf(sumsq[i]);
        ds[i] = r > 0.f ? (r - 1.f) / r * scale : 0.f;
    }
}

// out = out2 = x / max(||x||, eps);  norm_out[0] = ||x||.  One workgroup (n <= a few thousand).  `out` may be the
// module's persistent buffer (updated in place like torch's spectral_norm does) and `out2` the private copy the
// autograd node keeps; either may be NULL.
// `dot_out` (optional): <normalised x, x>, i.e. spectral_norm's sigma = u^T (W v) when x = W v -- the separate
// vec_dot launch of the power iteration folded in.
__global__ __launch_bounds__(LT) void vec_normalize_kernel(const float* __restrict__ x, float* out, float* out2,
                                                          float* __restrict__ norm_out, float* __restrict__ dot_out,
                                                          int n, float eps) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += LT) s += x[i] * x[i];
    s = block_sum(s, red);
    const float nrm = sqrtf(s);
    const float inv = 1.f / fmaxf(nrm, eps);
    float d = 0.f;
    for (int i = threadIdx.x; i < n; i += LT) {
        const float xi = x[i];      // read first: `out` may be x itself
        const float q = xi * inv;
        if (out) out[i] = q;
        if (out2) out2[i] = q;
        d += q * xi;
    }
    if (threadIdx.x == 0 && norm_out) norm_out[0] = nrm;
    if (dot_out) {
        d = block_sum(d, red);
        if (threadIdx.x == 0) dot_out[0] = d;
    }
}

// sigma[0] = <a, b>   (one workgroup)
__global__ __launch_bounds__(LT) void vec_dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    float* __restrict__ out, int n) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += LT) s += a[i] * b[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = s;
}

// out = x / sigma[0]
__global__ __launch_bounds__(LT) void div_scalar_kernel(const float* __restrict__ x, const float* __restrict__ sigma,
                                                       float* __restrict__ out, long long total4) {
    const float inv = 1.f / sigma[0];
    const long long stride = (long long)gridDim.x * LT;
    for (long long i = (long long)blockIdx.x * LT + threadIdx.x; i < total4; i += stride) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

// gradient of w = W / sigma(W), sigma = u^T W v with u, v held constant:
//   dW[r][l] = (G[r][l] - c * u[r] * v[l]) / sigma,   c = sum_r rowdots[r],  rowdots[r] = <G[r], w[r]>
__global__ __launch_bounds__(LT) void sn_bwd_kernel(const float* __restrict__ G, const float* __restrict__ rowdots,
                                                   const float* __restrict__ u, const float* __restrict__ v,
                                                   const float* __restrict__ sigma, float* __restrict__ out, int R,
                                                   int L4, long long total4, FastDiv div_l4) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < R; i += LT) s += rowdots[i];
    const float c = block_sum(s, red);
    const float inv = 1.f / sigma[0];
    const long long stride = (long long)gridDim.x * LT;
    for (long long i = (long long)blockIdx.x * LT + threadIdx.x; i < total4; i += stride) {
        const uint32_t r = fdiv((uint32_t)i, div_l4);
        const uint32_t q = (uint32_t)i - r * (uint32_t)L4;
        const float cu = c * u[r];
        const f32x4 g = reinterpret_cast<const f32x4*>(G)[i];
        const f32x4 vv = reinterpret_cast<const f32x4*>(v)[q];
        f32x4 o;
        o.x = (g.x - cu * vv.x) * inv; o.y = (g.y - cu * vv.y) * inv;
        o.z = (g.z - cu * vv.z) * inv; o.w = (g.w - cu * vv.w) * inv;
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

// ---- spectral normalisation's power iteration for several weights, four launches -----------------------------
// One iteration is v = normalise(W^T u), u = normalise(W v), sigma = u^T W v (torch.nn.utils.spectral_norm, used by
// core/models/hologan_discriminator.py:15,32 on three convolutions).  As separate launches that was coldot, slab sum,
// normalise, rowdot, normalise+dot PER LAYER and discriminator call -- launch-bound ~6 us kernels, 15 per call.  Here
// every layer of the call is a job of the same four launches.  (Folding the single-workgroup tails into the last
// workgroup to arrive at a ticket was built and measured first: the agent-scope release / acquire fences that make
// other XCDs' partial results visible write back and invalidate the whole L2 per workgroup -- 86 + 71 us for the two
// launches against 100 us for the fifteen they replaced.  Kernel boundaries are the cheaper barrier.)
constexpr int SN_MAX_JOBS = 4;
constexpr int SN_SLICES = 32;
struct SnJob {
    const float* W;        // [R][L] weight_orig
    float* u;              // [R] module buffer, updated in place
    float* v;              // [L] module buffer, updated in place
    float* us;             // [R] the autograd node's copy
    float* vs;             // [L]
    float* sigma;          // [1]
    float* slabs;          // [slices][L] workspace
    float* vraw;           // [L] workspace: W^T u before normalisation
    float* wv;             // [R] workspace: W v
    float* norm2;          // [colblocks] workspace: sum of squares of vraw per column block
    int R, L, slices, rps, blockA0, blocksA, blockC0, blocksC, blockR0, blocksR;
};
struct SnTable {
    int njobs, pad;
    SnJob jobs[SN_MAX_JOBS];
};

// slab[z][l] = sum_{r in slice z} u[r] * W[r][l]; grid per job: colblocks x slices
__global__ __launch_bounds__(LT) void sn_coldot_kernel(SnTable t) {
    int j = 0;
    while (j + 1 < t.njobs && t.jobs[j + 1].blockA0 <= (int)blockIdx.x) ++j;
    const SnJob& jb = t.jobs[j];
    const int L4 = jb.L >> 2, colblocks = (L4 + LT - 1) / LT;
    const int lb = blockIdx.x - jb.blockA0, cb = lb % colblocks, z = lb / colblocks;
    const int q = cb * LT + threadIdx.x;
    if (q >= L4) return;
    const int r0 = z * jb.rps, r1 = min(jb.R, r0 + jb.rps);
    const f32x4* pw = reinterpret_cast<const f32x4*>(jb.W) + q;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int r = r0;
    for (; r + 4 <= r1; r += 4) {            // four rows in flight, added in row order
        const f32x4 w0 = pw[(long long)r * L4], w1 = pw[(long long)(r + 1) * L4];
        const f32x4 w2 = pw[(long long)(r + 2) * L4], w3 = pw[(long long)(r + 3) * L4];
        acc = acc + w0 * jb.u[r];
        acc = acc + w1 * jb.u[r + 1];
        acc = acc + w2 * jb.u[r + 2];
        acc = acc + w3 * jb.u[r + 3];
    }
    for (; r < r1; ++r) acc = acc + pw[(long long)r * L4] * jb.u[r];
    reinterpret_cast<f32x4*>(jb.slabs)[(long long)z * L4 + q] = acc;
}

// vraw = sum of the slabs in slice order (one float4 column per lane), norm2[cb] = the column block's sum of squares
__global__ __launch_bounds__(LT) void sn_vsum_kernel(SnTable t) {
    __shared__ float red[4];
    int j = 0;
    while (j + 1 < t.njobs && t.jobs[j + 1].blockC0 <= (int)blockIdx.x) ++j;
    const SnJob& jb = t.jobs[j];
    const int L4 = jb.L >> 2, cb = blockIdx.x - jb.blockC0;
    const int q = cb * LT + threadIdx.x;
    float s = 0.f;
    if (q < L4) {
        const f32x4* p = reinterpret_cast<const f32x4*>(jb.slabs) + q;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        int sl = 0;
        for (; sl + 8 <= jb.slices; sl += 8) {
            f32x4 tv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) tv[k] = p[(long long)(sl + k) * L4];
#pragma unroll
            for (int k = 0; k < 8; ++k) a = a + tv[k];
        }
        for (; sl < jb.slices; ++sl) a = a + p[(long long)sl * L4];
        reinterpret_cast<f32x4*>(jb.vraw)[q] = a;
        s = (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) jb.norm2[cb] = s;
}

// wv[r] = <W[r], v> with v = vraw / max(||vraw||, eps) formed on the fly; one workgroup per row, which also writes
// its share of v to the module buffer and to the node's copy
__global__ __launch_bounds__(LT) void sn_rowdot_kernel(SnTable t, float eps) {
    __shared__ float red[4];
    int j = 0;
    while (j + 1 < t.njobs && t.jobs[j + 1].blockR0 <= (int)blockIdx.x) ++j;
    const SnJob& jb = t.jobs[j];
    const int L4 = jb.L >> 2, colblocks = (L4 + LT - 1) / LT, b = blockIdx.x - jb.blockR0;
    float tot = 0.f;
    for (int c = 0; c < colblocks; ++c) tot += jb.norm2[c];
    const float inv = 1.f / fmaxf(sqrtf(tot), eps);
    const f32x4* pv = reinterpret_cast<const f32x4*>(jb.vraw);
    const int share = (L4 + jb.blocksR - 1) / jb.blocksR;
    for (int c = b * share + threadIdx.x; c < min(L4, (b + 1) * share); c += LT) {
        const f32x4 a = pv[c] * inv;
        reinterpret_cast<f32x4*>(jb.vs)[c] = a;
        reinterpret_cast<f32x4*>(jb.v)[c] = a;
    }
    for (int r = b; r < jb.R; r += jb.blocksR) {
        const f32x4* pa = reinterpret_cast<const f32x4*>(jb.W) + (long long)r * L4;
        float s0 = 0.f, s1 = 0.f;
        int q = threadIdx.x;
        for (; q + LT < L4; q += 2 * LT) {
            const f32x4 a = pa[q], w = pv[q] * inv, a2 = pa[q + LT], w2 = pv[q + LT] * inv;
            s0 += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
            s1 += (a2.x * w2.x + a2.y * w2.y) + (a2.z * w2.z + a2.w * w2.w);
        }
        for (; q < L4; q += LT) {
            const f32x4 a = pa[q], w = pv[q] * inv;
            s0 += (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
        }
        const float sr = block_sum(s0 + s1, red);
        if (threadIdx.x == 0) jb.wv[r] = sr;
    }
}

// u = wv / max(||wv||, eps), sigma = <u, wv>; one workgroup per job
__global__ __launch_bounds__(LT) void sn_unorm_kernel(SnTable t, float eps) {
    __shared__ float red[4];
    const SnJob& jb = t.jobs[blockIdx.x];
    float s = 0.f;
    for (int i = threadIdx.x; i < jb.R; i += LT) {
        const float x = jb.wv[i];
        s += x * x;
    }
    s = block_sum(s, red);
    const float inv = 1.f / fmaxf(sqrtf(s), eps);
    float d = 0.f;
    for (int i = threadIdx.x; i < jb.R; i += LT) {
        const float x = jb.wv[i];
        const float qv = x * inv;
        jb.u[i] = qv;
        jb.us[i] = qv;
        d += qv * x;
    }
    d = block_sum(d, red);
    if (threadIdx.x == 0) jb.sigma[0] = d;
}

// ---- a spectral-normalised convolution followed by InstanceNorm without the per-call weight copy ------------------
// sigma enters such a block only as a scalar on the convolution's output, and the InstanceNorm behind it removes any
// scale except through eps: IN_eps(conv(x, W / sigma) + b) = IN_{eps sigma^2}(conv(x, W)) (b is removed by the mean).
// The gradient's sigma term.  With the row sums S2 = sum_i dz_i xh_i of the InstanceNorm backward,
//   dL/dsigma_g = -eps sigma_g sum_{rows of group g} rstd_row^2 S2_row     (= -(sum g_raw . y_raw) / sigma_g exactly:
//   the "- c u v^T / sigma" term of torch's spectral_norm backward), and d sigma / dW = u v^T.
// coefs[g][s] <- the share of slice s (SN_COEF_SLICES workgroups per group, fixed order inside and across the slices)
constexpr int SN_COEF_SLICES = 16;
__global__ __launch_bounds__(LT) void sn_sigma_coef_kernel(const f32x2* __restrict__ rowsums, const float* __restrict__ rstd,
                                                          const float* __restrict__ sigma, float* __restrict__ coefs,
                                                          int rows_per_group, float eps) {
    __shared__ float red[4];
    const int g = blockIdx.x, sl = blockIdx.y;
    const int per = (rows_per_group + SN_COEF_SLICES - 1) / SN_COEF_SLICES;
    const int i0 = sl * per, i1 = min(rows_per_group, i0 + per);
    float s = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += LT) {
        const int r = g * rows_per_group + i;
        const float rs = rstd[r];
        s += rs * rs * rowsums[r].y;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) coefs[g * SN_COEF_SLICES + sl] = -eps * sigma[g] * s;
}

// term[r][l] = sum_g coef_g u[g][r] v[g][l],  coef_g = the sum of the group's slices
__global__ __launch_bounds__(LT) void sn_sigma_term_kernel(const float* __restrict__ coefs, const float* __restrict__ u,
                                                          const float* __restrict__ v, float* __restrict__ term, int R,
                                                          int L4, int groups, long long total4, FastDiv div_l4) {
    float cg[8];
    for (int g = 0; g < groups; ++g) {
        float c = 0.f;
        for (int sl = 0; sl < SN_COEF_SLICES; ++sl) c += coefs[g * SN_COEF_SLICES + sl];
        cg[g] = c;
    }
    const long long stride = (long long)gridDim.x * LT;
    for (long long i = (long long)blockIdx.x * LT + threadIdx.x; i < total4; i += stride) {
        const uint32_t r = fdiv((uint32_t)i, div_l4);
        const uint32_t q = (uint32_t)i - r * (uint32_t)L4;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < groups; ++g) {
            const float cu = cg[g] * u[(long long)g * R + r];
            const f32x4 vv = reinterpret_cast<const f32x4*>(v)[(long long)g * L4 + q];
            o = o + vv * cu;
        }
        reinterpret_cast<f32x4*>(term)[i] = o;
    }
}

static int grid_for(long long items) {
    long long b = (items + LT - 1) / LT;
    if (b > 2048) b = 2048;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace gz

using namespace gz;

extern "C" {

int gz_bce_logits_mean(const float* x, float* loss, int n, float target, hipStream_t stream) {
    gz::clear_stale_error();
    if (n <= 0) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(bce_logits_mean_kernel, dim3(1), dim3(LT), 0, stream, x, loss, n, target);
    return launch_status();
}

int gz_bce_logits_mean_bwd(const float* x, const float* gloss, float* dx, int n, float target, hipStream_t stream) {
    gz::clear_stale_error();
    if (n <= 0) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(bce_logits_mean_bwd_kernel, dim3(grid_for(n)), dim3(LT), 0, stream, x, gloss, dx, n, target);
    return launch_status();
}

int gz_pair_loss(const float* x, float* loss, int n_each, float t0, float t1, int mode, hipStream_t stream) {
    gz::clear_stale_error();
    if (n_each <= 0 || mode < 0 || mode > 1) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(pair_loss_kernel, dim3(1), dim3(LT), 0, stream, x, loss, n_each, t0, t1, mode);
    return launch_status();
}

int gz_pair_loss_bwd(const float* x, const float* gloss, float* dx, int n_each, float t0, float t1, int mode,
                     hipStream_t stream) {
    gz::clear_stale_error();
    if (n_each <= 0 || mode < 0 || mode > 1) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(pair_loss_bwd_kernel, dim3(grid_for(2 * n_each)), dim3(LT), 0, stream, x, gloss, dx, n_each, t0,
                       t1, mode);
    return launch_status();
}

int gz_mse_mean(const float* a, const float* b, float* loss, int n, hipStream_t stream) {
    gz::clear_stale_error();
    if (n <= 0) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(mse_mean_kernel, dim3(1), dim3(LT), 0, stream, a, b, loss, n);
    return launch_status();
}

int gz_mse_mean_bwd(const float* a, const float* b, const float* gloss, float* da, int n, hipStream_t stream) {
    gz::clear_stale_error();
    if (n <= 0) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(mse_mean_bwd_kernel, dim3(grid_for(n)), dim3(LT), 0, stream, a, b, gloss, da, n);
    return launch_status();
}

int gz_gp_penalty(const float* sumsq, float* out, int n, hipStream_t stream) {
    gz::clear_stale_error();
    if (n <= 0 || !sumsq || !out) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(gp_penalty_kernel, dim3(1), dim3(LT), 0, stream, sumsq, out, n);
    return launch_status();
}

int gz_gp_penalty_bwd(const float* sumsq, const float* gout, float* dsumsq, int n, hipStream_t stream) {
    gz::clear_stale_error();
    if (n <= 0 || !sumsq || !gout || !dsumsq) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(gp_penalty_bwd_kernel, dim3(grid_for(n)), dim3(LT), 0, stream, sumsq, gout, dsumsq, n);
    return launch_status();
}

int gz_sn_max_jobs(void) { return SN_MAX_JOBS; }
size_t gz_sn_table_bytes(void) { return sizeof(SnTable); }
static int sn_colblocks(int L) { return (L / 4 + LT - 1) / LT; }
long long gz_sn_workspace_floats(int R, int L) {
    if (R <= 0 || L <= 0) return 0;
    return (long long)(R >= 64 ? SN_SLICES : 1) * L + L + ((R + 3) & ~3) + sn_colblocks(L);
}

int gz_sn_add(void* table_host, const float* W, float* u, float* v, float* us, float* vs, float* sigma, float* workspace,
              int R, int L) {
    SnTable* t = reinterpret_cast<SnTable*>(table_host);
    if (!t || !W || !u || !v || !us || !vs || !sigma || !workspace || R <= 0 || L <= 0 || (L & 3)) return GZ_ERR_BAD_SHAPE;
    if ((((uintptr_t)W | (uintptr_t)v | (uintptr_t)vs | (uintptr_t)workspace) & 15)) return GZ_ERR_BAD_SHAPE;
    if (t->njobs < 0 || t->njobs >= SN_MAX_JOBS) return GZ_ERR_UNSUPPORTED;
    SnJob& jb = t->jobs[t->njobs++];
    const int slices = R >= 64 ? SN_SLICES : 1;           // (gz_coldot's slicing)
    float* vraw = workspace + (long long)slices * L;
    float* wv = vraw + L;
    jb = SnJob{W, u, v, us, vs, sigma, workspace, vraw, wv, wv + ((R + 3) & ~3), R, L, slices, (R + slices - 1) / slices,
               0, 0, 0, 0, 0, 0};
    return GZ_OK;
}

int gz_sn_power_iteration(void* table_host, float eps, hipStream_t stream) {
    gz::clear_stale_error();
    SnTable* t = reinterpret_cast<SnTable*>(table_host);
    if (!t || t->njobs <= 0 || t->njobs > SN_MAX_JOBS) return GZ_ERR_BAD_SHAPE;
    int a = 0, c = 0, r = 0;
    for (int j = 0; j < t->njobs; ++j) {
        SnJob& jb = t->jobs[j];
        jb.blockA0 = a;
        jb.blocksA = sn_colblocks(jb.L) * jb.slices;
        a += jb.blocksA;
        jb.blockC0 = c;
        jb.blocksC = sn_colblocks(jb.L);
        c += jb.blocksC;
        jb.blockR0 = r;
        jb.blocksR = jb.R > 2048 ? 2048 : jb.R;
        r += jb.blocksR;
    }
    hipLaunchKernelGGL(sn_coldot_kernel, dim3(a), dim3(LT), 0, stream, *t);
    hipLaunchKernelGGL(sn_vsum_kernel, dim3(c), dim3(LT), 0, stream, *t);
    hipLaunchKernelGGL(sn_rowdot_kernel, dim3(r), dim3(LT), 0, stream, *t, eps);
    hipLaunchKernelGGL(sn_unorm_kernel, dim3(t->njobs), dim3(LT), 0, stream, *t, eps);
    return launch_status();
}

int gz_sn_sigma_coef_floats(int groups) { return groups > 0 ? groups * SN_COEF_SLICES : 0; }

int gz_sn_sigma_term(const float* rowsums, const float* rstd, const float* sigma, const float* u, const float* v,
                     float* coefs, float* term, int rows, int groups, int R, int L, float eps, hipStream_t stream) {
    gz::clear_stale_error();
    if (!rowsums || !rstd || !sigma || !u || !v || !coefs || !term || rows <= 0 || groups <= 0 || groups > 8 || rows % groups ||
        R <= 0 || L <= 0 || (L & 3))
        return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(sn_sigma_coef_kernel, dim3(groups, SN_COEF_SLICES), dim3(LT), 0, stream, reinterpret_cast<const f32x2*>(rowsums), rstd,
                       sigma, coefs, rows / groups, eps);
    const long long total4 = (long long)R * (L / 4);
    hipLaunchKernelGGL(sn_sigma_term_kernel, dim3(grid_for(total4)), dim3(LT), 0, stream, coefs, u, v, term, R, L / 4, groups,
                       total4, make_fastdiv(L / 4));
    return launch_status();
}

int gz_vec_normalize(const float* x, float* out, float* out2, float* norm_out, int n, float eps, hipStream_t stream) {
    gz::clear_stale_error();
    if (n <= 0) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(vec_normalize_kernel, dim3(1), dim3(LT), 0, stream, x, out, out2, norm_out, (float*)nullptr, n,
                       eps);
    return launch_status();
}

int gz_vec_normalize_dot(const float* x, float* out, float* out2, float* dot_out, int n, float eps,
                         hipStream_t stream) {
    gz::clear_stale_error();
    if (n <= 0 || !dot_out) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(vec_normalize_kernel, dim3(1), dim3(LT), 0, stream, x, out, out2, (float*)nullptr, dot_out, n,
                       eps);
    return launch_status();
}

int gz_vec_dot(const float* a, const float* b, float* out, int n, hipStream_t stream) {
    gz::clear_stale_error();
    if (n <= 0) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(vec_dot_kernel, dim3(1), dim3(LT), 0, stream, a, b, out, n);
    return launch_status();
}

int gz_div_scalar(const float* x, const float* sigma, float* out, long long count, hipStream_t stream) {
    gz::clear_stale_error();
    if (count <= 0 || (count & 3)) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(div_scalar_kernel, dim3(grid_for(count / 4)), dim3(LT), 0, stream, x, sigma, out, count / 4);
    return launch_status();
}

int gz_spectral_norm_bwd(const float* g, const float* rowdots, const float* u, const float* v, const float* sigma,
                         float* out, int R, int L, hipStream_t stream) {
    gz::clear_stale_error();
    if (R <= 0 || L <= 0 || (L & 3)) return GZ_ERR_BAD_SHAPE;
    const long long total4 = (long long)R * (L / 4);
    hipLaunchKernelGGL(sn_bwd_kernel, dim3(grid_for(total4)), dim3(LT), 0, stream, g, rowdots, u, v, sigma, out, R, L / 4,
                       total4, make_fastdiv(L / 4));
    return launch_status();
}

}  // extern "C"
