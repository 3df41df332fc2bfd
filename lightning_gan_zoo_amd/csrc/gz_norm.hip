// Normalisation + activation kernels (HBM-bound): BatchNorm2d(train/eval), InstanceNorm2d
// (affine or not) fused with ReLU / LeakyReLU, their backward, and the double backward the
// WGAN-GP gradient penalty needs (reference core/utils/utils.py:39-58 differentiates through
// InstanceNorm's backward; standard_networks.py:44-47,87-88 are the call sites).
//
// Data model: a tensor [N, C, inner] is a set of rows = N*C of `inner` contiguous floats.
// Statistics are per row (InstanceNorm / AdaIN) or per channel over all n (BatchNorm):
// every pass is "row sums by wavefront shuffle" (+ a tiny finalize over rows) or a float4
// elementwise apply, so each tensor is streamed once per pass with 16-byte accesses.
// `per_channel` = 1 selects coefficient index row % C (BatchNorm), 0 selects the row itself.
#include "gz_common.h"
#include "gz_knobs.h"
#include "../../include/gz_ops.h"

namespace gz {

constexpr int PW_THREADS = 256;

struct RowGeom {
    int rows, q4;       // rows, float4 per row
    int lpr, rpw;       // lanes per row (pow2 <= 64), rows per wave
};

static RowGeom row_geom(long long rows, int inner) {
    RowGeom g;
    g.rows = (int)rows;
    g.q4 = inner / 4;
    int lpr = 1;
    while (lpr < 64 && lpr < g.q4) lpr <<= 1;
    g.lpr = lpr;
    g.rpw = 64 / lpr;
    return g;
}

static int row_grid(const RowGeom& g) {
    long long groups = ((long long)g.rows + g.rpw - 1) / g.rpw;   // wave-sized work items
    long long blocks = (groups + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int NS>
__device__ __forceinline__ void sub_reduce(float (&s)[NS], int lpr) {
    for (int o = lpr >> 1; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] += __shfl_xor(s[k], o, 64);
    }
}

// ---------------------------------------------------------------------------
// forward statistics
// ---------------------------------------------------------------------------
// sums[row] = (sum x, sum x^2)
__global__ __launch_bounds__(PW_THREADS) void row_sums_kernel(const float* __restrict__ x, f32x2* __restrict__ sums,
                                                              RowGeom g) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * PW_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * PW_THREADS) >> 6;
    const int sub = lane / g.lpr, l = lane % g.lpr;
    for (long long row0 = (long long)wave * g.rpw; row0 < g.rows; row0 += (long long)nwaves * g.rpw) {
        long long row = row0 + sub;
        float s[2] = {0.f, 0.f};
        if (row < g.rows) {
            const f32x4* p = reinterpret_cast<const f32x4*>(x) + row * g.q4;
            for (int q = l; q < g.q4; q += g.lpr) {
                f32x4 v = p[q];
                s[0] += (v.x + v.y) + (v.z + v.w);
                s[1] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
        }
        sub_reduce<2>(s, g.lpr);
        if (row < g.rows && l == 0) sums[row] = f32x2{s[0], s[1]};
    }
}

// out[c] = sum_n sums[n*C + c].x : second half of gz_channel_sum (bias gradient of a convolution)
__global__ __launch_bounds__(64) void channel_sum_finalize_kernel(const f32x2* __restrict__ sums, float* __restrict__ out,
                                                                  int N, int C) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int n = lane; n < N; n += 64) s += (double)sums[(long long)n * C + c].x;
    s = wave_sum_d(s);
    if (lane == 0) out[c] = (float)s;
}

// BatchNorm finalize: one wavefront per channel; lanes stride over n, fp64 shuffle-combine.
// coef[0*C..] = scale, coef[1*C..] = shift, coef[2*C..] = mean, coef[3*C..] = rstd
// groups > 1 (round 4): the batch is `groups` independent statistics groups stacked along n -- the discriminator
// applied to [real; fake] in ONE pass behaves like the reference's two calls: group g owns rows [g N/groups,
// (g+1) N/groups), gets its own coefficients (index g*C + c, `count` elements per group) and updates the running
// buffers in turn, in group order (num_batches_tracked += groups).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void bn_finalize_kernel(const f32x2* __restrict__ sums,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ coef,
                                                         float* running_mean, float* running_var, long long* nbt,
                                                         int N, int C, int inner, float eps, float momentum,
                                                         double count, int groups) {
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c == 0 && lane == 0 && nbt) *nbt += groups;
    const int Ng = N / groups, NC = groups * C;
    for (int g = 0; g < groups; ++g) {
        double s1 = 0.0, s2 = 0.0;
        for (int n = lane; n < Ng; n += 64 * WAVES) {
            f32x2 v = sums[(long long)(g * Ng + n) * C + c];
            s1 += (double)v.x;
            s2 += (double)v.y;
        }
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
        if constexpr (WAVES > 1) {       // many partial rows (per-tile statistics of a large layer): 4 wavefronts share the walk
            __shared__ double part[WAVES][2];
            __syncthreads();             // (the previous group's readers are done)
            if ((lane & 63) == 0) {
                part[lane >> 6][0] = s1;
                part[lane >> 6][1] = s2;
            }
            __syncthreads();
            s1 = s2 = 0.0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                s1 += part[w][0];
                s2 += part[w][1];
            }
        }
        if (lane != 0) continue;
        double cnt = count > 0.0 ? count : (double)Ng * inner;      // count > 0: `sums` are per-tile partials, not rows
        double mean = s1 / cnt;
        double var = s2 / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        float rstd = (float)(1.0 / sqrt(var + (double)eps));
        float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
        float scale = ga * rstd;
        const int ci = g * C + c;
        coef[ci] = scale;
        coef[NC + ci] = be - (float)mean * scale;
        coef[2 * NC + ci] = (float)mean;
        coef[3 * NC + ci] = rstd;
        if (running_mean) {
            double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// BatchNorm eval: coefficients from the running statistics
__global__ void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ rm, const float* __restrict__ rv,
                                    float* __restrict__ coef, int C, float eps) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float rstd = 1.f / sqrtf(rv[c] + eps);
    float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    coef[c] = ga * rstd;
    coef[C + c] = be - rm[c] * ga * rstd;
    coef[2 * C + c] = rm[c];
    coef[3 * C + c] = rstd;
}

// ---------------------------------------------------------------------------
// forward apply: out = act(x * scale[i] + shift[i])
// ---------------------------------------------------------------------------
struct ApplyGeom {
    long long total4;
    int q4;             // float4 per row
    FastDiv div_q4, div_c;
    int C, per_channel, ncoef;   // ncoef = number of coefficient entries (groups * C or rows)
    int group_rows;              // per_channel with statistics groups: rows (n, c) per group = (N / groups) * C; 0 = one group
    FastDiv div_grows;
};

__device__ __forceinline__ int coef_index(long long i4, const ApplyGeom& g) {
    uint32_t row = fdiv((uint32_t)i4, g.div_q4);
    if (g.per_channel) {
        const uint32_t grp = g.group_rows ? fdiv(row, g.div_grows) : 0u;
        row = row - fdiv(row, g.div_c) * (uint32_t)g.C + grp * (uint32_t)g.C;
    }
    return (int)row;
}

__global__ __launch_bounds__(PW_THREADS) void norm_act_apply_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ coef,
                                                                    float* __restrict__ out, ApplyGeom g, int act,
                                                                    float slope) {
    const long long stride = (long long)gridDim.x * PW_THREADS;
    for (long long i = (long long)blockIdx.x * PW_THREADS + threadIdx.x; i < g.total4; i += stride) {
        int ci = coef_index(i, g);
        float sc = coef[ci], sh = coef[g.ncoef + ci];
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = act_fwd(v[k] * sc + sh, act, slope);
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

// ---------------------------------------------------------------------------
// BatchNorm (training): finalize + apply in ONE launch (round 5)
// ---------------------------------------------------------------------------
// A G+D pair at bs 128 ran 12 bn_finalize + 10 bn_bwd_finalize launches of ~5 us each whose only job was to turn partial
// sums into per-channel coefficients for the kernel right behind them -- 0.11 ms of kernels plus a dependent-launch gap
// each.  Here the consumer does it itself: workgroup (channel c, statistics group g, slice s of the group's samples)
// reduces the partial rows of its own (g, c) -- a few KB, cache-resident, fp64 like the separate kernel -- and then
// streams its slice of channel c's planes.  The workgroup with g = 0, s = 0 ("owner") reduces EVERY group: it writes
// the coefficient array the backward pass reads and updates the running buffers group after group (real, then fake),
// as the separate finalize did.
struct BnFuse {
    int N, C, q4, groups, Ng, S, P;     // S slices per (channel block, group), P samples per slice
    int CB, run4;                       // channels per workgroup (a power of two <= 16), float4 per sample run = CB * q4
    FastDiv div_q4, div_run;
};

// Channel blocks: with small maps (4 x 4, 8 x 8) one channel's plane is 64 / 256 bytes, and a workgroup that walked ONE
// channel touched a 64-byte piece per sample (13.3 us per launch against 9.7 for the flat apply).  CB neighbouring
// channels of a sample are contiguous in NCHW, so a workgroup that owns CB = 64 / q4 channels streams >= 1 KB runs.
static BnFuse bn_fuse_geom(int N, int C, int inner, int groups) {
    BnFuse f;
    f.N = N; f.C = C; f.q4 = inner / 4; f.groups = groups; f.Ng = N / groups;
    int cb = 1;
    while (cb < 16 && cb * f.q4 < 64 && C % (cb * 2) == 0) cb *= 2;
    f.CB = cb;
    f.run4 = cb * f.q4;
    f.div_q4 = make_fastdiv(f.q4);
    f.div_run = make_fastdiv(f.run4);
    // >= ~1024 workgroups, each with >= 512 float4 where the tensor allows
    const long long per_cg = (long long)f.Ng * f.run4;
    const long long blocks = (long long)(C / cb) * groups;
    int S = 1;
    while (blocks * S < 1024 && S * 2 <= f.Ng && per_cg / (S * 2) >= 512) S *= 2;
    f.S = S;
    f.P = (f.Ng + S - 1) / S;
    return f;
}

// sums of (x, y) pairs over `count` entries strided by `stride` f32x2, for the CB channels of this workgroup: thread t
// takes channel t % CB and entries t / CB, t / CB + 256 / CB, ...; the partial sums meet in LDS by halving.  Result for
// channel cb in red[cb] (valid for all threads after the call).
__device__ __forceinline__ void block_channel_sums(const f32x2* __restrict__ base, long long stride, int count, int CB,
                                                   double (*red)[2]) {
    const int tid = threadIdx.x, cb = tid % CB, rr = tid / CB, lanes = PW_THREADS / CB;
    double s1 = 0.0, s2 = 0.0;
    for (int r = rr; r < count; r += lanes) {
        const f32x2 v = base[(long long)r * stride + cb];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    __syncthreads();                       // (the previous use of `red` is over)
    red[tid][0] = s1;
    red[tid][1] = s2;
    __syncthreads();
    for (int h = lanes >> 1; h >= 1; h >>= 1) {
        if (rr < h) {
            red[tid][0] += red[tid + h * CB][0];
            red[tid][1] += red[tid + h * CB][1];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(PW_THREADS) void bn_apply_fused_kernel(
    const float* __restrict__ x, const f32x2* __restrict__ sums, int rows, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ coef, float* running_mean, float* running_var, long long* nbt,
    float* __restrict__ out, BnFuse f, int inner, float eps, float momentum, int act, float slope) {
    __shared__ double red[PW_THREADS][2];
    __shared__ float mine[16][2];
    const int tid = threadIdx.x;
    const int s = blockIdx.x % f.S, bg = blockIdx.x / f.S, g = bg % f.groups, c0 = (bg / f.groups) * f.CB;
    const bool owner = g == 0 && s == 0;
    const int Rg = rows / f.groups, NC = f.groups * f.C;
    for (int gg = 0; gg < f.groups; ++gg) {
        if (!owner && gg != g) continue;                       // (uniform per workgroup)
        block_channel_sums(sums + (long long)gg * Rg * f.C + c0, f.C, Rg, f.CB, red);
        if (tid < f.CB) {
            const int c = c0 + tid;
            const double s1 = red[tid][0], s2 = red[tid][1];
            const double cnt = count > 0.0 ? count : (double)f.Ng * inner;
            const double mean = s1 / cnt;
            double var = s2 / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)eps));
            const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
            const float scale = ga * rstd, shift = be - (float)mean * scale;
            if (gg == g) {
                mine[tid][0] = scale;
                mine[tid][1] = shift;
            }
            if (owner) {
                const int ci = gg * f.C + c;
                coef[ci] = scale;
                coef[NC + ci] = shift;
                coef[2 * NC + ci] = (float)mean;
                coef[3 * NC + ci] = rstd;
                if (running_mean) {
                    const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
                    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
                    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
                }
            }
        }
    }
    if (owner && c0 == 0 && tid == 0 && nbt) *nbt += f.groups;
    __syncthreads();
    const int n0 = g * f.Ng + s * f.P;
    const int n1 = min(g * f.Ng + f.Ng, n0 + f.P);
    const int total = (n1 - n0) * f.run4;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x);
    f32x4* __restrict__ o4 = reinterpret_cast<f32x4*>(out);
    for (int i = tid; i < total; i += PW_THREADS) {
        const uint32_t p = fdiv((uint32_t)i, f.div_run);
        const uint32_t rem = (uint32_t)i - p * (uint32_t)f.run4;
        const uint32_t cb = fdiv(rem, f.div_q4);
        const long long idx = ((long long)(n0 + (int)p) * f.C + c0) * f.q4 + rem;
        const float sc = mine[cb][0], sh = mine[cb][1];
        const f32x4 v = x4[idx];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = act_fwd(v[k] * sc + sh, act, slope);
        o4[idx] = o;
    }
}

// Per-row statistics, their finalize and the apply in ONE launch (InstanceNorm / AdaIN; round 2): the sub-wave that
// summed a row turns the sums into (scale, shift, mean, rstd) itself (gamma / beta per channel, per row, or per row
// from a packed [N][2C] array: affine_per_row 0 / 1 / 2; unbiased = 1 uses var * n/(n-1)) and streams the row again while it is still in this CU's cache.  Replaces row_sums + row_finalize +
// norm_act_apply: HoloGAN ran 25 such triples per optimizer-step pair, each launch costing its ~2.5 us dispatch gap
// on top of ~5 us of work.
// CACHE > 0: a lane keeps its (at most CACHE) float4 of the row in registers, so the row is read from memory once;
// CACHE = 0 streams it three times (rows longer than 64 lanes x 16 float4 = 4096 floats).
template <int CACHE>
__global__ __launch_bounds__(PW_THREADS) void rownorm_act_fused_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta,
                                                                       float* __restrict__ coef,
                                                                       float* __restrict__ out, RowGeom g, int C,
                                                                       int inner, float eps, int affine_per_row,
                                                                       int unbiased, int act, float slope,
                                                                       int x_rows, const float* __restrict__ sigma,
                                                                       int sigma_rows) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * PW_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * PW_THREADS) >> 6;
    const int sub = lane / g.lpr, l = lane % g.lpr;
    for (long long row0 = (long long)wave * g.rpw; row0 < g.rows; row0 += (long long)nwaves * g.rpw) {
        const long long row = row0 + sub;
        const bool live = row < g.rows;
        // x_rows > 0: x has only x_rows rows, shared by every sample (HoloGAN's learned constant, which the reference
        // materialises with .repeat(batch, ...), hologan_generator.py:141)
        const f32x4* p = reinterpret_cast<const f32x4*>(x) + (x_rows > 0 ? row % x_rows : row) * g.q4;
        // Two passes over the (cache-resident) row: the mean first, then the centred second moment with the
        // correction term of the "corrected two-pass" formula.  The one-pass E[x^2] - mean^2 of row_sums_kernel loses
        // mean^2 / var digits, and AdaIN rows do have |mean| >> sigma (a convolution of an all-positive, nearly
        // constant AdaIN+ReLU output): two summation orders of the SAME convolution then moved HoloGAN's
        // second-pair gradients by 3 %, which the reference's torch.var (two passes) does not do.
        f32x4 c[CACHE > 0 ? CACHE : 1];
        float s[1] = {0.f};
        if constexpr (CACHE > 0) {
#pragma unroll
            for (int i = 0; i < CACHE; ++i) {
                const int q = l + i * g.lpr;
                c[i] = (live && q < g.q4) ? p[q] : f32x4{0.f, 0.f, 0.f, 0.f};
                s[0] += (c[i].x + c[i].y) + (c[i].z + c[i].w);
            }
        } else if (live) {
            for (int q = l; q < g.q4; q += g.lpr) {
                f32x4 v = p[q];
                s[0] += (v.x + v.y) + (v.z + v.w);
            }
        }
        sub_reduce<1>(s, g.lpr);        // xor butterfly: every lane of the sub-wave holds the row's sum
        const double cnt = (double)inner;
        const float mean_f = (float)((double)s[0] / cnt);
        float d[2] = {0.f, 0.f};
        if constexpr (CACHE > 0) {
#pragma unroll
            for (int i = 0; i < CACHE; ++i) {
                if (live && l + i * g.lpr < g.q4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float e = c[i][k] - mean_f;
                        d[0] += e;
                        d[1] += e * e;
                    }
                }
            }
        } else if (live) {
            for (int q = l; q < g.q4; q += g.lpr) {
                f32x4 v = p[q];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float e = v[k] - mean_f;
                    d[0] += e;
                    d[1] += e * e;
                }
            }
        }
        sub_reduce<2>(d, g.lpr);
        if (!live) continue;
        const int r = (int)row;
        const double mean = (double)mean_f + (double)d[0] / cnt;
        double var = ((double)d[1] - (double)d[0] * (double)d[0] / cnt) / cnt;
        if (var < 0.0) var = 0.0;
        if (unbiased && inner > 1) var = var * cnt / (cnt - 1.0);
        // sigma (round 4): a spectral-normalised convolution's 1 / sigma folded into the InstanceNorm behind it
        // (functional.sn_conv_in_act): IN_eps(y / sigma) = IN_{eps sigma^2}(y); sigma[g] belongs to rows [g, g + 1) *
        // sigma_rows (the discriminator calls stacked along the batch)
        float eps_r = eps;
        if (sigma) {
            const float sg = sigma[r / sigma_rows];
            eps_r = eps * sg * sg;
        }
        const float rstd = (float)(1.0 / sqrt(var + (double)eps_r));
        const int gi = affine_per_row == 2 ? (r / C) * 2 * C + r % C : (affine_per_row ? r : r % C);
        const float ga = gamma ? gamma[gi] : 1.f, be = beta ? beta[gi] : 0.f;
        const float sc = ga * rstd;
        const float sh = be - (float)mean * sc;
        if (l == 0) {
            coef[r] = sc;
            coef[g.rows + r] = sh;
            coef[2 * g.rows + r] = (float)mean;
            coef[3 * g.rows + r] = rstd;
        }
        if (!out) continue;       // statistics only (gz_rownorm_stats)
        f32x4* po = reinterpret_cast<f32x4*>(out) + row * g.q4;
        if constexpr (CACHE > 0) {
#pragma unroll
            for (int i = 0; i < CACHE; ++i) {
                const int q = l + i * g.lpr;
                if (q < g.q4) {
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = act_fwd(c[i][k] * sc + sh, act, slope);
                    po[q] = o;
                }
            }
        } else {
            for (int q = l; q < g.q4; q += g.lpr) {
                f32x4 v = p[q], o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = act_fwd(v[k] * sc + sh, act, slope);
                po[q] = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// first backward
//   z = x*scale + shift, dz = gout * act'(z), xh = (x - mean) * rstd
//   rowsums: (sum dz, sum dz*xh)
//   dx = scale * (dz - k1 - xh*k2),  k1 = mean(dz), k2 = mean(dz*xh) over the statistics group
// ---------------------------------------------------------------------------
__device__ __forceinline__ float act_grad_z(float z, int act, float slope) {
    if (act == ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == ACT_LRELU) return z > 0.f ? 1.f : slope;
    return 1.f;
}

__global__ __launch_bounds__(PW_THREADS) void norm_bwd_rowsums_kernel(const float* __restrict__ gout,
                                                                      const float* __restrict__ x,
                                                                      const float* __restrict__ coef,
                                                                      f32x2* __restrict__ sums, RowGeom g, int C,
                                                                      int per_channel, int ncoef, int act,
                                                                      float slope, int group_rows) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * PW_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * PW_THREADS) >> 6;
    const int sub = lane / g.lpr, l = lane % g.lpr;
    for (long long row0 = (long long)wave * g.rpw; row0 < g.rows; row0 += (long long)nwaves * g.rpw) {
        long long row = row0 + sub;
        float s[2] = {0.f, 0.f};
        if (row < g.rows) {
            int ci = per_channel ? (int)(row % C) + (group_rows ? (int)(row / group_rows) * C : 0) : (int)row;
            float sc = coef[ci], sh = coef[ncoef + ci], mean = coef[2 * ncoef + ci], rstd = coef[3 * ncoef + ci];
            const f32x4* px = reinterpret_cast<const f32x4*>(x) + row * g.q4;
            const f32x4* pg = reinterpret_cast<const f32x4*>(gout) + row * g.q4;
            for (int q = l; q < g.q4; q += g.lpr) {
                f32x4 xv = px[q], gv = pg[q];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float dz = gv[k] * act_grad_z(xv[k] * sc + sh, act, slope);
                    s[0] += dz;
                    s[1] += dz * ((xv[k] - mean) * rstd);
                }
            }
        }
        sub_reduce<2>(s, g.lpr);
        if (row < g.rows && l == 0) sums[row] = f32x2{s[0], s[1]};
    }
}

// BatchNorm: combine row sums over n -> k[0*C] = mean(dz), k[1*C] = mean(dz*xh); dgamma, dbeta.
// One wavefront per channel.
// (statistics groups: k per (group, channel); the affine gradients sum over the groups.  accumulate != 0: dgamma /
// dbeta already hold earlier contributions -- the gradient sink of functional/_base.py -- and are added to)
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const f32x2* __restrict__ sums, float* __restrict__ k,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int N, int C, int inner, int groups, int accumulate) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int Ng = N / groups, NC = groups * C;
    double t1 = 0.0, t2 = 0.0;
    for (int g = 0; g < groups; ++g) {
        double s1 = 0.0, s2 = 0.0;
        for (int n = lane; n < Ng; n += 64) {
            f32x2 v = sums[(long long)(g * Ng + n) * C + c];
            s1 += (double)v.x;
            s2 += (double)v.y;
        }
        s1 = wave_sum_d(s1);
        s2 = wave_sum_d(s2);
        if (lane == 0) {
            double cnt = (double)Ng * inner;
            k[g * C + c] = (float)(s1 / cnt);
            k[NC + g * C + c] = (float)(s2 / cnt);
        }
        t1 += s1;
        t2 += s2;
    }
    if (lane != 0) return;
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)t2;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)t1;
}

// per-row statistics: k = sums / inner (and per-row dgamma / dbeta for AdaIN-style affine)
// With the unbiased variance (AdaIN) d xh_i/d x_j = rstd (delta_ij - 1/n - xh_i xh_j / (n-1)), hence the
// second coefficient is sum(dz*xh) / (n-1).
__global__ void row_bwd_finalize_kernel(const f32x2* __restrict__ sums, float* __restrict__ k,
                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C,
                                        int inner, int affine_per_row, int unbiased) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int rows = N * C;
    if (i < rows) {
        f32x2 v = sums[i];
        k[i] = v.x / (float)inner;
        k[rows + i] = v.y / (float)((unbiased && inner > 1) ? inner - 1 : inner);
        if (affine_per_row) {
            const int gi = affine_per_row == 2 ? (i / C) * 2 * C + i % C : i;     // 2: packed [N][2C] gradient
            if (dgamma) dgamma[gi] = v.y;
            if (dbeta) dbeta[gi] = v.x;
        }
    }
}

// per-channel affine over per-row statistics (InstanceNorm): dgamma[c] = sum_n s2[n,c], dbeta[c] = sum_n s1[n,c]
__global__ __launch_bounds__(64) void row_bwd_affine_kernel(const f32x2* __restrict__ sums,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int N, int C, int accumulate = 0) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int n = lane; n < N; n += 64) {
        f32x2 v = sums[(long long)n * C + c];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    s1 = wave_sum_d(s1);
    s2 = wave_sum_d(s2);
    if (lane != 0) return;
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
}

__global__ __launch_bounds__(PW_THREADS) void norm_bwd_apply_kernel(const float* __restrict__ gout,
                                                                    const float* __restrict__ x,
                                                                    const float* __restrict__ coef,
                                                                    const float* __restrict__ k,
                                                                    float* __restrict__ dx, ApplyGeom g, int act,
                                                                    float slope) {
    const long long stride = (long long)gridDim.x * PW_THREADS;
    for (long long i = (long long)blockIdx.x * PW_THREADS + threadIdx.x; i < g.total4; i += stride) {
        int ci = coef_index(i, g);
        float sc = coef[ci], sh = coef[g.ncoef + ci], mean = coef[2 * g.ncoef + ci], rstd = coef[3 * g.ncoef + ci];
        float k1 = k[ci], k2 = k[g.ncoef + ci];
        f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 gv = reinterpret_cast<const f32x4*>(gout)[i];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dz = gv[q] * act_grad_z(xv[q] * sc + sh, act, slope);
            float xh = (xv[q] - mean) * rstd;
            o[q] = sc * (dz - k1 - xh * k2);
        }
        reinterpret_cast<f32x4*>(dx)[i] = o;
    }
}

// BatchNorm backward, finalize + apply in ONE launch (see bn_apply_fused_kernel): workgroup (channel block, g, s) sums
// the row sums (sum dz, sum dz xh) of its channels over the group's samples, forms k1, k2 and streams dx for its slice;
// the owner (g = 0, s = 0) also writes / adds dgamma, dbeta summed over all groups (and k, for callers that read it).
__global__ __launch_bounds__(PW_THREADS) void bn_bwd_apply_fused_kernel(
    const float* __restrict__ gout, const float* __restrict__ x, const float* __restrict__ coef,
    const f32x2* __restrict__ sums, float* __restrict__ kout, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dx, BnFuse f, int inner, int act, float slope, int accumulate) {
    __shared__ double red[PW_THREADS][2];
    __shared__ float mine[16][6];          // k1, k2, scale, shift, mean, rstd
    const int tid = threadIdx.x;
    const int s = blockIdx.x % f.S, bg = blockIdx.x / f.S, g = bg % f.groups, c0 = (bg / f.groups) * f.CB;
    const bool owner = g == 0 && s == 0;
    const int NC = f.groups * f.C;
    double t1 = 0.0, t2 = 0.0;
    for (int gg = 0; gg < f.groups; ++gg) {
        if (!owner && gg != g) continue;
        block_channel_sums(sums + (long long)gg * f.Ng * f.C + c0, f.C, f.Ng, f.CB, red);
        if (tid < f.CB) {
            const double s1 = red[tid][0], s2 = red[tid][1];
            const double cnt = (double)f.Ng * inner;
            const float k1 = (float)(s1 / cnt), k2 = (float)(s2 / cnt);
            if (gg == g) {
                mine[tid][0] = k1;
                mine[tid][1] = k2;
            }
            if (owner && kout) {
                kout[gg * f.C + c0 + tid] = k1;
                kout[NC + gg * f.C + c0 + tid] = k2;
            }
            t1 += s1;
            t2 += s2;
        }
    }
    if (tid < f.CB) {
        const int c = c0 + tid, ci = g * f.C + c;
        if (owner) {
            if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)t2;
            if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)t1;
        }
        mine[tid][2] = coef[ci];
        mine[tid][3] = coef[NC + ci];
        mine[tid][4] = coef[2 * NC + ci];
        mine[tid][5] = coef[3 * NC + ci];
    }
    __syncthreads();
    const int n0 = g * f.Ng + s * f.P;
    const int n1 = min(g * f.Ng + f.Ng, n0 + f.P);
    const int total = (n1 - n0) * f.run4;
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x);
    const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(gout);
    f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dx);
    for (int i = tid; i < total; i += PW_THREADS) {
        const uint32_t p = fdiv((uint32_t)i, f.div_run);
        const uint32_t rem = (uint32_t)i - p * (uint32_t)f.run4;
        const uint32_t cb = fdiv(rem, f.div_q4);
        const long long idx = ((long long)(n0 + (int)p) * f.C + c0) * f.q4 + rem;
        const float k1 = mine[cb][0], k2 = mine[cb][1], sc = mine[cb][2], sh = mine[cb][3], mean = mine[cb][4],
                    rstd = mine[cb][5];
        const f32x4 xv = x4[idx], gv = g4[idx];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float dz = gv[q] * act_grad_z(xv[q] * sc + sh, act, slope);
            const float xh = (xv[q] - mean) * rstd;
            o[q] = sc * (dz - k1 - xh * k2);
        }
        d4[idx] = o;
    }
}

// First backward of a per-row normalisation in ONE launch: row sums of (dz, dz*xh), the two coefficients, the per-row
// affine gradients (AdaIN) and dx = scale * (dz - k1 - xh*k2) from a second, cache-resident read of the row.
// `sums` (optional) keeps the raw row sums for row_bwd_affine_kernel (per-channel gamma / beta).
template <int CACHE>      // as rownorm_act_fused_kernel: dz and xh of the lane's float4 stay in registers
__global__ __launch_bounds__(PW_THREADS) void rownorm_bwd_fused_kernel(const float* __restrict__ gout,
                                                                       const float* __restrict__ x,
                                                                       const float* __restrict__ coef,
                                                                       float* __restrict__ dx,
                                                                       float* __restrict__ dgamma,
                                                                       float* __restrict__ dbeta,
                                                                       f32x2* __restrict__ sums, RowGeom g, int C,
                                                                       int inner, int affine_per_row, int unbiased,
                                                                       int act, float slope) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * PW_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * PW_THREADS) >> 6;
    const int sub = lane / g.lpr, l = lane % g.lpr;
    for (long long row0 = (long long)wave * g.rpw; row0 < g.rows; row0 += (long long)nwaves * g.rpw) {
        const long long row = row0 + sub;
        const bool live = row < g.rows;
        const int ci = live ? (int)row : 0;
        const float sc = coef[ci], sh = coef[g.rows + ci], mean = coef[2 * g.rows + ci], rstd = coef[3 * g.rows + ci];
        const f32x4* px = reinterpret_cast<const f32x4*>(x) + row * g.q4;
        const f32x4* pg = reinterpret_cast<const f32x4*>(gout) + row * g.q4;
        float s[2] = {0.f, 0.f};
        f32x4 cdz[CACHE > 0 ? CACHE : 1], cxh[CACHE > 0 ? CACHE : 1];
        if constexpr (CACHE > 0) {
#pragma unroll
            for (int i = 0; i < CACHE; ++i) {
                const int q = l + i * g.lpr;
                const bool in = live && q < g.q4;
                const f32x4 xv = in ? px[q] : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 gv = in ? pg[q] : f32x4{0.f, 0.f, 0.f, 0.f};     // gv = 0: nothing added to the sums
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dz = gv[k] * act_grad_z(xv[k] * sc + sh, act, slope);
                    const float xh = (xv[k] - mean) * rstd;
                    cdz[i][k] = dz;
                    cxh[i][k] = xh;
                    s[0] += dz;
                    s[1] += dz * xh;
                }
            }
        } else if (live) {
            for (int q = l; q < g.q4; q += g.lpr) {
                f32x4 xv = px[q], gv = pg[q];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float dz = gv[k] * act_grad_z(xv[k] * sc + sh, act, slope);
                    s[0] += dz;
                    s[1] += dz * ((xv[k] - mean) * rstd);
                }
            }
        }
        sub_reduce<2>(s, g.lpr);
        if (!live) continue;
        const float k1 = s[0] / (float)inner;
        const float k2 = s[1] / (float)((unbiased && inner > 1) ? inner - 1 : inner);
        if (l == 0) {
            if (sums) sums[row] = f32x2{s[0], s[1]};
            if (affine_per_row) {
                const int r = (int)row;
                const int gi = affine_per_row == 2 ? (r / C) * 2 * C + r % C : r;
                if (dgamma) dgamma[gi] = s[1];
                if (dbeta) dbeta[gi] = s[0];
            }
        }
        if (!dx) continue;
        f32x4* po = reinterpret_cast<f32x4*>(dx) + row * g.q4;
        if constexpr (CACHE > 0) {
#pragma unroll
            for (int i = 0; i < CACHE; ++i) {
                const int q = l + i * g.lpr;
                if (q < g.q4) {
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = sc * (cdz[i][k] - k1 - cxh[i][k] * k2);
                    po[q] = o;
                }
            }
        } else {
            for (int q = l; q < g.q4; q += g.lpr) {
                f32x4 xv = px[q], gv = pg[q], o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float dz = gv[k] * act_grad_z(xv[k] * sc + sh, act, slope);
                    float xh = (xv[k] - mean) * rstd;
                    o[k] = sc * (dz - k1 - xh * k2);
                }
                po[q] = o;
            }
        }
    }
}

// First backward of AdaIN over a constant shared by every sample (x: [C][inner], gout: [N][C][inner]): one workgroup
// per channel; its 256 / lpr lane groups take every (256 / lpr)-th sample, write that sample's packed
// (d scale | d shift) and accumulate scale (dz - k1 - xh k2) in registers; the groups' partial d x[c] meet in LDS in a
// fixed order.  (The framework spelling was a [N, C, inner] dx plus a sum; a first version with one lane group per
// channel walking all N samples took 110 us for 512 channels -- 128 wavefronts of serial, dependent loads.)
__global__ __launch_bounds__(PW_THREADS) void adain_const_bwd_kernel(const float* __restrict__ gout,
                                                                     const float* __restrict__ x,
                                                                     const float* __restrict__ coef,
                                                                     float* __restrict__ dx, float* __restrict__ dsb,
                                                                     RowGeom g, int N, int C, int inner, int act,
                                                                     float slope) {
    constexpr int CACHE = 4;
    __shared__ float part[4096];                 // groups x inner <= (256 / lpr) x (lpr x 16) floats
    const int c = blockIdx.x;
    const int grp = threadIdx.x / g.lpr, l = threadIdx.x % g.lpr, groups = PW_THREADS / g.lpr;
    const int rows = N * C;
    const f32x4* px = reinterpret_cast<const f32x4*>(x) + (long long)c * g.q4;
    f32x4 xv[CACHE], acc[CACHE];
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
        const int q = l + i * g.lpr;
        xv[i] = q < g.q4 ? px[q] : f32x4{0.f, 0.f, 0.f, 0.f};
        acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int n = grp; n < N; n += groups) {
        const int r = n * C + c;
        const float sc = coef[r], sh = coef[rows + r], mean = coef[2 * rows + r], rstd = coef[3 * rows + r];
        const f32x4* pg = reinterpret_cast<const f32x4*>(gout) + (long long)r * g.q4;
        f32x4 dz[CACHE], xh[CACHE];
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CACHE; ++i) {
            const int q = l + i * g.lpr;
            const f32x4 gv = q < g.q4 ? pg[q] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dz[i][k] = gv[k] * act_grad_z(xv[i][k] * sc + sh, act, slope);
                xh[i][k] = (xv[i][k] - mean) * rstd;
                s[0] += dz[i][k];
                s[1] += dz[i][k] * xh[i][k];
            }
        }
        sub_reduce<2>(s, g.lpr);
        const float k1 = s[0] / (float)inner, k2 = s[1] / (float)(inner > 1 ? inner - 1 : inner);
        if (l == 0) {
            dsb[(long long)n * 2 * C + c] = s[1];
            dsb[(long long)n * 2 * C + C + c] = s[0];
        }
#pragma unroll
        for (int i = 0; i < CACHE; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[i][k] += sc * (dz[i][k] - k1 - xh[i][k] * k2);
    }
#pragma unroll
    for (int i = 0; i < CACHE; ++i) {
        const int q = l + i * g.lpr;
        if (q < g.q4) reinterpret_cast<f32x4*>(part + grp * inner)[q] = acc[i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < inner; e += PW_THREADS) {
        float t = 0.f;
        for (int k = 0; k < groups; ++k) t += part[k * inner + e];
        dx[(long long)c * inner + e] = t;
    }
}

// ---------------------------------------------------------------------------
// double backward (per-row statistics only: InstanceNorm).  With g = gout*act'(z) and
// v = dL/d(dx):   sums5[row] = (sum v, sum g, sum g*xh, sum v*xh, sum v*g)
//   gg_out = act'(z) * scale * (v - mean(v) - xh*mean(v*xh))
//   gx     = -scale*rstd * [ xh*(mean(vg) - mean(v)mean(g) - c*mvx) + c*(v - mean(v) - xh*mvx)
//                            + mvx*(g - mean(g) - xh*c) ],   c = mean(g*xh), mvx = mean(v*xh)
//   ggamma[c] += rstd * inner * (mean(vg) - mean(v)mean(g) - mvx*c)        (summed over n)
// ---------------------------------------------------------------------------
struct Sums5 {
    float v, g, gx, vx, vg;
};

__global__ __launch_bounds__(PW_THREADS) void norm_bwd2_rowsums_kernel(const float* __restrict__ gout,
                                                                       const float* __restrict__ v,
                                                                       const float* __restrict__ x,
                                                                       const float* __restrict__ coef,
                                                                       Sums5* __restrict__ sums, RowGeom g,
                                                                       int act, float slope) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * PW_THREADS + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * PW_THREADS) >> 6;
    const int sub = lane / g.lpr, l = lane % g.lpr;
    const int ncoef = g.rows;
    for (long long row0 = (long long)wave * g.rpw; row0 < g.rows; row0 += (long long)nwaves * g.rpw) {
        long long row = row0 + sub;
        float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (row < g.rows) {
            float sc = coef[row], sh = coef[ncoef + row], mean = coef[2 * ncoef + row], rstd = coef[3 * ncoef + row];
            const f32x4* px = reinterpret_cast<const f32x4*>(x) + row * g.q4;
            const f32x4* pg = reinterpret_cast<const f32x4*>(gout) + row * g.q4;
            const f32x4* pv = reinterpret_cast<const f32x4*>(v) + row * g.q4;
            for (int q = l; q < g.q4; q += g.lpr) {
                f32x4 xv = px[q], gv = pg[q], vv = pv[q];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float gz = gv[k] * act_grad_z(xv[k] * sc + sh, act, slope);
                    float xh = (xv[k] - mean) * rstd;
                    s[0] += vv[k];
                    s[1] += gz;
                    s[2] += gz * xh;
                    s[3] += vv[k] * xh;
                    s[4] += vv[k] * gz;
                }
            }
        }
        sub_reduce<5>(s, g.lpr);
        if (row < g.rows && l == 0) sums[row] = Sums5{s[0], s[1], s[2], s[3], s[4]};
    }
}

__global__ __launch_bounds__(PW_THREADS) void norm_bwd2_apply_kernel(
    const float* __restrict__ gout, const float* __restrict__ v, const float* __restrict__ x,
    const float* __restrict__ coef, const Sums5* __restrict__ sums, float* __restrict__ gg_out,
    float* __restrict__ gx, ApplyGeom g, int act, float slope) {
    const long long stride = (long long)gridDim.x * PW_THREADS;
    const float inv = 1.f / (float)(g.q4 * 4);
    for (long long i = (long long)blockIdx.x * PW_THREADS + threadIdx.x; i < g.total4; i += stride) {
        int r = (int)fdiv((uint32_t)i, g.div_q4);
        float sc = coef[r], sh = coef[g.ncoef + r], mean = coef[2 * g.ncoef + r], rstd = coef[3 * g.ncoef + r];
        Sums5 s = sums[r];
        float mv = s.v * inv, mg = s.g * inv, c = s.gx * inv, mvx = s.vx * inv, mvg = s.vg * inv;
        float t1 = mvg - mv * mg - c * mvx;
        f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 gv = reinterpret_cast<const f32x4*>(gout)[i];
        f32x4 vv = reinterpret_cast<const f32x4*>(v)[i];
        f32x4 o1, o2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float m = act_grad_z(xv[q] * sc + sh, act, slope);
            float gz = gv[q] * m;
            float xh = (xv[q] - mean) * rstd;
            float pv = vv[q] - mv - xh * mvx;
            o1[q] = m * sc * pv;
            o2[q] = -sc * rstd * (xh * t1 + c * pv + mvx * (gz - mg - xh * c));
        }
        if (gg_out) reinterpret_cast<f32x4*>(gg_out)[i] = o1;
        if (gx) reinterpret_cast<f32x4*>(gx)[i] = o2;
    }
}

// ggamma[c] = sum_n rstd[n,c] * inner * (mvg - mv*mg - mvx*c); one wavefront per channel
__global__ __launch_bounds__(64) void norm_bwd2_ggamma_kernel(const Sums5* __restrict__ sums,
                                                              const float* __restrict__ coef,
                                                              float* __restrict__ ggamma, int N, int C, int inner) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int rows = N * C;
    double acc = 0.0;
    const float inv = 1.f / (float)inner;
    for (int n = lane; n < N; n += 64) {
        int r = n * C + c;
        Sums5 s = sums[r];
        float rstd = coef[3 * rows + r];
        acc += (double)(rstd * (s.vg - s.v * s.g * inv - s.vx * s.gx * inv));
    }
    acc = wave_sum_d(acc);
    if (lane == 0) ggamma[c] = (float)acc;
}

// ---------------------------------------------------------------------------
// plain activation backward: dx = g * act'(out) expressed through the output
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(PW_THREADS) void act_bwd_kernel(const float* __restrict__ g,
                                                             const float* __restrict__ out,
                                                             float* __restrict__ dx, long long total4, int act,
                                                             float slope) {
    const long long stride = (long long)gridDim.x * PW_THREADS;
    for (long long i = (long long)blockIdx.x * PW_THREADS + threadIdx.x; i < total4; i += stride) {
        f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 ov = reinterpret_cast<const f32x4*>(out)[i];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = gv[q] * act_bwd_from_out(ov[q], act, slope);
        reinterpret_cast<f32x4*>(dx)[i] = o;
    }
}

// d/d(out) of g * (1 - out^2) contracted with v: the tanh leg of a second-order backward
__global__ __launch_bounds__(PW_THREADS) void tanh_bwd2_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                               const float* __restrict__ out, float* __restrict__ res,
                                                               long long total4) {
    const long long stride = (long long)gridDim.x * PW_THREADS;
    for (long long i = (long long)blockIdx.x * PW_THREADS + threadIdx.x; i < total4; i += stride) {
        const f32x4 a = reinterpret_cast<const f32x4*>(v)[i], b = reinterpret_cast<const f32x4*>(g)[i];
        const f32x4 o = reinterpret_cast<const f32x4*>(out)[i];
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = a[q] * b[q] * (-2.f * o[q]);
        reinterpret_cast<f32x4*>(res)[i] = r;
    }
}

static int ew_grid(long long total4) {
    long long b = (total4 + PW_THREADS - 1) / PW_THREADS;
    if (b > 256 * 8) b = 256 * 8;
    if (b < 1) b = 1;
    return (int)b;
}

static ApplyGeom apply_geom(int N, int C, int inner, int per_channel, int groups = 1) {
    ApplyGeom g;
    g.total4 = (long long)N * C * inner / 4;
    g.q4 = inner / 4;
    g.div_q4 = make_fastdiv(g.q4);
    g.div_c = make_fastdiv(C);
    g.C = C;
    g.per_channel = per_channel;
    g.ncoef = per_channel ? groups * C : N * C;
    g.group_rows = (per_channel && groups > 1) ? (N / groups) * C : 0;
    g.div_grows = make_fastdiv(g.group_rows ? g.group_rows : 1);
    return g;
}

static bool groups_ok(int N, int groups) { return groups >= 1 && groups <= 8 && N % groups == 0; }

static bool norm_shape_ok(int N, int C, int inner) {
    return N > 0 && C > 0 && inner > 0 && (inner % 4) == 0 && (long long)N * C * inner * 4 < (1ll << 33) &&
           (long long)N * C * inner / 4 < (1ll << 31);
}

static void launch_rownorm_fused(const float* x, const float* gamma, const float* beta, float* coef, float* out,
                                 const RowGeom& g, int C, int inner, float eps, int affine_per_row, int unbiased,
                                 int act, float slope, hipStream_t stream, int x_rows = 0,
                                 const float* sigma = nullptr, int sigma_rows = 1) {
    const int per_lane = (g.q4 + g.lpr - 1) / g.lpr;      // float4 per lane
#define GZ_RN(CACHE)                                                                                               \
    hipLaunchKernelGGL(rownorm_act_fused_kernel<CACHE>, dim3(row_grid(g)), dim3(PW_THREADS), 0, stream, x, gamma, \
                       beta, coef, out, g, C, inner, eps, affine_per_row, unbiased, act, slope, x_rows, sigma, sigma_rows)
    if (per_lane <= 1) GZ_RN(1);
    else if (per_lane <= 4) GZ_RN(4);
    else if (per_lane <= 16) GZ_RN(16);
    else GZ_RN(0);
#undef GZ_RN
}

}  // namespace gz

using namespace gz;

extern "C" {

size_t gz_norm_workspace_bytes(int N, int C) { return (size_t)N * C * sizeof(Sums5); }

int gz_norm_coef_elems(int N, int C, int per_channel) { return 4 * (per_channel ? C : N * C); }

int gz_batchnorm_stats_g(const float* x, const float* gamma, const float* beta, float* coef, float* running_mean,
                         float* running_var, long long* num_batches_tracked, void* workspace, int N, int C,
                         int inner, float eps, float momentum, int groups, hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner) || !groups_ok(N, groups)) return GZ_ERR_BAD_SHAPE;
    RowGeom g = row_geom((long long)N * C, inner);
    hipLaunchKernelGGL(row_sums_kernel, dim3(row_grid(g)), dim3(PW_THREADS), 0, stream, x, (f32x2*)workspace, g);
    hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3(C), dim3(64), 0, stream, (const f32x2*)workspace, gamma, beta, coef,
                       running_mean, running_var, num_batches_tracked, N, C, inner, eps, momentum, 0.0, groups);
    return launch_status();
}

int gz_batchnorm_stats(const float* x, const float* gamma, const float* beta, float* coef, float* running_mean,
                       float* running_var, long long* num_batches_tracked, void* workspace, int N, int C,
                       int inner, float eps, float momentum, hipStream_t stream) {
    return gz_batchnorm_stats_g(x, gamma, beta, coef, running_mean, running_var, num_batches_tracked, workspace, N, C,
                                inner, eps, momentum, 1, stream);
}

int gz_batchnorm_finalize_g(const float* partials, int rows, long long count, const float* gamma, const float* beta,
                            float* coef, float* running_mean, float* running_var, long long* num_batches_tracked,
                            int C, float eps, float momentum, int groups, hipStream_t stream) {
    gz::clear_stale_error();
    if (rows <= 0 || C <= 0 || count <= 0 || !groups_ok(rows, groups)) return GZ_ERR_BAD_SHAPE;
    if (rows / groups > 512)
        hipLaunchKernelGGL(bn_finalize_kernel<8>, dim3(C), dim3(512), 0, stream, (const f32x2*)partials, gamma, beta, coef,
                           running_mean, running_var, num_batches_tracked, rows, C, 0, eps, momentum, (double)count,
                           groups);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3(C), dim3(64), 0, stream, (const f32x2*)partials, gamma, beta, coef,
                           running_mean, running_var, num_batches_tracked, rows, C, 0, eps, momentum, (double)count,
                           groups);
    return launch_status();
}

/* Training-mode BatchNorm + activation forward in as few launches as the statistics allow: with the producing
 * convolution's partial rows (`partials`, rows, count per group) ONE launch (bn_apply_fused_kernel: finalize + apply);
 * without (partials == NULL) the row sums of x first (workspace: gz_norm_workspace_bytes).  Same coefficient array,
 * running-buffer update and group semantics as gz_batchnorm_finalize_g / gz_batchnorm_stats_g + gz_norm_act_fwd_g. */
int gz_batchnorm_act_fwd_fused(const float* x, const float* partials, int rows, long long count, const float* gamma,
                               const float* beta, float* coef, float* running_mean, float* running_var,
                               long long* num_batches_tracked, void* workspace, float* out, int N, int C, int inner,
                               float eps, float momentum, int groups, int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner) || !groups_ok(N, groups)) return GZ_ERR_BAD_SHAPE;
    const f32x2* sums = (const f32x2*)partials;
    double cnt = (double)count;
    if (!partials) {
        if (!workspace) return GZ_ERR_WORKSPACE;
        RowGeom g = row_geom((long long)N * C, inner);
        hipLaunchKernelGGL(row_sums_kernel, dim3(row_grid(g)), dim3(PW_THREADS), 0, stream, x, (f32x2*)workspace, g);
        sums = (const f32x2*)workspace;
        rows = N;
        cnt = 0.0;
    } else if (rows <= 0 || count <= 0 || rows % groups) {
        return GZ_ERR_BAD_SHAPE;
    }
    const BnFuse f = bn_fuse_geom(N, C, inner, groups);
    hipLaunchKernelGGL(bn_apply_fused_kernel, dim3((C / f.CB) * groups * f.S), dim3(PW_THREADS), 0, stream, x, sums, rows, cnt,
                       gamma, beta, coef, running_mean, running_var, num_batches_tracked, out, f, inner, eps, momentum,
                       act, slope);
    return launch_status();
}

int gz_batchnorm_finalize(const float* partials, int rows, long long count, const float* gamma, const float* beta,
                          float* coef, float* running_mean, float* running_var, long long* num_batches_tracked, int C,
                          float eps, float momentum, hipStream_t stream) {
    return gz_batchnorm_finalize_g(partials, rows, count, gamma, beta, coef, running_mean, running_var,
                                   num_batches_tracked, C, eps, momentum, 1, stream);
}

int gz_channel_sum(const float* x, float* out, void* workspace, int N, int C, int inner, hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner)) return GZ_ERR_BAD_SHAPE;
    RowGeom g = row_geom((long long)N * C, inner);
    hipLaunchKernelGGL(row_sums_kernel, dim3(row_grid(g)), dim3(PW_THREADS), 0, stream, x, (f32x2*)workspace, g);
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(C), dim3(64), 0, stream, (const f32x2*)workspace, out, N, C);
    return launch_status();
}

int gz_batchnorm_eval_coef(const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float* coef, int C, float eps, hipStream_t stream) {
    gz::clear_stale_error();
    if (C <= 0) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((C + 127) / 128), dim3(128), 0, stream, gamma, beta, running_mean,
                       running_var, coef, C, eps);
    return launch_status();
}

int gz_rownorm_stats(const float* x, const float* gamma, const float* beta, float* coef, void* workspace, int N,
                     int C, int inner, float eps, int affine_per_row, int unbiased, hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner)) return GZ_ERR_BAD_SHAPE;
    (void)workspace;
    RowGeom g = row_geom((long long)N * C, inner);
    launch_rownorm_fused(x, gamma, beta, coef, nullptr, g, C, inner, eps, affine_per_row, unbiased, ACT_NONE, 0.f, stream);
    return launch_status();
}

int gz_rownorm_act_fwd(const float* x, const float* gamma, const float* beta, float* coef, float* out, int N, int C,
                       int inner, float eps, int affine_per_row, int unbiased, int act, float slope,
                       hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner)) return GZ_ERR_BAD_SHAPE;
    RowGeom g = row_geom((long long)N * C, inner);
    launch_rownorm_fused(x, gamma, beta, coef, out, g, C, inner, eps, affine_per_row, unbiased, act, slope, stream);
    return launch_status();
}

int gz_rownorm_act_fwd_sigma(const float* x, const float* sigma, int groups, float* coef, float* out, int N, int C,
                             int inner, float eps, int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner) || !sigma || groups <= 0 || N % groups) return GZ_ERR_BAD_SHAPE;
    RowGeom g = row_geom((long long)N * C, inner);
    launch_rownorm_fused(x, nullptr, nullptr, coef, out, g, C, inner, eps, 0, 0, act, slope, stream, 0, sigma,
                         (N / groups) * C);
    return launch_status();
}

int gz_rownorm_act_bwd_rows(const float* gout, const float* x, const float* coef, float* dx, float* rowsums, int N, int C,
                            int inner, int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner) || !dx || !rowsums) return GZ_ERR_BAD_SHAPE;
    RowGeom rg = row_geom((long long)N * C, inner);
    const int per_lane = (rg.q4 + rg.lpr - 1) / rg.lpr;
    const int max_cache = knobs().norm_bwd_cache;
#define GZ_RB(CACHE)                                                                                                  \
    hipLaunchKernelGGL(rownorm_bwd_fused_kernel<CACHE>, dim3(row_grid(rg)), dim3(PW_THREADS), 0, stream, gout, x, coef, \
                       dx, (float*)nullptr, (float*)nullptr, (f32x2*)rowsums, rg, C, inner, 0, 0, act, slope)
    if (per_lane <= 1) GZ_RB(1);
    else if (per_lane <= 4 && max_cache >= 4) GZ_RB(4);
    else if (per_lane <= 16 && max_cache >= 16) GZ_RB(16);
    else GZ_RB(0);
#undef GZ_RB
    return launch_status();
}

int gz_adain_const_fwd(const float* x, const float* sb, float* coef, float* out, int N, int C, int inner, float eps,
                       int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner)) return GZ_ERR_BAD_SHAPE;
    RowGeom g = row_geom((long long)N * C, inner);
    launch_rownorm_fused(x, sb, sb + C, coef, out, g, C, inner, eps, 2, 1, act, slope, stream, C);
    return launch_status();
}

int gz_adain_const_bwd(const float* gout, const float* x, const float* coef, float* dx, float* dsb, int N, int C,
                       int inner, int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner)) return GZ_ERR_BAD_SHAPE;
    RowGeom g = row_geom(C, inner);
    if ((g.q4 + g.lpr - 1) / g.lpr > 4) return GZ_ERR_UNSUPPORTED;        // rows of more than 1024 floats
    hipLaunchKernelGGL(adain_const_bwd_kernel, dim3(C), dim3(PW_THREADS), 0, stream, gout, x, coef, dx, dsb, g, N, C,
                       inner, act, slope);
    return launch_status();
}

int gz_norm_act_fwd_g(const float* x, const float* coef, float* out, int N, int C, int inner, int per_channel,
                      int groups, int act, float slope, hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner) || !groups_ok(N, groups) || (groups > 1 && !per_channel)) return GZ_ERR_BAD_SHAPE;
    ApplyGeom g = apply_geom(N, C, inner, per_channel, groups);
    hipLaunchKernelGGL(norm_act_apply_kernel, dim3(ew_grid(g.total4)), dim3(PW_THREADS), 0, stream, x, coef, out, g,
                       act, slope);
    return launch_status();
}

int gz_norm_act_fwd(const float* x, const float* coef, float* out, int N, int C, int inner, int per_channel,
                    int act, float slope, hipStream_t stream) {
    return gz_norm_act_fwd_g(x, coef, out, N, C, inner, per_channel, 1, act, slope, stream);
}

static int norm_act_bwd_impl(const float* gout, const float* x, const float* coef, float* dx, float* dgamma, float* dbeta,
                             void* workspace, float* kbuf, int N, int C, int inner, int per_channel, int affine_per_row,
                             int unbiased, int act, float slope, int groups, int accumulate, hipStream_t stream);

int gz_norm_act_bwd(const float* gout, const float* x, const float* coef, float* dx, float* dgamma, float* dbeta,
                    void* workspace, float* kbuf, int N, int C, int inner, int per_channel, int affine_per_row,
                    int unbiased, int act, float slope, hipStream_t stream) {
    return norm_act_bwd_impl(gout, x, coef, dx, dgamma, dbeta, workspace, kbuf, N, C, inner, per_channel, affine_per_row,
                             unbiased, act, slope, 1, 0, stream);
}

/* InstanceNorm with per-channel affine (per_channel = 0, affine_per_row = 0): accumulate != 0 adds the affine gradients
 * to dgamma / dbeta (gradient sinks; first-order backward only) */
int gz_rownorm_act_bwd_acc(const float* gout, const float* x, const float* coef, float* dx, float* dgamma, float* dbeta,
                           void* workspace, float* kbuf, int N, int C, int inner, int act, float slope, int accumulate,
                           hipStream_t stream) {
    return norm_act_bwd_impl(gout, x, coef, dx, dgamma, dbeta, workspace, kbuf, N, C, inner, 0, 0, 0, act, slope, 1,
                             accumulate, stream);
}

/* BatchNorm (per_channel) with statistics groups; accumulate != 0 adds the affine gradients to dgamma / dbeta */
int gz_batchnorm_act_bwd_g(const float* gout, const float* x, const float* coef, float* dx, float* dgamma, float* dbeta,
                           void* workspace, float* kbuf, int N, int C, int inner, int act, float slope, int groups,
                           int accumulate, hipStream_t stream) {
    if (!groups_ok(N, groups)) return GZ_ERR_BAD_SHAPE;
    return norm_act_bwd_impl(gout, x, coef, dx, dgamma, dbeta, workspace, kbuf, N, C, inner, 1, 0, 0, act, slope, groups,
                             accumulate, stream);
}

static int norm_act_bwd_impl(const float* gout, const float* x, const float* coef, float* dx, float* dgamma, float* dbeta,
                             void* workspace, float* kbuf, int N, int C, int inner, int per_channel, int affine_per_row,
                             int unbiased, int act, float slope, int groups, int accumulate, hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner)) return GZ_ERR_BAD_SHAPE;
    RowGeom rg = row_geom((long long)N * C, inner);
    ApplyGeom g = apply_geom(N, C, inner, per_channel, groups);
    const bool unfused = knobs().norm_unfused;      // experiment: the three-launch path
    if (!per_channel && !unfused) {
        const bool channel_affine = !affine_per_row && (dgamma || dbeta);
        const int max_cache = knobs().norm_bwd_cache;
        const int per_lane = (rg.q4 + rg.lpr - 1) / rg.lpr;
#define GZ_RB(CACHE)                                                                                                  \
    hipLaunchKernelGGL(rownorm_bwd_fused_kernel<CACHE>, dim3(row_grid(rg)), dim3(PW_THREADS), 0, stream, gout, x, coef, \
                       dx, dgamma, dbeta, channel_affine ? (f32x2*)workspace : (f32x2*)nullptr, rg, C, inner,          \
                       affine_per_row, unbiased, act, slope)
        if (per_lane <= 1) GZ_RB(1);
        else if (per_lane <= 4 && max_cache >= 4) GZ_RB(4);
        else if (per_lane <= 16 && max_cache >= 16) GZ_RB(16);
        else GZ_RB(0);
#undef GZ_RB
        if (channel_affine)
            hipLaunchKernelGGL(row_bwd_affine_kernel, dim3(C), dim3(64), 0, stream, (const f32x2*)workspace, dgamma,
                               dbeta, N, C, accumulate);
        return launch_status();
    }
    hipLaunchKernelGGL(norm_bwd_rowsums_kernel, dim3(row_grid(rg)), dim3(PW_THREADS), 0, stream, gout, x, coef,
                       (f32x2*)workspace, rg, C, per_channel, g.ncoef, act, slope, g.group_rows);
    if (per_channel && dx && !unfused) {
        const BnFuse f = bn_fuse_geom(N, C, inner, groups);
        hipLaunchKernelGGL(bn_bwd_apply_fused_kernel, dim3((C / f.CB) * groups * f.S), dim3(PW_THREADS), 0, stream, gout, x, coef,
                           (const f32x2*)workspace, kbuf, dgamma, dbeta, dx, f, inner, act, slope, accumulate);
        return launch_status();
    }
    if (per_channel) {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, stream, (const f32x2*)workspace, kbuf, dgamma,
                           dbeta, N, C, inner, groups, accumulate);
    } else {
        int rows = N * C;
        hipLaunchKernelGGL(row_bwd_finalize_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream,
                           (const f32x2*)workspace, kbuf, dgamma, dbeta, N, C, inner, affine_per_row, unbiased);
        if (!affine_per_row && (dgamma || dbeta))
            hipLaunchKernelGGL(row_bwd_affine_kernel, dim3(C), dim3(64), 0, stream, (const f32x2*)workspace, dgamma,
                               dbeta, N, C, accumulate);
    }
    if (dx)
        hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(ew_grid(g.total4)), dim3(PW_THREADS), 0, stream, gout, x, coef,
                           kbuf, dx, g, act, slope);
    return launch_status();
}

int gz_rownorm_act_bwd2(const float* gout, const float* v, const float* x, const float* coef, float* gg_out,
                        float* gx, float* ggamma, void* workspace, int N, int C, int inner, int act, float slope,
                        hipStream_t stream) {
    gz::clear_stale_error();
    if (!norm_shape_ok(N, C, inner)) return GZ_ERR_BAD_SHAPE;
    RowGeom rg = row_geom((long long)N * C, inner);
    ApplyGeom g = apply_geom(N, C, inner, 0);
    hipLaunchKernelGGL(norm_bwd2_rowsums_kernel, dim3(row_grid(rg)), dim3(PW_THREADS), 0, stream, gout, v, x, coef,
                       (Sums5*)workspace, rg, act, slope);
    if (gg_out || gx)
        hipLaunchKernelGGL(norm_bwd2_apply_kernel, dim3(ew_grid(g.total4)), dim3(PW_THREADS), 0, stream, gout, v, x,
                           coef, (const Sums5*)workspace, gg_out, gx, g, act, slope);
    if (ggamma)
        hipLaunchKernelGGL(norm_bwd2_ggamma_kernel, dim3(C), dim3(64), 0, stream, (const Sums5*)workspace, coef, ggamma,
                           N, C, inner);
    return launch_status();
}

int gz_act_bwd(const float* g, const float* out, float* dx, long long count, int act, float slope,
               hipStream_t stream) {
    gz::clear_stale_error();
    if (count <= 0 || (count & 3)) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(count / 4)), dim3(PW_THREADS), 0, stream, g, out, dx, count / 4,
                       act, slope);
    return launch_status();
}

int gz_tanh_bwd2(const float* v, const float* g, const float* out, float* res, long long count, hipStream_t stream) {
    gz::clear_stale_error();
    if (count <= 0 || (count & 3)) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(tanh_bwd2_kernel, dim3(ew_grid(count / 4)), dim3(PW_THREADS), 0, stream, v, g, out, res,
                       count / 4);
    return launch_status();
}

}  // extern "C"
