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

// Small per-step host tensors (latent noise, GP alpha, HoloGAN's view matrices) are read by the device straight out of
// the PINNED host buffer they were drawn into (hipHostMalloc'ed memory is mapped into the device's address space):
// a plain kernel on the compute stream instead of hipMemcpyAsync, whose enqueue costs the host 180-500 us while the
// stream is busy (measured inside the bs-128 step; 50 us when idle).
__global__ __launch_bounds__(MT) void copy_words_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst,
                                                        long long n) {
    const long long stride = (long long)gridDim.x * MT;
    for (long long i = (long long)blockIdx.x * MT + threadIdx.x; i < n; i += stride) dst[i] = src[i];
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

// ---- several small Linear layers over ONE input, one launch --------------------------------------------------
// HoloGAN's generator feeds the same z through five ZMapping layers (Linear(128 -> 2C) + ReLU, reference
// core/models/hologan_generator.py:7-19,33,57,141): 33 MFLOP in all, which as five GEMM launches forward and
// fifteen launches backward (mask, dW, db each) cost ~0.2 ms of launch-bound time per generator pass.  A job table
// passed by value carries the layers; plain FMA tiles through LDS are enough at this size.
constexpr int LIN_MAX_JOBS = 8;
struct LinJob {
    const float* w;        // [J][K]
    const float* b;        // [J] or null
    float* out;            // [N][J]   forward: written; backward: read for the activation's derivative
    const float* g;        // [N][J]   backward only
    float* dw;             // [J][K]
    float* db;             // [J] or null
    int J, block0;
};
struct LinTable {
    int njobs, pad;
    LinJob jobs[LIN_MAX_JOBS];
};

__device__ __forceinline__ const LinJob& lin_job(const LinTable& t, int b) {
    int j = 0;
    while (j + 1 < t.njobs && t.jobs[j + 1].block0 <= b) ++j;
    return t.jobs[j];
}

// out_j[n][o] = act(sum_k x[n][k] * w_j[o][k] + b_j[o]); a workgroup owns 64 outputs x 64 rows of one job, a lane
// 4 x 4 of them (k-major LDS images: two 16-byte LDS reads per 16 FMAs)
__global__ __launch_bounds__(MT) void linear_multi_fwd_kernel(LinTable t, const float* __restrict__ x, int N, int K,
                                                              int act, float slope) {
    __shared__ __attribute__((aligned(16))) float xs[32][68], ws[32][68];
    const LinJob& jb = lin_job(t, blockIdx.x);
    const int j0 = (blockIdx.x - jb.block0) * 64, n0 = blockIdx.y * 64;
    const int tj = threadIdx.x & 15, tn = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 32) {
        float xr[8], wr[8];          // all sixteen loads first (clamped addresses), the edge zeros on the way to LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + MT * i, r = idx >> 5, kc = min(k0 + (idx & 31), K - 1);
            xr[i] = x[(long long)min(n0 + r, N - 1) * K + kc];
            wr[i] = jb.w[(long long)min(j0 + r, jb.J - 1) * K + kc];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + MT * i, r = idx >> 5, c = idx & 31;
            const bool kin = k0 + c < K;
            xs[c][r] = (kin && n0 + r < N) ? xr[i] : 0.f;
            ws[c][r] = (kin && j0 + r < jb.J) ? wr[i] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < 32; ++kk) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[kk][tn * 4]);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&ws[kk][tj * 4]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(wv[a], xv[b], acc[a][b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int j = j0 + tj * 4 + a;
        if (j >= jb.J) continue;
        const float bv = jb.b ? jb.b[j] : 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n = n0 + tn * 4 + b;
            if (n < N) jb.out[(long long)n * jb.J + j] = act_fwd(acc[a][b] + bv, act, slope);
        }
    }
}

// dw_j[o][k] = sum_n gm[n][o] * x[n][k], db_j[o] = sum_n gm[n][o], gm = g_j * act'(out_j); a workgroup owns
// 64 outputs x 64 inputs of one job and walks the rows in a fixed order
__global__ __launch_bounds__(MT) void linear_multi_bwd_kernel(LinTable t, const float* __restrict__ x, int N, int K,
                                                              int act, float slope) {
    __shared__ __attribute__((aligned(16))) float gs[32][68], xs[32][68];
    const LinJob& jb = lin_job(t, blockIdx.x);
    const int j0 = (blockIdx.x - jb.block0) * 64, k0 = blockIdx.y * 64;
    const int tk = threadIdx.x & 15, tj = threadIdx.x >> 4;
    float acc[4][4] = {}, bacc[4] = {};
    for (int n0 = 0; n0 < N; n0 += 32) {
        float gr[8], orr[8], xr[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + MT * i, r = min(n0 + (idx >> 6), N - 1), c = idx & 63;
            const long long at = (long long)r * jb.J + min(j0 + c, jb.J - 1);
            gr[i] = jb.g[at];
            orr[i] = act != ACT_NONE ? jb.out[at] : 1.f;
            xr[i] = x[(long long)r * K + min(k0 + c, K - 1)];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = threadIdx.x + MT * i, r = idx >> 6, c = idx & 63;
            const bool nin = n0 + r < N;
            gs[r][c] = (nin && j0 + c < jb.J) ? gr[i] * act_bwd_from_out(orr[i], act, slope) : 0.f;
            xs[r][c] = (nin && k0 + c < K) ? xr[i] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int n = 0; n < 32; ++n) {
            const f32x4 gv = *reinterpret_cast<const f32x4*>(&gs[n][tj * 4]);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[n][tk * 4]);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bacc[a] += gv[a];
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(gv[a], xv[b], acc[a][b]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int j = j0 + tj * 4 + a;
        if (j >= jb.J) continue;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = k0 + tk * 4 + b;
            if (k < K) jb.dw[(long long)j * K + k] = acc[a][b];
        }
        if (jb.db && blockIdx.y == 0 && tk == 0) jb.db[j] = bacc[a];
    }
}

}  // namespace gz

using namespace gz;

extern "C" {

int gz_linear_multi_max_jobs(void) { return LIN_MAX_JOBS; }
size_t gz_linear_multi_table_bytes(void) { return sizeof(LinTable); }

int gz_linear_multi_add(void* table_host, const float* weight, const float* bias, float* out, const float* g, float* dw,
                        float* db, int J) {
    LinTable* t = reinterpret_cast<LinTable*>(table_host);
    if (!t || !out || J <= 0) return GZ_ERR_BAD_SHAPE;
    if (t->njobs < 0 || t->njobs >= LIN_MAX_JOBS) return GZ_ERR_UNSUPPORTED;
    t->jobs[t->njobs++] = LinJob{weight, bias, out, g, dw, db, J, 0};
    return GZ_OK;
}

static int linear_multi_launch(void* table_host, const float* x, int N, int K, int act, float slope, bool bwd,
                               hipStream_t stream) {
    gz::clear_stale_error();
    LinTable* t = reinterpret_cast<LinTable*>(table_host);
    if (!t || !x || N <= 0 || K <= 0 || t->njobs <= 0 || t->njobs > LIN_MAX_JOBS) return GZ_ERR_BAD_SHAPE;
    if (act < GZ_ACT_NONE || act > GZ_ACT_TANH) return GZ_ERR_BAD_SHAPE;
    long long blocks = 0;
    for (int j = 0; j < t->njobs; ++j) {
        const LinJob& jb = t->jobs[j];
        if (bwd ? (!jb.g || !jb.dw) : !jb.w) return GZ_ERR_BAD_SHAPE;
        t->jobs[j].block0 = (int)blocks;
        blocks += (jb.J + 63) / 64;
    }
    const long long by = bwd ? (K + 63) / 64 : (N + 63) / 64;
    if (blocks >= (1ll << 31) || by > 65535) return GZ_ERR_TOO_LARGE;
    if (bwd)
        hipLaunchKernelGGL(linear_multi_bwd_kernel, dim3((unsigned)blocks, (unsigned)by), dim3(MT), 0, stream, *t, x, N, K,
                           act, slope);
    else
        hipLaunchKernelGGL(linear_multi_fwd_kernel, dim3((unsigned)blocks, (unsigned)by), dim3(MT), 0, stream, *t, x, N, K,
                           act, slope);
    return launch_status();
}

int gz_linear_multi_fwd(void* table_host, const float* x, int N, int K, int act, float slope, hipStream_t stream) {
    return linear_multi_launch(table_host, x, N, K, act, slope, false, stream);
}

int gz_linear_multi_bwd(void* table_host, const float* x, int N, int K, int act, float slope, hipStream_t stream) {
    return linear_multi_launch(table_host, x, N, K, act, slope, true, stream);
}

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

int gz_coldot_partial(const float* g, const float* x, float* out, float* workspace, size_t ws_bytes, int R, int L,
                      int* nz_out, hipStream_t stream) {
    gz::clear_stale_error();
    if (R <= 0 || L <= 0 || (L & 3) || !nz_out) return GZ_ERR_BAD_SHAPE;
    int slices = R >= 64 ? 32 : 1;
    if (slices > 1 && ws_bytes < (size_t)slices * L * 4) return GZ_ERR_WORKSPACE;
    int rps = (R + slices - 1) / slices;
    int L4 = L / 4;
    hipLaunchKernelGGL(coldot_kernel, dim3((L4 + MT - 1) / MT, slices), dim3(MT), 0, stream, g, x,
                       slices > 1 ? workspace : out, R, L4, rps);
    *nz_out = slices;
    return launch_status();
}

int gz_copy_words(const void* src, void* dst, long long words, hipStream_t stream) {
    gz::clear_stale_error();
    if (words <= 0 || !src || !dst) return GZ_ERR_BAD_SHAPE;
    if (((unsigned long long)src | (unsigned long long)dst) & 3ull) return GZ_ERR_BAD_SHAPE;
    hipLaunchKernelGGL(copy_words_kernel, dim3(grid_for(words)), dim3(MT), 0, stream, (const unsigned*)src,
                       (unsigned*)dst, words);
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
