// The slab sum shared by gz_reduce_multi (gz_conv.hip) and the fused optimizers that read unreduced weight-gradient slabs
// directly (gz_optim.hip): ONE definition of the summation order, so that "reduce, then step" and "step from the slabs"
// give the same bits.  A workgroup of four wavefronts owns 64 float4: wavefront w takes slabs w, w+4, ... of source 0,
// then of source 1, ... (four independent accumulators per lane), the four partial sums meet in LDS in wavefront order.
// The result is valid in wavefront 0.
#pragma once
#include "gz_common.h"

namespace gz {

constexpr int REDUCE_MAX_SRC = 4;
struct ReduceSrc {
    const float* p;
    long long stride;          // floats between consecutive slabs
    int nz, pad;
};

__device__ __forceinline__ f32x4 reduce_sources(const ReduceSrc* src, int nsrc, long long i, bool live, int lane, int wave,
                                                f32x4 (*part)[64]) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (live) {
        for (int q = 0; q < nsrc; ++q) {
            const float* p = src[q].p + i;
            const long long st = src[q].stride;
            const int nz = src[q].nz;
            int z = wave;
            // eight independent 16-byte loads in flight per lane (round 5; four before): with hundreds of slabs a
            // wavefront's share is a chain of HBM round trips -- 512 slabs were 32 of them, now 16
            for (; z + 28 < nz; z += 32) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (long long)z * st);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (long long)(z + 4) * st);
                const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (long long)(z + 8) * st);
                const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (long long)(z + 12) * st);
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(p + (long long)(z + 16) * st);
                const f32x4 v5 = *reinterpret_cast<const f32x4*>(p + (long long)(z + 20) * st);
                const f32x4 v6 = *reinterpret_cast<const f32x4*>(p + (long long)(z + 24) * st);
                const f32x4 v7 = *reinterpret_cast<const f32x4*>(p + (long long)(z + 28) * st);
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
                a0 += v4; a1 += v5; a2 += v6; a3 += v7;
            }
            for (; z + 12 < nz; z += 16) {
                a0 += *reinterpret_cast<const f32x4*>(p + (long long)z * st);
                a1 += *reinterpret_cast<const f32x4*>(p + (long long)(z + 4) * st);
                a2 += *reinterpret_cast<const f32x4*>(p + (long long)(z + 8) * st);
                a3 += *reinterpret_cast<const f32x4*>(p + (long long)(z + 12) * st);
            }
            for (; z < nz; z += 4) a0 += *reinterpret_cast<const f32x4*>(p + (long long)z * st);
        }
    }
    a0 = (a0 + a1) + (a2 + a3);
    if (wave > 0) part[wave - 1][lane] = a0;
    __syncthreads();
    if (wave == 0 && live) return ((a0 + part[0][lane]) + part[1][lane]) + part[2][lane];
    return a0;
}

}  // namespace gz
