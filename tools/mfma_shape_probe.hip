// Which fp32 MFMA shape sustains more FLOP/s on random data from LDS: 32x32x2 (2x2 tiles per wave) or
// 16x16x4 (4x4 tiles per wave)?  Same 64x64 wave tile, same LDS bytes per FLOP.  (MI355X_MICROARCH.md,
// DVFS give-back item 7, reports a 1.15x advantage for the 16x16 shape in bf16.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 16;

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* __restrict__ out, int iters) {
    constexpr int LD = SHAPE == 32 ? 128 : 144;
    __shared__ float As[BK * 144], Bs[BK * 144];
    for (int i = threadIdx.x; i < BK * LD; i += 256) { As[i] = src[i]; Bs[i] = src[4096 + i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    if constexpr (SHAPE == 32) {
        f32x16 acc[2][2] = {};
        const float* Ar = As + (lane >> 5) * LD + wm * 64 + (lane & 31);
        const float* Br = Bs + (lane >> 5) * LD + wn * 64 + (lane & 31);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < BK / 2; ++s) {
                float a0 = Ar[2 * s * LD], a1 = Ar[2 * s * LD + 32], b0 = Br[2 * s * LD], b1 = Br[2 * s * LD + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x4 acc[4][4] = {};
        const float* Ar = As + (lane >> 4) * LD + wm * 64 + (lane & 15);
        const float* Br = Bs + (lane >> 4) * LD + wn * 64 + (lane & 15);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < BK / 4; ++s) {
                float a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[i] = Ar[4 * s * LD + i * 16]; b[i] = Br[4 * s * LD + i * 16]; }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

int main() {
    const int n = 8192;
    float* h = (float*)malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *src, *out;
    hipMalloc(&src, n * 4); hipMalloc(&out, 3072 * 256 * 4);
    hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    const int blocks = 256 * 3, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        for (int shape : {32, 16}) {
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
            else hipLaunchKernelGGL(probe<16>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)blocks * 128.0 * 128.0 * BK * 2.0 * iters;
            printf("shape %dx%d: %.3f ms  %.1f TFLOP/s\n", shape, shape, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
