// Common definitions for the gfx950 (MI355X / CDNA4) kernels of the GAN training step.
// Everything here is fp32: the parity target is the reference's PyTorch-CPU fp32 path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GZ_OK 0
#define GZ_ERR_BAD_SHAPE (-1)
#define GZ_ERR_UNSUPPORTED (-2)
#define GZ_ERR_WORKSPACE (-3)
#define GZ_ERR_HIP (-4)
#define GZ_ERR_TOO_LARGE (-5)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace gz {

// activation codes shared by the C ABI and the kernels
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_TANH = 3 };

__device__ __forceinline__ float act_fwd(float v, int act, float slope) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_LRELU: return v > 0.f ? v : v * slope;
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// derivative of the activation expressed through its OUTPUT o (valid for slope > 0)
__device__ __forceinline__ float act_bwd_from_out(float o, int act, float slope) {
    switch (act) {
        case ACT_RELU: return o > 0.f ? 1.f : 0.f;
        case ACT_LRELU: return o > 0.f ? 1.f : slope;
        case ACT_TANH: return 1.f - o * o;
        default: return 1.f;
    }
}

// Unsigned division by a run-time invariant (n < 2^31), branch-free.
struct FastDiv {
    uint32_t d, mul, sh1, sh2;
};

inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    f.mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}

__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    uint32_t t = __umulhi(n, f.mul);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

inline const char*& last_error_slot() {
    static thread_local const char* msg = "";
    return msg;
}

inline int hip_status(hipError_t e) {
    if (e == hipSuccess) return GZ_OK;
    last_error_slot() = hipGetErrorString(e);
    return GZ_ERR_HIP;
}

inline int launch_status() { return hip_status(hipGetLastError()); }

// hipGetLastError is sticky per host thread: an unrelated, already-handled failure inside another
// library (e.g. a pointer-attribute query on host memory) would otherwise be reported by the next
// launcher.  Every C-ABI entry point clears it before its own launches.
inline void clear_stale_error() { (void)hipGetLastError(); }

}  // namespace gz
