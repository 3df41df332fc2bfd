// CU-contention rehearsal on ONE GPU: a bounded "channel" kernel that sits on R workgroup slots
// of the chip and streams a buffer, launched on a side stream while the training step runs on the main one -- what
// RCCL's channel kernels do underneath G's backward when the gradient exchange overlaps it.
//   R workgroups of 256 threads, `lds_bytes` of dynamic LDS each (RCCL's kernels keep their communicator state in
//   LDS: with >= 56 KB one igemm2 workgroup (48-52 KB) fits beside it on a CU instead of two), each streaming its
//   slice of `buf` (read + write, 16 bytes per lane) `iters` times at most, with `sleep` x 64 clocks of s_sleep
//   between passes (throttles the stream to a link-like rate); leaves early once *stop != 0.
// Bounded by construction: no unbounded spin, the host only ever shortens the run.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/bin/libgz_probe.so tools/contention_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void occupy_kernel(float* buf, unsigned long long floats_per_wg, int iters, int sleep,
                                                     const int* stop, unsigned long long* passes) {
    extern __shared__ float lds[];
    float4* p = reinterpret_cast<float4*>(buf + (unsigned long long)blockIdx.x * floats_per_wg);
    const unsigned long long n4 = floats_per_wg / 4;
    unsigned long long done = 0;
    for (int it = 0; it < iters; ++it) {
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
        for (unsigned long long i = threadIdx.x; i < n4; i += 256) {
            float4 v = p[i];
            v.x += 1.f;
            p[i] = v;
        }
        for (int s = 0; s < sleep; ++s) __builtin_amdgcn_s_sleep(64);
        ++done;
    }
    lds[threadIdx.x] = (float)done;
    if (threadIdx.x == 0) passes[blockIdx.x] = done;
}

extern "C" int gz_probe_occupy(float* buf, unsigned long long bytes, int R, int lds_bytes, int iters, int sleep,
                               const int* stop, unsigned long long* passes, hipStream_t stream) {
    if (R <= 0 || !buf || !stop || !passes) return -1;
    if (lds_bytes > 0 &&
        hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        return -2;
    const unsigned long long per = (bytes / 4 / R) & ~3ull;
    hipLaunchKernelGGL(occupy_kernel, dim3(R), dim3(256), (size_t)(lds_bytes > 1024 ? lds_bytes : 1024), stream, buf, per,
                       iters, sleep, stop, passes);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
