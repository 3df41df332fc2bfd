// Timing harness for the igemm2 skeleton with the transposed-convolution loaders (no packing, random data, results
// not checked: correctness lives in tests/test_ops_gpu.py): per-workgroup s_memtime stamps -> prologue / main loop /
// epilogue cycles, with the shader clock warmed up.  Experiment macros of csrc/gz_igemm.h (-DGZ2_EXP_*) apply.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGZ2_STAMPS -I include -I lightning_gan_zoo_amd/csrc \
//         -o tools/bin/igemm2_conv_probe tools/igemm2_conv_probe.hip
//   tools/bin/igemm2_conv_probe N K OH C [tile: 128|256] [lds_extra]
#include "gz_igemm.h"
#include <cstdio>
#include <vector>
using namespace gz;

template <class Cfg>
static void run(int N, int K, int OH, int C, size_t lds_extra) {
    using AL = ConvDgA2<Cfg::BM>;
    using BL = MContigB2<Cfg::BN>;
    using Epi = EpiPhaseB<2>;
    const int OW = OH, H = 2 * OH, W = 2 * OW;
    ConvShape s{N, C, H, W, K, OH, OW};
    const int AH = OH, AW = OW, Kg = K * 4, ldc = (C + 3) & ~3;
    const long long M = (long long)N * AH * AW;
    float *y, *wp, *x;
    size_t ny = (size_t)N * K * OH * OW, nw = (size_t)4 * Kg * ldc, nx = (size_t)N * C * H * W;
    hipMalloc(&y, ny * 4); hipMalloc(&wp, nw * 4); hipMalloc(&x, nx * 4);
    std::vector<float> h(ny > nw ? ny : nw);
    uint32_t r = 7;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    hipMemcpy(y, h.data(), ny * 4, hipMemcpyHostToDevice);
    hipMemcpy(wp, h.data(), nw * 4, hipMemcpyHostToDevice);
    typename AL::Params pa{y, s, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW)};
    typename BL::Params pb{wp, Kg, ldc, ldc, (long long)Kg * ldc};
    typename Epi::Params pe{x, (int)M, C, H, W, AH, AW, make_fastdiv(AH * AW), make_fastdiv(AW), nullptr, 0, 0.f, nullptr, 0};
    GridMap gm{};
    gm.tiles_m = (int)((M + Cfg::BM - 1) / Cfg::BM); gm.tiles_n = (C + Cfg::BN - 1) / Cfg::BN; gm.ny = 4;
    gm.chunks = Kg / BK; gm.chunks_per_split = gm.chunks; gm.slab = nullptr; gm.slab_m = (int)M; gm.slab_n = C;
    gm.stagger = getenv("STAGGER") ? (int)((long long)gm.chunks * 8 * Cfg::TM * Cfg::TN * 64 * 2 * atoi(getenv("STAGGER")) / 100) : 0;
    dim3 grid(gm.tiles_m * gm.tiles_n * 4);
    size_t lds = igemm2_lds_bytes<Cfg, AL, BL>() + lds_extra;
    auto kern = igemm2_kernel<Cfg, AL, BL, Epi>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const double flop = 2.0 * M * C * Kg * 4;
    int warm = (int)(0.4 / (flop / 120e12)) + 3;
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(kern, grid, dim3(NT), lds, 0, pa, pb, pe, gm);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(NT), lds, 0, pa, pb, pe, gm);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    int nwg = grid.x < 8192 ? grid.x : 8192;
    std::vector<unsigned long long> st((size_t)nwg * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(gz2_stamps), st.size() * 8, 0, hipMemcpyDeviceToHost);
    double pro = 0, loop = 0, epi = 0, clk = 0;
    for (int i = 0; i < nwg; ++i) {
        auto* o = &st[(size_t)i * 8];
        pro += o[1] - o[0]; loop += o[2] - o[1]; epi += o[3] - o[2]; clk += (double)(o[3] - o[0]) / (o[5] - o[4]) * 100;
    }
#ifdef GZ2_STEP_STAMPS
    {
        std::vector<unsigned long long> ss(64 * 8);
        hipMemcpyFromSymbol(ss.data(), HIP_SYMBOL(gz2_stamps), ss.size() * 8, (size_t)(8192 - 64) * 64, hipMemcpyDeviceToHost);
        printf("   cycles per k-step position (avg over chunks, workgroup 0 / 1): ");
        for (int w = 0; w < 2; ++w) { for (int q = 0; q < 8; ++q) printf("%.0f ", (double)ss[w * 8 + q] / gm.chunks); printf(" | "); }
        printf("(ideal %d)\n", Cfg::TM * Cfg::TN * 64);
    }
#endif
    const double ideal = (double)gm.chunks * 8 * Cfg::TM * Cfg::TN * 64;
    printf("N %d K %d OH %d C %d tile %dx%d lds %zu: %.4f ms %.1f TFLOP/s | wg %d prologue %.0f loop %.0f (ideal %.0f, %.3f; %.0f extra/chunk) "
           "epilogue %.0f clock %.0f MHz%s\n", N, K, OH, C, Cfg::BM, Cfg::BN, lds, ms, flop / ms / 1e9, (int)grid.x, pro / nwg, loop / nwg, ideal,
           ideal / (loop / nwg), (loop / nwg - ideal) / gm.chunks, epi / nwg, clk / nwg, hipGetLastError() == hipSuccess ? "" : " HIP ERROR");
}

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 512, K = argc > 2 ? atoi(argv[2]) : 256, OH = argc > 3 ? atoi(argv[3]) : 16,
        C = argc > 4 ? atoi(argv[4]) : 128, tile = argc > 5 ? atoi(argv[5]) : 128;
    size_t extra = argc > 6 ? (size_t)atoi(argv[6]) : 0;
    if (tile == 256) run<TileCfg2<2, 2, 4, 1>>(N, K, OH, C, extra);
    else run<TileCfg2<2, 2, 2, 2>>(N, K, OH, C, extra);
    return 0;
}
