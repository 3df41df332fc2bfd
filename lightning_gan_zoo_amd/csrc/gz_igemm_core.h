// Part of the fp32 implicit-GEMM core (see gz_igemm.h): the round-1/2 kernel (four co-resident workgroups per CU,
// register-staged loaders, two LDS buffers), its launcher and the split-K finish kernels.
#pragma once
#include "gz_igemm_epilogues.h"

namespace gz {

// ---------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------
struct GridMap {
    int tiles_m, tiles_n, ny;  // grid.x = tiles_m * tiles_n * ny (ny = dgrad phases / batches, fastest)
    int chunks;                // total K chunks
    int chunks_per_split;      // grid.z = ceil(chunks / chunks_per_split)
    // split-K of an op with a structured epilogue (F / Dg / GEMM with few output tiles): every (y, z)
    // workgroup writes its raw accumulators to slab[y * gridDim.z + z][slab_m][slab_n] (row-major) and
    // splitk_finish_kernel<Epi> sums the slabs and applies the real epilogue.  Null: Epi::store directly.
    float* slab;
    int slab_m, slab_n;
    // transposed convolutions whose phases have different tap counts (k3 s2, k5 s2): chunks of phase y
    int var_chunks;
    int phase_chunks[8];
    int phase_order[8];        // phases by decreasing chunk count
    // split-K of such a launch: phase y owns phase_nz[y] = ceil(phase_chunks[y] / chunks_per_split) slabs starting at
    // slab phase_slab0[y]; the (y, z) workgroups past a short phase's last slab exit at once, so that every
    // remaining workgroup carries about the same number of chunks (with one slab count for all phases the 8-tap
    // phase of the 4^3 -> 8^3 transposed convolution ran 8x longer than the 1-tap phase beside it)
    int phase_nz[8];
    int phase_slab0[8];
    int no_swizzle;            // experiment (GZ_NO_XCD_SWIZZLE): plain blockIdx order
    int stagger;               // igemm2: shader cycles by which the first-round workgroups in odd CU slots start late
};

// SWAP (transposed accumulators, lanes along m): the slab is kept [n][m] so that its stores and the finish
// kernel's loads stay contiguous along the lanes; otherwise [m][n].
template <bool SWAP, int TM, int TN>
__device__ __forceinline__ void store_slab(const GridMap& gm, f32x16 (&acc)[TM][TN], int m_base, int n_base, int lane,
                                           int slab_idx) {
    float* c = gm.slab + (long long)slab_idx * gm.slab_m * gm.slab_n;
    const int col_l = lane & 31, half = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
                if constexpr (SWAP) {
                    int m = m_base + i * 32 + col_l, n = n_base + j * 32 + rr;
                    if (m < gm.slab_m && n < gm.slab_n) c[(long long)n * gm.slab_m + m] = acc[i][j][r];
                } else {
                    int m = m_base + i * 32 + rr, n = n_base + j * 32 + col_l;
                    if (m < gm.slab_m && n < gm.slab_n) c[(long long)m * gm.slab_n + n] = acc[i][j][r];
                }
            }
        }
    }
}

#ifndef GZ_IGEMM_INTERLEAVE
#define GZ_IGEMM_INTERLEAVE 0
#endif
// __launch_bounds__'s second argument (minimum waves per SIMD).  With the default of 1 hipcc gave the 128x128
// kernels 106 VGPRs + 64 AGPRs = 170 registers, i.e. TWO waves per SIMD; asked for 4 it keeps the accumulators in
// the same 106 VGPRs (no AGPRs, no spill) and four workgroups share a CU: +7...8 % on the G.block3 / block4-sized
// launches (F 122 -> 130, Dg 114 -> 124 TFLOP/s), nothing lost on the smaller ones.
#ifndef GZ_IGEMM_WAVES_PER_SIMD
#define GZ_IGEMM_WAVES_PER_SIMD 4
#endif

template <class T, class = void>
struct is_fwdrows : std::false_type {};
template <class T>
struct is_fwdrows<T, std::void_t<decltype(T::FWDROWS)>> : std::true_type {};
template <class T, class = void>
struct is_rowshare : std::false_type {};
template <class T>
struct is_rowshare<T, std::void_t<decltype(T::ROWSHARE)>> : std::true_type {};
// (round 6) a row-shared loader whose reduction has a SECOND part of plain chunks (ConvDg5A2: the third horizontal tap of
// the 5x5 s2 p2 transposed convolution's even columns), see igemm2_kernel
template <class T, class = void>
struct is_dualmode : std::false_type {};
template <class T>
struct is_dualmode<T, std::void_t<decltype(T::DUALMODE)>> : std::true_type {};

template <class Cfg, class AL, class BL, class Epi>
__global__ __launch_bounds__(NT, GZ_IGEMM_WAVES_PER_SIMD) void igemm_kernel(typename AL::Params pa, typename BL::Params pb,
                                                   typename Epi::Params pe, GridMap gm) {
    constexpr int LDA = AL::LD, LDB = BL::LD;
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    // row-shared A images are read one column to the left / right of the tile: keep them behind the B images so that
    // such a (masked) read stays inside this workgroup's allocation
    constexpr bool RS0 = is_rowshare<AL>::value || is_fwdrows<AL>::value;
    float* As = RS0 ? smem + 2 * BK * LDB : smem;
    float* Bs = RS0 ? smem : smem + 2 * BK * LDA;

    const int tid = threadIdx.x;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of tiles (neighbouring tiles share operand panels in that XCD's L2).
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    int y;
    if (gm.var_chunks) {
        // phases of unequal length (1..8 taps): longest phases first over ALL tiles, so that the workgroups still
        // running when the grid drains are the short ones (with the phases of a tile adjacent, the 8-tap phase of
        // the last tiles ran alone for a quarter of the kernel).  Plain round-robin over the XCDs keeps every
        // XCD's mix of phases equal.
        const int tiles = gm.tiles_m * gm.tiles_n;
        y = gm.phase_order[bid / tiles];
        bid %= tiles;
    } else {
        if (!gm.no_swizzle) {
            const int q = nwg >> 3, rr = nwg & 7, x = bid & 7, i = bid >> 3;
            bid = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + i;
        }
        // phases of one tile are adjacent (same XCD, close in time): they read the same feature patch and
        // their interleaved stores meet in that XCD's L2.
        y = bid % gm.ny;
        bid /= gm.ny;
    }
    const int tile_n = bid % gm.tiles_n;
    const int tile_m = bid / gm.tiles_n;
    const int z = blockIdx.z;
    const int kc0 = z * gm.chunks_per_split;
    const int kc1 = min(gm.var_chunks ? gm.phase_chunks[y] : gm.chunks, kc0 + gm.chunks_per_split);

    if (gm.slab && gm.var_chunks && kc0 >= kc1) return;      // past this phase's last slab (uniform per workgroup)

    AL al;
    BL bl;
    al.init(pa, tile_m, y, tid);
    bl.init(pb, tile_n, y, tid);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int half = lane >> 5, l32 = lane & 31;
    constexpr bool RS = is_rowshare<AL>::value;       // ConvDgALoaderRow4: one LDS row per k-step, taps applied on read
    int a_rd = half * LDA + wm * TM * 32 + l32;
    constexpr int A_STEP = RS ? LDA : 2 * LDA;
    bool a_zero[TM];
    if constexpr (RS) {
        const int sh = al.frag_shift(half);
        a_rd = wm * TM * 32 + l32 + sh;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int b = (tile_m * Cfg::BM + wm * TM * 32 + i * 32 + l32) % pa.AW;      // m = (n * AH + a) * AW + b
            a_zero[i] = (sh < 0 && b == 0) || (sh > 0 && b == pa.AW - 1);
        }
    }
    constexpr bool FR = is_fwdrows<AL>::value;        // ConvFwdALoaderRow4: raw input rows, taps applied on read
    int fr_base[TM];
    bool fr_ze[TM], fr_zo[TM];
    int fr_2ow = 0;
    if constexpr (FR) {
        fr_2ow = al.twoOW;
#pragma unroll
        for (int i = 0; i < TM; ++i) al.frag(wm * TM * 32 + i * 32 + l32, half, fr_base[i], fr_ze[i], fr_zo[i]);
    }
    const int b_rd = half * LDB + wn * TN * 32 + l32;

    if (kc0 < kc1) {
        if constexpr (AL::DMA) al.issue_lds(kc0, As); else al.issue(kc0);
        if constexpr (BL::DMA) bl.issue_lds(kc0, Bs); else bl.issue(kc0);
        al.commit(As);
        bl.commit(Bs);
    }
    __syncthreads();     // (waits for outstanding LDS-DMA too: hipcc emits vmcnt(0) in front of the barrier)

    for (int kc = kc0; kc < kc1; ++kc) {
        const int cur = (kc - kc0) & 1;
        const bool more = kc + 1 < kc1;
        constexpr bool INTERLEAVE = GZ_IGEMM_INTERLEAVE && AL::DMA && BL::DMA;
#if !defined(GZ_EXP_NOLOAD) && !defined(GZ_EXP_NOISSUE)   // timing experiments only (wrong results)
        if (more && !INTERLEAVE) {
            // the other LDS buffer was last read in the previous iteration, behind its closing barrier
#ifdef GZ_EXP_SAMECHUNK      // timing experiment: always re-load chunk kc0 (cache-resident)
            if constexpr (AL::DMA) al.issue_lds(kc0, As + (cur ^ 1) * BK * LDA); else al.issue(kc0);
            if constexpr (BL::DMA) bl.issue_lds(kc0, Bs + (cur ^ 1) * BK * LDB); else bl.issue(kc0);
#else
            if constexpr (AL::DMA) al.issue_lds(kc + 1, As + (cur ^ 1) * BK * LDA); else al.issue(kc + 1);
            if constexpr (BL::DMA) bl.issue_lds(kc + 1, Bs + (cur ^ 1) * BK * LDB); else bl.issue(kc + 1);
#endif
        }
#endif
        const float* Ar = As + cur * BK * LDA + a_rd;
        const float* Br = Bs + cur * BK * LDB + b_rd;
        // fragment double buffering in registers: the ds_reads of k-step s+1 are in flight while the
        // TM*TN MFMAs of k-step s issue, so a wave does not depend on its SIMD neighbours to cover
        // the LDS latency
        float af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if constexpr (FR) {
                const float v = (As + cur * BK * LDA)[fr_base[i]];
                af[0][i] = fr_ze[i] ? 0.f : v;
            } else {
                af[0][i] = Ar[i * 32];
                if constexpr (RS) af[0][i] = a_zero[i] ? 0.f : af[0][i];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = Br[j * 32];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int c = s & 1, n = c ^ 1;
            if (s + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if constexpr (FR) {
                        const float v = (As + cur * BK * LDA)[fr_base[i] + ((s + 1) >> 1) * fr_2ow + 2 * ((s + 1) & 1)];
                        af[n][i] = (((s + 1) & 1) ? fr_zo[i] : fr_ze[i]) ? 0.f : v;
                    } else {
                        af[n][i] = Ar[(s + 1) * A_STEP + i * 32];
                        if constexpr (RS) af[n][i] = a_zero[i] ? 0.f : af[n][i];
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[n][j] = Br[2 * (s + 1) * LDB + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if constexpr (Epi::SWAP)      // D'[n][m]: lanes along m (see EpiNCHW)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[c][j], af[c][i], acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i], bf[c][j], acc[i][j], 0, 0, 0);
            // pin the order "next step's LDS reads, then this step's MFMAs" (hipcc otherwise sinks the
            // reads behind the MFMAs to save two registers)
            if constexpr (INTERLEAVE) {
                // spread the staging loads of chunk kc+1 over the k-steps so that no wave has a long
                // MFMA-free stretch at the top of the chunk (co-resident workgroups run in lockstep)
                if (more) {
                    if constexpr (AL::NPARTS > 0) {
                        constexpr int PER = (AL::NPARTS + BK / 2 - 1) / (BK / 2);
#pragma unroll
                        for (int q = 0; q < PER; ++q)
                            if (s * PER + q < AL::NPARTS) al.issue_lds_part(kc + 1, As + (cur ^ 1) * BK * LDA, s * PER + q);
                    }
                    if constexpr (BL::NPARTS > 0) {
                        constexpr int STRIDE = (BK / 2) / BL::NPARTS;
                        if (s % STRIDE == 0 && s / STRIDE < BL::NPARTS)
                            bl.issue_lds_part(kc + 1, Bs + (cur ^ 1) * BK * LDB, s / STRIDE);
                    }
                }
            }
#ifndef GZ_IGEMM_NO_FRAG_PREFETCH
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);   // MFMA
            if constexpr (INTERLEAVE) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);   // VMEM reads of this step
#endif
        }
#if !defined(GZ_EXP_NOLOAD) && !defined(GZ_EXP_NOCOMMIT)
        if (more) {
            al.commit(As + (cur ^ 1) * BK * LDA);
            bl.commit(Bs + (cur ^ 1) * BK * LDB);
        }
#endif
#ifndef GZ_EXP_NOBARRIER
        __syncthreads();
#endif
    }

#ifdef GZ_EXP_NOSTORE   // timing experiment: keep one store so the accumulators stay live
    if (acc[0][0][0] == 123456.789f)
#endif
    if (gm.slab)
        store_slab<Epi::SWAP, TM, TN>(gm, acc, tile_m * Cfg::BM + wm * TM * 32, tile_n * Cfg::BN + wn * TN * 32, lane,
                           gm.var_chunks ? gm.phase_slab0[y] + z : y * (int)gridDim.z + z);
    else
        Epi::template store<TM, TN>(pe, acc, tile_m * Cfg::BM + wm * TM * 32, tile_n * Cfg::BN + wn * TN * 32,
                                    lane, y, z);
}

struct SlabMap {      // per-phase slab ranges of a split launch with phases of unequal length (GridMap::phase_nz)
    int var;
    int nz[8];
    int slab0[8];
};

// Second pass of a split-K launch: a workgroup owns one 32x32 output block; its WAVES wavefronts each sum every
// WAVES-th slab into the accumulator image Epi::store expects (the MFMA C/D layout), the partial images meet in
// LDS in a fixed order, and wavefront 0 runs the op's own epilogue (bias, activation, NCHW / phase scatter).
template <class Epi, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void splitk_finish_kernel(const float* __restrict__ slab, int nz, int M, int N,
                                                                   typename Epi::Params pe, int tiles_n, int ny,
                                                                   SlabMap sm) {
    __shared__ float part[WAVES - 1][16][64];
    int bid = blockIdx.x;
    const int y = bid % ny;
    bid /= ny;
    const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m_base = tile_m * 32, n_base = tile_n * 32;
    const int col_l = lane & 31, half = lane >> 5;
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    const int slab0 = sm.var ? sm.slab0[y] : y * nz;
    if (sm.var) nz = sm.nz[y];
    for (int z = wave; z < nz; z += WAVES) {
        const float* c = slab + (long long)(slab0 + z) * M * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
            if constexpr (Epi::SWAP) {
                int m = m_base + col_l, n = n_base + rr;
                if (m < M && n < N) acc[0][0][r] += c[(long long)n * M + m];
            } else {
                int m = m_base + rr, n = n_base + col_l;
                if (m < M && n < N) acc[0][0][r] += c[(long long)m * N + n];
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wave - 1][r][lane] = acc[0][0][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES - 1; ++w) t += part[w][r][lane];
            acc[0][0][r] += t;
        }
        Epi::template store<1, 1>(pe, acc, m_base, n_base, lane, y, 0);
    }
}

// The same second pass for the row-major epilogue when there are many slabs over few outputs (the nn.Linear heads:
// 64 x 128 outputs, K = 8192 / 32768): an output element per lane, four lanes per element taking every fourth slab
// (coalesced along the row), partial sums meeting in LDS in a fixed order.  The block-per-32x32 form above gave such a
// launch 8 workgroups walking 64+ slabs one after the other (27 us at K = 8192, 96 us at K = 32768).
template <int ZL>      // lanes per output element (a template so that the header can be included by several sources)
__global__ __launch_bounds__(64 * ZL) void splitk_finish_rowmajor_kernel(const float* __restrict__ slab, int nz,
                                                                        EpiRowMajor::Params pe, int ny) {
    static_assert(ZL == 4, "the LDS combine below is written for four slab lanes");
    __shared__ float part[3][64];
    const int lane = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const long long mn = (long long)pe.M * pe.N;
    const long long e = (long long)blockIdx.x * 64 + lane;
    const bool live = e < mn * ny;
    const long long y = live ? e / mn : 0, idx = live ? e - y * mn : 0;
    float s = 0.f;
    if (live) {
        const float* c = slab + y * nz * mn + idx;
        for (int z = zl; z < nz; z += 4) s += c[(long long)z * mn];
    }
    if (zl > 0) part[zl - 1][lane] = s;
    __syncthreads();
    if (zl == 0 && live) {
        s += (part[0][lane] + part[1][lane]) + part[2][lane];
        const int m = (int)(idx / pe.N), n = (int)(idx - (long long)m * pe.N);
        const float bv = pe.bias ? pe.bias[n] : 0.f;
        pe.c[y * pe.slab_stride + (long long)m * pe.ldc + n] = act_fwd(s + bv, pe.act, pe.slope);
    }
}

inline int split_nz(int K, int splits) {
    int chunks = (K + BK - 1) / BK;
    if (splits < 1) splits = 1;
    int per = (chunks + splits - 1) / splits;
    int nz = (chunks + per - 1) / per;
    return nz < 1 ? 1 : nz;
}

// `slab` != null: split-K through raw-accumulator slabs + splitk_finish_kernel<Epi> (needs
// ny * split_nz(K, splits) * M * N floats); null with splits > 1: the epilogue itself is slab-aware (Wg).
template <class Cfg, class AL, class BL, class Epi>
inline int launch_igemm(const typename AL::Params& pa, const typename BL::Params& pb,
                        const typename Epi::Params& pe, int M, int N, int K, int ny, int splits,
                        hipStream_t stream, float* slab = nullptr, const int* phase_chunks = nullptr) {
    GridMap gm;
    gm.no_swizzle = knobs().no_xcd_swizzle;
    gm.stagger = 0;
    gm.var_chunks = 0;
    if (phase_chunks && ny <= 8) {
        gm.var_chunks = 1;
        for (int i = 0; i < ny; ++i) {
            gm.phase_chunks[i] = phase_chunks[i];
            gm.phase_order[i] = i;
        }
        for (int i = 1; i < ny; ++i)          // insertion sort, stable, descending
            for (int j = i; j > 0 && gm.phase_chunks[gm.phase_order[j]] > gm.phase_chunks[gm.phase_order[j - 1]]; --j) {
                int t = gm.phase_order[j];
                gm.phase_order[j] = gm.phase_order[j - 1];
                gm.phase_order[j - 1] = t;
            }
    }
    gm.slab = nullptr;
    gm.slab_m = M;
    gm.slab_n = N;
    gm.tiles_m = (M + Cfg::BM - 1) / Cfg::BM;
    gm.tiles_n = (N + Cfg::BN - 1) / Cfg::BN;
    gm.chunks = (K + BK - 1) / BK;
    if (splits < 1) splits = 1;
    gm.chunks_per_split = (gm.chunks + splits - 1) / splits;
    int nz = (gm.chunks + gm.chunks_per_split - 1) / gm.chunks_per_split;
    if (nz < 1) nz = 1;
    gm.ny = ny;
    dim3 grid(gm.tiles_m * gm.tiles_n * ny, 1, nz);
    if (slab && nz > 1) gm.slab = slab;
    SlabMap sm;
    sm.var = gm.slab && gm.var_chunks;
    for (int i = 0, at = 0; i < 8; ++i) {
        int n = (sm.var && i < ny) ? (gm.phase_chunks[i] + gm.chunks_per_split - 1) / gm.chunks_per_split : 0;
        gm.phase_nz[i] = sm.nz[i] = n;
        gm.phase_slab0[i] = sm.slab0[i] = at;
        at += n;
    }
    const int dyn_lds = knobs().dyn_lds;   // experiment: throttle workgroups per CU
    hipLaunchKernelGGL((igemm_kernel<Cfg, AL, BL, Epi>), grid, dim3(NT), dyn_lds, stream, pa, pb, pe, gm);
    if (gm.slab) {
        const int fm = (M + 31) / 32, fn = (N + 31) / 32;
        if constexpr (std::is_same<Epi, EpiRowMajor>::value) {
            if (nz > 8) {
                const long long total = (long long)M * N * ny;
                hipLaunchKernelGGL(splitk_finish_rowmajor_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream,
                                   slab, nz, pe, ny);
                return launch_status();
            }
        }
        // many slabs over few output blocks: 8 wavefronts share the slab walk
        if (nz > 16)
            hipLaunchKernelGGL((splitk_finish_kernel<Epi, 8>), dim3(fm * fn * ny), dim3(512), 0, stream, slab, nz, M, N,
                               pe, fn, ny, sm);
        else
            hipLaunchKernelGGL((splitk_finish_kernel<Epi, 4>), dim3(fm * fn * ny), dim3(NT), 0, stream, slab, nz, M, N, pe,
                               fn, ny, sm);
    }
    return launch_status();
}

}  // namespace gz
