// Part of the fp32 implicit-GEMM core (see gz_igemm.h): the igemm2 skeleton of rounds 3-5 (one wavefront per SIMD, LDS-DMA
// ring of three stages, hand-ordered k-step; igemm2_kernel / igemm2w_kernel / igemm2r_kernel, their loaders and launchers).
#pragma once
#include "gz_igemm_core.h"

namespace gz {

// =====================================================================================================================
// Round 3: the one-wavefront-per-SIMD skeleton (igemm2).
//
// Why: four co-resident 128x128 workgroups per CU (igemm_kernel) keep the matrix pipe busy 80-83 % of the time; the
// loop itself (no global loads) tops out at 0.87-0.91 because four wavefronts per SIMD arbitrate for one pipe and each
// of them stalls at a barrier every 32 MFMAs.  Here ONE wavefront per SIMD owns a 128 x 128 (or 128 x 64) accumulator
// tile -- a 256 x 256 / 256 x 128 workgroup tile -- and runs an in-order, software-pipelined stream in which the
// MFMAs issue back to back and everything else sits in their shadow:
//   * operands arrive by LDS-DMA into a THREE-deep ring of stages; the pieces of chunk t+2 are issued one per k-step
//     between the MFMAs of chunk t, waited for with a COUNTED s_waitcnt vmcnt(N) (never 0 inside the loop) and made
//     visible by ONE raw s_barrier per chunk (128 MFMAs per wavefront) -- __syncthreads() would drain the DMA queue;
//   * the wait + barrier sit inside the LAST k-step of a chunk, in front of its MFMAs, so the first fragments of the
//     next chunk are fetched under those MFMAs;
//   * fragments are double-buffered in registers one k-step ahead (8 ds_read_b32 per 16 MFMAs).
// tools/igemm2_probe.hip is this loop on a plain GEMM: 148-149 TFLOP/s (0.945 of the 157.3 peak) at K = 2048, main
// loop 98.2 % of the MFMA-issue bound (in-kernel s_memtime stamps), against 125-130 for igemm_kernel on the same size.
// Fragment / accumulator maps, LDS images ([k][m], m contiguous), GridMap, split-K slabs and the epilogues are the
// ones of igemm_kernel; the loaders are piece-wise:
//   static constexpr int LD, ROWS (LDS rows per chunk), PIECES (LDS-DMA instructions per wavefront and chunk);
//   void init(const Params&, int tile, int y, int tid);
//   void issue_piece(int kc, float* stage_base, int p, bool live);   // p in [0, PIECES), wave-uniform control flow
// =====================================================================================================================
constexpr int STAGES2 = 3;
constexpr uint32_t SOFF_OOB = 0x80000000u;      // scalar offset that puts every lane of a buffer access out of range

template <int WM_, int WN_, int TN_, int OCC_, int TM_ = 4>
struct TileCfg2 {
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, OCC = OCC_;    // OCC: workgroups per CU (= waves / SIMD)
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(WM * WN == 4, "4 wavefronts per workgroup, one per SIMD");
};

// B operand [K rows][ld], N contiguous (packed weights; plain GEMM): a piece is 256 consecutive floats of the
// [BK][BN] image = one k row (BN = 256) or two (BN = 128).  Rows past K are out of the descriptor's range (zeros).
template <int BN>
struct MContigB2 {
    static_assert(BN == 64 || BN == 128 || BN == 256, "piece mapping");
    using Params = typename MContigLoader<BN>::Params;
    static constexpr int LD = BN, ROWS = BK;
    static constexpr int PIECES = BK * BN / 256 / 4;      // per wavefront: 4 (BN 256), 2 (128), 1 (64: four k rows each)
    static constexpr int RPP = 256 / BN;                  // k rows per piece
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff, ldb;
    int wave;
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        rsrc = make_rsrc(p.base + (long long)y * p.batch_stride, (uint32_t)p.K * p.ld * 4u);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int f = lane * 4, r = f / BN, col = f % BN;
        const int n = tile * BN + col;
        ldb = (uint32_t)p.ld * 4u;
        voff = n < p.MN ? (uint32_t)(r * p.ld + n) * 4u : OOB;
    }
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {
        const int piece = wave * PIECES + p;
        bload_lds16(rsrc, stage + piece * 256, voff, live ? (uint32_t)(kc * BK + piece * RPP) * ldb : SOFF_OOB);
    }
};

// A operand of the k4 s2 p1 TRANSPOSED convolution, row-shared like ConvDgALoaderRow4 (a chunk's image is 8 rows =
// 4 feature channels x 2 vertical taps of UNSHIFTED feature rows; the horizontal tap is applied on the fragment
// read, the one column a shifted read takes from the neighbouring row is zeroed in the register).  One piece = one
// LDS row of 256 pixels; wavefront w stages channel w of the chunk, its two vertical taps.
template <int BM>
struct ConvDgA2 {
    static_assert(BM == 256 || BM == 512, "whole 256-pixel pieces per LDS row");
    using Params = typename ConvDgALoader<BM, 4, 4, 2, 1>::Params;
    // LDS rows carry 4 pad floats that are zeroed once and never written again: a lane whose shifted read would take
    // its value from the neighbouring image row reads that column instead (no v_cndmask in the loop -- an f32 MFMA
    // holds the SIMD's vector issue for its whole duration, so every VALU instruction between MFMAs costs its full
    // issue time)
    static constexpr int PPR = BM / 256;                   // pieces per LDS row
    static constexpr int LD = BM + 4, ROWS = BK / 2, PIECES = 2 * PPR;
    static constexpr int ZERO_COL = BM;
    static constexpr bool ROWSHARE = true;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[2 * PPR];       // [vertical tap][256-pixel block]
    int wave, K, OHW, shift_half1;
    __device__ __forceinline__ void init(const Params& p, int tile, int phase, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.y, (uint32_t)s.N * s.K * s.OH * s.OW * 4u);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        K = s.K; OHW = s.OH * s.OW;
        const int py = phase / 2, px = phase % 2;
        shift_half1 = (px + 1) / 2 - 1;
#pragma unroll
        for (int hb = 0; hb < PPR; ++hb) {
            const uint32_t m = (uint32_t)tile * BM + hb * 256 + lane * 4;
            const bool m_ok = m < (uint32_t)s.N * p.AH * p.AW;
            const uint32_t n = fdiv(m, p.div_ahw);
            const uint32_t pix = m - n * (uint32_t)(p.AH * p.AW);
            const uint32_t a = fdiv(pix, p.div_aw);
            const uint32_t b = pix - a * (uint32_t)p.AW;
#pragma unroll
            for (int ty = 0; ty < 2; ++ty) {
                const int oy = (int)a + (py + 1) / 2 - ty;
                const bool ok = m_ok && (unsigned)oy < (unsigned)s.OH;
                voff[ty * PPR + hb] = ok ? (n * (uint32_t)(s.K * OHW) + (uint32_t)(oy * s.OW) + b) * 4u : OOB;
            }
        }
    }
    __device__ __forceinline__ int frag_shift(int half) const { return half ? shift_half1 : shift_half1 + 1; }
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {     // p = ty * PPR + block
        const int kol = kc * (BK / 4) + wave;
        bload_lds16(rsrc, stage + (wave * 2 + p / PPR) * LD + (p % PPR) * 256, voff[p],
                    (live && kol < K) ? (uint32_t)kol * (uint32_t)OHW * 4u : SOFF_OOB);
    }
};

// A operand of the k4 s2 p1 FORWARD convolution: the raw input rows the tile touches (ConvFwdALoaderRow4's image:
// per segment of OW output pixels its four input rows of 2*OW columns, 8*BM floats per input channel = one chunk),
// taps applied on the fragment read:  lane (segment, ox), k = (ky, kx) reads  image[seg][ky][2*ox + kx - 1].
// OW is a template parameter so that the row step 2*OW sits in the ds_read's immediate offset (no VALU in the loop);
// the column left of the image (ox = 0, kx = 0) and right of it (ox = OW-1, kx = 3) are read from a zeroed region
// behind the image instead of being masked in registers.  Needs W = 2*OW, H = 2*OH, BM % OW == 0, 16-byte alignment.
template <int BM, int OWC>
struct ConvFwdA2 {
    static_assert(BM == 256 && BM % OWC == 0 && OWC >= 2, "piece mapping");
    using Params = typename ConvFwdALoader<BM, 4, 4, 2, 1>::Params;
    static constexpr int LD = BM, ROWS = BK / 2, PIECES = 2;
    static constexpr int OW_C = OWC;
    static constexpr int EXTRA = (6 * OWC + 4 + 3) & ~3;     // zeroed floats behind the image: every immediate row offset
                                                             // (up to 3 * 2*OW) of a redirected read stays inside
    static constexpr bool FWDROWS = true;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[2];
    int wave, HW, C;
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.x, (uint32_t)(s.N * s.C * s.H * s.W) * 4u);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        HW = s.H * s.W; C = s.C;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = ((wave * 2 + q) * 64 + lane) * 4;
            const int seg = f / (8 * OWC), rem = f - seg * 8 * OWC;
            const int ky = rem / (2 * OWC), col = rem - ky * 2 * OWC;
            const uint32_t m = (uint32_t)tile * BM + seg * OWC;           // first pixel of the segment
            const bool m_ok = m < (uint32_t)s.N * s.OH * s.OW;
            const uint32_t n = fdiv(m, p.div_ohw);
            const uint32_t oy = fdiv(m - n * (uint32_t)(s.OH * s.OW), p.div_ow);
            const int iy = (int)oy * 2 - 1 + ky;
            const bool ok = m_ok && (unsigned)iy < (unsigned)s.H;
            voff[q] = ok ? (n * (uint32_t)(s.C * HW) + (uint32_t)(iy * s.W + col)) * 4u : OOB;
        }
    }
    // float offsets (inside the A image) of the lane that owns tile pixel m_local, half-wave `half` (kx = 2*(s&1) +
    // half): even k-steps read `even`, odd ones `odd`, both plus (s >> 1) * 2*OW; ZERO = the zeroed region
    static constexpr int ZERO = ROWS * LD;
    __device__ __forceinline__ void frag(int m_local, int half, int& even, int& odd) const {
        const int seg = m_local / OWC, ox = m_local - seg * OWC;
        const int base = seg * 8 * OWC + 2 * ox - 1 + half;
        even = (half == 0 && ox == 0) ? ZERO : base;                  // kx = 0 left of the image
        odd = (half == 1 && ox == OWC - 1) ? ZERO : base + 2;         // kx = 3 right of it
    }
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {
        bload_lds16(rsrc, stage + (wave * 2 + p) * 256, voff[p], (live && kc < C) ? (uint32_t)kc * (uint32_t)HW * 4u : SOFF_OOB);
    }
};

// A operand of a forward convolution of ANY geometry, tap-major reduction (k = (tap, channel), channels padded to a
// chunk: the layout of ConvFwdALoaderTap and of pack_fwd_tap's weight rows): a chunk is 16 channels at ONE tap, its LDS
// image [16 channels][BM pixels] a plain gather -- pixel m of channel c sits at x[n][c][S*oy - P + dy][S*ox - P + dx],
// a fixed per-lane offset plus a per-(channel, tap) scalar.  The elements of a row are S floats apart in memory, so
// the pieces are 4-BYTE LDS-DMA instructions (64 pixels of one channel each; 16 per wavefront and chunk at BM = 256,
// three per k-step); taps that fall into the padding are out-of-range lanes (zeros), re-evaluated per lane only when
// the tap changes (once every C/16 chunks).  Fragment reads are igemm_kernel's plain [k][m] ones: no VALU in the loop.
template <int BM, int KH, int KW, int S, int P>
struct ConvTapA2 {
    static constexpr bool TAPGATHER = true;      // (igemm2_kg2_built: the two-wave-group form is instantiated)
    using Params = typename ConvFwdALoader<BM, KH, KW, S, P>::Params;
    static constexpr int LD = BM, ROWS = BK;
    static constexpr int G = BM / 64;                      // 64-pixel groups per LDS row
    static constexpr int PIECES = BK * G / 4;              // per wavefront and chunk: its 4 channel rows x G groups
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t vbase[G], veff[G];
    int iy0[G], ix0[G];
    int wave, C, H, W, HW, cblocks, last_tap, last_kc, cb;
    uint32_t tap_soff;
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        const uint32_t shift = (uint32_t)(P * s.W + P) * 4u;
        rsrc = make_rsrc(reinterpret_cast<const char*>(p.x) - shift, (uint32_t)s.N * s.C * s.H * s.W * 4u + shift);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        C = s.C; H = s.H; W = s.W; HW = s.H * s.W;
        cblocks = round_bk(s.C) / BK;
        last_tap = -1; last_kc = -1; cb = 0; tap_soff = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t m = (uint32_t)tile * BM + g * 64 + lane;
            const bool m_ok = m < (uint32_t)s.N * s.OH * s.OW;
            const uint32_t n = fdiv(m, p.div_ohw);
            const uint32_t pix = m - n * (uint32_t)(s.OH * s.OW);
            const uint32_t oy = fdiv(pix, p.div_ow);
            const uint32_t ox = pix - oy * (uint32_t)s.OW;
            iy0[g] = m_ok ? (int)oy * S - P : -(1 << 20);      // rows past M: every tap out of range
            ix0[g] = (int)ox * S - P;
            vbase[g] = (n * (uint32_t)(s.C * HW) + (uint32_t)((iy0[g] + P) * W + (ix0[g] + P))) * 4u;   // shifted base
            veff[g] = OOB;
        }
    }
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {
        if (p == 0) {        // (p is a literal at every call site) chunk -> (tap, channel block), wave-uniform;
                             // chunks arrive in increasing order, so only the first one costs a division
            int tap = last_tap;
            if (kc == last_kc + 1 && last_kc >= 0) {
                cb += BK;
                if (cb >= cblocks * BK) { cb = 0; ++tap; }
            } else {
                tap = kc / cblocks;
                cb = (kc - tap * cblocks) * BK;
            }
            last_kc = kc;
            if (tap != last_tap) {
                last_tap = tap;
                const int dy = tap / KW, dx = tap - dy * KW;
                tap_soff = (uint32_t)(dy * W + dx) * 4u;
#pragma unroll
                for (int g = 0; g < G; ++g)
                    veff[g] = ((unsigned)(iy0[g] + dy) < (unsigned)H && (unsigned)(ix0[g] + dx) < (unsigned)W) ? vbase[g] : OOB;
            }
        }
        const int row = wave * 4 + p / G, g = p % G;
        const int c = cb + row;
        bload_lds4(rsrc, stage + row * LD + g * 64, veff[g],
                   (live && c < C) ? (uint32_t)c * (uint32_t)HW * 4u + tap_soff : SOFF_OOB);
    }
};

// ConvTapA2 with the geometry at RUN time (round 6: the evaluation path's InceptionV3 -- 1x7 / 7x1 / 3x3 / 5x5 / 1x1,
// strides 1 and 2, asymmetric padding; reference core/submodules/gan_stability/metrics/inception.py): the same chunk =
// (tap, 16 channels) gather, the same pieces; only the per-tap validity and scalar offset use run-time KW / strides /
// paddings -- evaluated once per tap, not per chunk.
template <int BM>
struct ConvTapAnyA2 {
    static constexpr bool TAPGATHER = true;
    using Params = typename ConvFwdALoaderTapAny<BM>::Params;
    static constexpr int LD = BM, ROWS = BK;
    static constexpr int G = BM / 64;                      // 64-pixel groups per LDS row
    static constexpr int PIECES = BK * G / 4;              // per wavefront and chunk: its 4 channel rows x G groups
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t vbase[G], veff[G];
    int iy0[G], ix0[G];
    int wave, C, H, W, HW, KW, cblocks, last_tap, last_kc, cb;
    uint32_t tap_soff;
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        const uint32_t shift = (uint32_t)(p.PH * s.W + p.PW) * 4u;
        rsrc = make_rsrc(reinterpret_cast<const char*>(p.x) - shift, (uint32_t)s.N * s.C * s.H * s.W * 4u + shift);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        C = s.C; H = s.H; W = s.W; HW = s.H * s.W; KW = p.KW;
        cblocks = round_bk(s.C) / BK;
        last_tap = -1; last_kc = -1; cb = 0; tap_soff = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t m = (uint32_t)tile * BM + g * 64 + lane;
            const bool m_ok = m < (uint32_t)s.N * s.OH * s.OW;
            const uint32_t n = fdiv(m, p.div_ohw);
            const uint32_t pix = m - n * (uint32_t)(s.OH * s.OW);
            const uint32_t oy = fdiv(pix, p.div_ow);
            const uint32_t ox = pix - oy * (uint32_t)s.OW;
            iy0[g] = m_ok ? (int)oy * p.SH - p.PH : -(1 << 20);      // rows past M: every tap out of range
            ix0[g] = (int)ox * p.SW - p.PW;
            vbase[g] = (n * (uint32_t)(s.C * HW) + (uint32_t)((iy0[g] + p.PH) * W + (ix0[g] + p.PW))) * 4u;   // shifted base
            veff[g] = OOB;
        }
    }
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {
        if (p == 0) {
            int tap = last_tap;
            if (kc == last_kc + 1 && last_kc >= 0) {
                cb += BK;
                if (cb >= cblocks * BK) { cb = 0; ++tap; }
            } else {
                tap = kc / cblocks;
                cb = (kc - tap * cblocks) * BK;
            }
            last_kc = kc;
            if (tap != last_tap) {
                last_tap = tap;
                const int dy = tap / KW, dx = tap - dy * KW;
                tap_soff = (uint32_t)(dy * W + dx) * 4u;
#pragma unroll
                for (int g = 0; g < G; ++g)
                    veff[g] = ((unsigned)(iy0[g] + dy) < (unsigned)H && (unsigned)(ix0[g] + dx) < (unsigned)W) ? vbase[g] : OOB;
            }
        }
        const int row = wave * 4 + p / G, g = p % G;
        const int c = cb + row;
        bload_lds4(rsrc, stage + row * LD + g * 64, veff[g],
                   (live && c < C) ? (uint32_t)c * (uint32_t)HW * 4u + tap_soff : SOFF_OOB);
    }
};

// The same gather for the TRANSPOSED convolution, phase (py, px), tap-major (ConvDgALoaderTap's reduction order and
// pack_dgrad_tap's weight rows): chunk = 16 feature channels at one of the phase's ny x nx taps,
// A[k = (tap, ko)][m = (n, a, b)] = y[n][ko][oy0 - ty][ox0 - tx].  Phases have their own chunk counts.
template <int BM, int KH, int KW, int S, int P>
struct ConvDgTapA2 {
    static constexpr bool TAPGATHER = true;      // (igemm2_kg2_built: the two-wave-group form is instantiated)
    static constexpr int TY = (KH + S - 1) / S, TX = (KW + S - 1) / S;
    using Params = typename ConvDgALoaderTap<BM, KH, KW, S, P>::Params;
    static constexpr int LD = BM, ROWS = BK;
    static constexpr int G = BM / 64;
    static constexpr int PIECES = BK * G / 4;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t vbase[G], veff[G];
    int oy0[G], ox0[G];
    int wave, K, OH, OW, OHW, kblocks, nx_p, last_tap, last_kc, kob;
    uint32_t tap_soff;
    __device__ __forceinline__ void init(const Params& p, int tile, int phase, int tid) {
        const ConvShape& s = p.s;
        const uint32_t shift = (uint32_t)((TY - 1) * s.OW + (TX - 1)) * 4u;
        rsrc = make_rsrc(reinterpret_cast<const char*>(p.y) - shift, (uint32_t)s.N * s.K * s.OH * s.OW * 4u + shift);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int py = phase / S, px = phase % S;
        nx_p = dg_taps(KW, S, P, px);
        K = s.K; OH = s.OH; OW = s.OW; OHW = s.OH * s.OW;
        kblocks = round_bk(s.K) / BK;
        last_tap = -1; last_kc = -1; kob = 0; tap_soff = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t m = (uint32_t)tile * BM + g * 64 + lane;
            const bool m_ok = m < (uint32_t)s.N * p.AH * p.AW;
            const uint32_t n = fdiv(m, p.div_ahw);
            const uint32_t pix = m - n * (uint32_t)(p.AH * p.AW);
            const uint32_t a = fdiv(pix, p.div_aw);
            const uint32_t b = pix - a * (uint32_t)p.AW;
            oy0[g] = m_ok ? (int)a + (py + P) / S : -(1 << 20);
            ox0[g] = (int)b + (px + P) / S;
            // addresses (oy0 - (TY-1), ox0 - (TX-1)) through the shifted base; the tap's scalar offset walks forward
            vbase[g] = (n * (uint32_t)(s.K * OHW) + (uint32_t)(((int)a + (py + P) / S) * OW + ox0[g])) * 4u;
            veff[g] = OOB;
        }
    }
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {
        if (p == 0) {
            int tap = last_tap;
            if (kc == last_kc + 1 && last_kc >= 0) {
                kob += BK;
                if (kob >= kblocks * BK) { kob = 0; ++tap; }
            } else {
                tap = kc / kblocks;
                kob = (kc - tap * kblocks) * BK;
            }
            last_kc = kc;
            if (tap != last_tap) {
                last_tap = tap;
                const int ty = tap / nx_p, tx = tap - ty * nx_p;
                tap_soff = (uint32_t)((TY - 1 - ty) * OW + (TX - 1 - tx)) * 4u;
#pragma unroll
                for (int g = 0; g < G; ++g)
                    veff[g] = ((unsigned)(oy0[g] - ty) < (unsigned)OH && (unsigned)(ox0[g] - tx) < (unsigned)OW) ? vbase[g] : OOB;
            }
        }
        const int row = wave * 4 + p / G, g = p % G;
        const int ko = kob + row;
        bload_lds4(rsrc, stage + row * LD + g * 64, veff[g],
                   (live && ko < K) ? (uint32_t)ko * (uint32_t)OHW * 4u + tap_soff : SOFF_OOB);
    }
};

// ---- round 6: the 5x5 s2 p2 TRANSPOSED convolution, row-shared (HoloGAN's critic, input gradients) -------------------
// ConvDgTapA2 above gathers a chunk per (tap, 16 channels): the 9 / 6 / 6 / 4 taps of the four output phases each fetch
// the SAME gradient rows again, 4 bytes per lane and instruction, and every phase stores every other float of a row
// (EXT-128 D.block1: 6.6x its algorithmic bytes).  But along m = (n, a, b) the operand IS contiguous in memory --
//   dx[2a + py][2b + px] = sum_{ty < ny, tx < nx}  dy[a + 1 - ty][b + 1 - tx] . w[py + 2 ty][px + 2 tx],  ny = 3 - py, nx = 3 - px
// -- so the row-shared form of ConvDgA2 applies: an LDS row = 256 consecutive pixels of ONE (feature channel, ty),
// fetched ONCE by a 16-byte LDS-DMA piece, the horizontal taps applied as shifts on the fragment read (zero column for
// the lanes at an image row's edge).  What is new against k4 s2 p1:
//   * ALL FOUR output phases in one workgroup: tile = 256 pixels x 32 channels x (py, px), four wavefronts of 64 pixels
//     (TM = 2), accumulator block j = 2 py + px.  Row a + 1 - ty serves py = 0 as its tap ty and, for ty < 2, py = 1 as ITS
//     tap ty, and the column phases share the shifts (+1, 0): the A fragments are shared outright, the B image is the
//     ordinary 128-column one with columns = (phase, channel), and the epilogue stores a lane's four values as two
//     8-byte pairs (EpiPhaseQuadB: whole 64-byte lines leave the CU);
//   * TWELVE k-steps per chunk of 8 LDS rows: 8 row-shared ones (half-wave = shift +1 / 0) and then 4 PLAIN ones (half-wave
//     = next row, shift -1: px = 0's third tap) on the SAME staged rows -- every dy row is staged once;
//   * two chunk MODES, one loop each (a mode switch inside one loop made the register allocator spill accumulators):
//     rows ty < 2 feed all four phases (8 x 8 + 4 x 4 = 80 MFMAs per wavefront and chunk), rows ty = 2 the py = 0 phases
//     (40).  3 K/8 chunks, 25 K MFMAs per wavefront = exactly the 25 taps, and every workgroup is identical (two earlier
//     forms of the round were not: DESIGN 3.1);
//   * the weights are read from the EXISTING tap-major pack (pack_dgrad_tap: [phase (py, px)][tap = ty * nx + tx][ko][C])
//     by a B loader that picks the rows (DgQuadB2): no second packed image, no pack-cache changes.
// Needs AH = OH, AW = OW a multiple of 4 dividing 256, K % 16 == 0, C % 32 == 0, 16-byte aligned tensors, no bias /
// activation (an input gradient).  Reference: core/models/hologan_discriminator.py:7-23 (Conv2d k5 s2 p2).
template <int BM>
struct ConvDg5A2 {
    static_assert(BM == 256, "one 256-pixel piece per LDS row");
    using Params = typename ConvDgALoaderTap<BM, 5, 5, 2, 2>::Params;
    static constexpr int LD = BM + 4, ROWS = 8, PIECES = 2;
    static constexpr int ZERO_COL = BM;
    static constexpr bool ROWSHARE = true, DUALMODE = true;
    static constexpr int MODES = 2, KSTEPS = 12;
    static constexpr bool step_plain(int m, int S) { return S >= 8; }
    static constexpr unsigned step_mask(int m, int S) {       // accumulator blocks (output phases) the k-step's taps feed
        return m == 0 ? (S < 8 ? 15u : 5u) : (S < 8 ? 3u : 1u);
    }
    static constexpr int step_arow(int m, int S) { return S < 8 ? S : 2 * (S - 8); }
    static constexpr int step_brow(int m, int S) { return S < 8 ? 2 * S : 16 + 2 * (S - 8); }
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff0, voff1, voff2;      // [ty]: this lane's pixel quad in row a + 1 - ty (out of range: outside the image)
    int wave, K, OHW;                  // (three scalars, not an array: a ty-indexed array ends up in scratch)
    int bound[1];                      // first chunk of mode 1 (rows ty = 2)
    __device__ __forceinline__ int mode_of(int kc) const { return kc >= bound[0]; }
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.y, (uint32_t)s.N * s.K * s.OH * s.OW * 4u);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        K = s.K; OHW = s.OH * s.OW;
        bound[0] = 2 * K / 8;
        const uint32_t m = (uint32_t)tile * BM + lane * 4;
        const bool m_ok = m < (uint32_t)s.N * p.AH * p.AW;
        const uint32_t n = fdiv(m, p.div_ahw);
        const uint32_t pix = m - n * (uint32_t)(p.AH * p.AW);
        const uint32_t a = fdiv(pix, p.div_aw);
        const uint32_t b = pix - a * (uint32_t)p.AW;
        auto row = [&](int ty) {
            const int oy = (int)a + 1 - ty;
            const bool ok = m_ok && (unsigned)oy < (unsigned)s.OH;
            return ok ? (n * (uint32_t)(s.K * OHW) + (uint32_t)(oy * s.OW) + b) * 4u : OOB;
        };
        voff0 = row(0); voff1 = row(1); voff2 = row(2);
    }
    __device__ __forceinline__ int frag_shift(int half) const { return half ? 0 : 1; }      // row-shared k-steps: tx = half-wave
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {
        const int row = wave * 2 + p;
        const int r = kc * 8 + row;          // (ty, ko), ty-major: r = ty * K + ko
        const int ty = (r >= K) + (r >= 2 * K);
        const int ko = r - ty * K;
        // (masks, not a select chain: the compiler turns `ty == 0 ? voff0 : ...` into a ty-indexed table in scratch)
        const uint32_t m0 = 0u - (uint32_t)(ty == 0), m1 = 0u - (uint32_t)(ty == 1), m2 = 0u - (uint32_t)(ty == 2);
        const uint32_t v = (voff0 & m0) | (voff1 & m1) | (voff2 & m2);
        bload_lds16(rsrc, stage + row * LD, v, live ? (uint32_t)ko * (uint32_t)OHW * 4u : SOFF_OOB);
    }
};

// B operand of ConvDg5A2: image [24 k rows][128 columns = (phase j = 2 py + px, 32 channels)] per chunk of 8 LDS rows
// r = (ty, ko), gathered row-wise out of the tap-major dgrad pack of a 5x5 s2 p2 weight (gz_conv.hip pack_dgrad_tap_body:
// phase (py, px) at phase * 9 * K * ldc floats, row (ty * nx + tx) * K + ko, nx = 3 - px): rows 0-15 = (r, tx in {0, 1}) of
// every phase that has tap row ty (py = 1 has none at ty = 2: zeros, never multiplied), rows 16-23 = (r, tx = 2) of the
// px = 0 phases.  A piece = two image rows x 128 columns (lane = (image row, 4 columns)); the lane's phase and tap offsets
// are per-lane constants, ty enters through a per-lane stride (nx differs between the column phases).
struct DgQuadB2 {
    struct Params {
        const float* wp;
        int K, C, ldc;                  // feature channels (multiple of 16), image-side channels, row pitch of the pack
    };
    static constexpr int LD = 128, ROWS = 24, PIECES = 3;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t vA, vA_step, vA2, vB, vB2, ldb;
    int wave, K, b0;
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const uint32_t phase_floats = 9u * (uint32_t)p.K * (uint32_t)p.ldc;
        rsrc = make_rsrc(p.wp, 4u * phase_floats * 4u);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        K = p.K;
        b0 = 2 * K / 8;
        const int t = lane >> 5, q = (lane & 31) * 4;             // image row inside the piece, first of 4 image columns
        const int j = q >> 5, py = j >> 1, px = j & 1;
        const int ch = tile * 32 + (q & 31);
        const bool ok = ch < p.C;
        ldb = (uint32_t)p.ldc * 4u;
        const uint32_t phase = (uint32_t)j * phase_floats;
        const uint32_t kld = (uint32_t)(p.K * p.ldc);
        vA = ok ? (phase + (uint32_t)t * kld + (uint32_t)ch) * 4u : OOB;                 // + ty * vA_step, + ko * ldb
        vA_step = ok ? (uint32_t)(3 - px) * kld * 4u : 0u;
        vA2 = (ok && py == 0) ? vA + 2u * vA_step : OOB;                                    // ty = 2
        vB = (ok && px == 0) ? (phase + (uint32_t)t * (uint32_t)p.ldc + (uint32_t)ch) * 4u : OOB;   // + ((3 ty + 2) K + ko) * ldb
        vB2 = j == 0 ? vB : OOB;                                                             // ty = 2: phase (0, 0) alone
    }
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {
        const int piece = wave * PIECES + p;                      // 0..11: image rows 2 piece, 2 piece + 1
        const int r0 = kc * 8;                                    // the chunk's first LDS row (all of one ty: K % 8 == 0)
        const int ty = (r0 >= K) + (r0 >= 2 * K);
        uint32_t v, so;
        if (piece < 8) {                     // (LDS row r0 + piece, tx = lane's image row)
            v = kc < b0 ? vA + (uint32_t)ty * vA_step : vA2;
            so = (uint32_t)(r0 + piece - ty * K) * ldb;
        } else {                             // (LDS rows r0 + 2 (piece - 8) + lane's image row, tx = 2)
            v = kc < b0 ? vB : vB2;
            so = (uint32_t)((ty * 3 + 2) * K + (r0 + 2 * (piece - 8) - ty * K)) * ldb;
        }
        bload_lds16(rsrc, stage + piece * 256, v, live ? so : SOFF_OOB);
    }
};

// 1x1 stride-1 layers are plain GEMMs over an NCHW tensor: A[k = channel][m = (n, pixel)], 256 consecutive rows of a
// channel are 1 KB of contiguous memory whenever H*W is a multiple of 4 (a lane's 16-byte quad never straddles two
// samples).  One 16-byte LDS-DMA piece per LDS row instead of the gather's four 4-byte ones (HoloGAN's 1024 -> 1024
// projection, core/models/hologan_generator.py:130: 114 -> see DESIGN.md).
template <int BM>
struct PlaneA2 {
    static_assert(BM % 256 == 0, "whole 256-pixel pieces per LDS row");
    struct Params {
        const float* base;
        int CH, HW, M;
        FastDiv div_hw;
    };
    static constexpr int LD = BM, ROWS = BK;
    static constexpr int PPR = BM / 256;
    static constexpr int PIECES = BK * PPR / 4;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[PPR], plane;
    int wave, CH;
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        rsrc = make_rsrc(p.base, (uint32_t)(p.M / p.HW) * (uint32_t)p.CH * (uint32_t)p.HW * 4u);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        CH = p.CH;
        plane = (uint32_t)p.HW * 4u;
#pragma unroll
        for (int g = 0; g < PPR; ++g) {
            const uint32_t m = (uint32_t)tile * BM + g * 256 + 4 * lane;
            const uint32_t n = fdiv(m, p.div_hw);
            voff[g] = m < (uint32_t)p.M ? (n * (uint32_t)(p.CH * p.HW) + (m - n * (uint32_t)p.HW)) * 4u : OOB;
        }
    }
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {
        const int row = wave * 4 + p / PPR, g = p % PPR;
        const int c = kc * BK + row;
        bload_lds16(rsrc, stage + row * LD + g * 256, voff[g], (live && c < CH) ? (uint32_t)c * plane : SOFF_OOB);
    }
};

// The two gathers in three dimensions (HoloGAN's ConvTranspose3d k3 s2 p1 op1, core/models/hologan_generator.py:29-30:
// its forward is the transposed form with 8 phases of 1..8 taps, its input gradient the plain strided convolution).
// Tap-major: chunk = 16 channels at one tap; the weight rows follow pack_fwd3_tap / pack_dgrad3_tap (gz_conv3d.hip).
template <int BM, int KS, int S, int P>
struct Conv3DTapA2 {
    using Params = typename Conv3DFwdALoader<BM, KS, S, P>::Params;
    static constexpr int LD = BM, ROWS = BK;
    static constexpr int G = BM / 64;
    static constexpr int PIECES = BK * G / 4;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t vbase[G], veff[G];
    int id0[G], iy0[G], ix0[G];
    int wave, C, D, H, W, DHW, cblocks, last_tap, last_kc, cb;
    uint32_t tap_soff;
    __device__ __forceinline__ void init(const Params& p, int tile, int y, int tid) {
        const Conv3DShape& s = p.s;
        const uint32_t shift = (uint32_t)((P * s.H + P) * s.W + P) * 4u;
        rsrc = make_rsrc(reinterpret_cast<const char*>(p.x) - shift,
                         (uint32_t)s.N * s.C * s.D * s.H * s.W * 4u + shift);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        C = s.C; D = s.D; H = s.H; W = s.W; DHW = s.D * s.H * s.W;
        cblocks = round_bk(s.C) / BK;
        last_tap = -1; last_kc = -1; cb = 0; tap_soff = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t m = (uint32_t)tile * BM + g * 64 + lane;
            const bool m_ok = m < (uint32_t)s.N * s.OD * s.OH * s.OW;
            const uint32_t n = fdiv(m, p.div_odhw);
            uint32_t v = m - n * (uint32_t)(s.OD * s.OH * s.OW);
            const uint32_t od = fdiv(v, p.div_ohw);
            v -= od * (uint32_t)(s.OH * s.OW);
            const uint32_t oy = fdiv(v, p.div_ow);
            const uint32_t ox = v - oy * (uint32_t)s.OW;
            id0[g] = m_ok ? (int)od * S - P : -(1 << 20);      // rows past M: every tap out of range
            iy0[g] = (int)oy * S - P;
            ix0[g] = (int)ox * S - P;
            vbase[g] = (n * (uint32_t)(s.C * DHW) +
                        (uint32_t)((((int)od * S) * H + (iy0[g] + P)) * W + (ix0[g] + P))) * 4u;   // shifted base
            veff[g] = OOB;
        }
    }
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {
        if (p == 0) {
            int tap = last_tap;
            if (kc == last_kc + 1 && last_kc >= 0) {
                cb += BK;
                if (cb >= cblocks * BK) { cb = 0; ++tap; }
            } else {
                tap = kc / cblocks;
                cb = (kc - tap * cblocks) * BK;
            }
            last_kc = kc;
            if (tap != last_tap) {
                last_tap = tap;
                const int kd = tap / (KS * KS), r = tap - kd * (KS * KS), ky = r / KS, kx = r - ky * KS;
                tap_soff = (uint32_t)((kd * H + ky) * W + kx) * 4u;
#pragma unroll
                for (int g = 0; g < G; ++g)
                    veff[g] = ((unsigned)(id0[g] + kd) < (unsigned)D && (unsigned)(iy0[g] + ky) < (unsigned)H &&
                               (unsigned)(ix0[g] + kx) < (unsigned)W) ? vbase[g] : OOB;
            }
        }
        const int row = wave * 4 + p / G, g = p % G;
        const int c = cb + row;
        bload_lds4(rsrc, stage + row * LD + g * 64, veff[g],
                   (live && c < C) ? (uint32_t)c * (uint32_t)DHW * 4u + tap_soff : SOFF_OOB);
    }
};

// A[k = (tap, ko)][m = (n, a, b, c)] = y[n][ko][od0 - td][oy0 - ty][ox0 - tx] of phase (pd, py, px); a phase has its own
// nd x ny x nx taps (tap = (td * ny + ty) * nx + tx) and chunk count.
template <int BM, int KS, int S, int P>
struct Conv3DDgTapA2 {
    static constexpr int T = (KS + S - 1) / S;
    using Params = typename Conv3DDgALoader<BM, KS, S, P>::Params;
    static constexpr int LD = BM, ROWS = BK;
    static constexpr int G = BM / 64;
    static constexpr int PIECES = BK * G / 4;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t vbase[G], veff[G];
    int od0[G], oy0[G], ox0[G];
    int wave, K, OD, OH, OW, OSP, kblocks, ny_p, nx_p, last_tap, last_kc, kob;
    uint32_t tap_soff;
    __device__ __forceinline__ void init(const Params& p, int tile, int phase, int tid) {
        const Conv3DShape& s = p.s;
        const uint32_t shift = (uint32_t)(((T - 1) * s.OH + (T - 1)) * s.OW + (T - 1)) * 4u;
        rsrc = make_rsrc(reinterpret_cast<const char*>(p.y) - shift,
                         (uint32_t)s.N * s.K * s.OD * s.OH * s.OW * 4u + shift);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int pd = phase / (S * S), py = (phase / S) % S, px = phase % S;
        ny_p = dg_taps(KS, S, P, py);
        nx_p = dg_taps(KS, S, P, px);
        K = s.K; OD = s.OD; OH = s.OH; OW = s.OW; OSP = s.OD * s.OH * s.OW;
        kblocks = round_bk(s.K) / BK;
        last_tap = -1; last_kc = -1; kob = 0; tap_soff = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const uint32_t m = (uint32_t)tile * BM + g * 64 + lane;
            const bool m_ok = m < (uint32_t)s.N * p.AD * p.AH * p.AW;
            const uint32_t n = fdiv(m, p.div_adhw);
            uint32_t v = m - n * (uint32_t)(p.AD * p.AH * p.AW);
            const uint32_t a = fdiv(v, p.div_ahw);
            v -= a * (uint32_t)(p.AH * p.AW);
            const uint32_t b = fdiv(v, p.div_aw);
            const uint32_t c = v - b * (uint32_t)p.AW;
            const int od = (int)a + (pd + P) / S;
            od0[g] = m_ok ? od : -(1 << 20);
            oy0[g] = (int)b + (py + P) / S;
            ox0[g] = (int)c + (px + P) / S;
            // addresses (od - (T-1), oy0 - (T-1), ox0 - (T-1)) through the shifted base; the tap's scalar offset walks forward
            vbase[g] = (n * (uint32_t)(s.K * OSP) + (uint32_t)((od * OH + oy0[g]) * OW + ox0[g])) * 4u;
            veff[g] = OOB;
        }
    }
    __device__ __forceinline__ void issue_piece(int kc, float* stage, int p, bool live) {
        if (p == 0) {
            int tap = last_tap;
            if (kc == last_kc + 1 && last_kc >= 0) {
                kob += BK;
                if (kob >= kblocks * BK) { kob = 0; ++tap; }
            } else {
                tap = kc / kblocks;
                kob = (kc - tap * kblocks) * BK;
            }
            last_kc = kc;
            if (tap != last_tap) {
                last_tap = tap;
                const int td = tap / (ny_p * nx_p), r = tap - td * (ny_p * nx_p), ty = r / nx_p, tx = r - ty * nx_p;
                tap_soff = (uint32_t)(((T - 1 - td) * OH + (T - 1 - ty)) * OW + (T - 1 - tx)) * 4u;
#pragma unroll
                for (int g = 0; g < G; ++g)
                    veff[g] = ((unsigned)(od0[g] - td) < (unsigned)OD && (unsigned)(oy0[g] - ty) < (unsigned)OH &&
                               (unsigned)(ox0[g] - tx) < (unsigned)OW) ? vbase[g] : OOB;
            }
        }
        const int row = wave * 4 + p / G, g = p % G;
        const int ko = kob + row;
        bload_lds4(rsrc, stage + row * LD + g * 64, veff[g],
                   (live && ko < K) ? (uint32_t)ko * (uint32_t)OSP * 4u + tap_soff : SOFF_OOB);
    }
};

#ifdef GZ2_STAMPS       // diagnostic builds only (tools/igemm2_conv_probe.hip, tools/conv_bench2.py --stamps)
__device__ unsigned long long gz2_stamps[8192 * 8];
#define GZ2_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define GZ2_STAMP(var)
#endif

// ---- hand-ordered instruction stream of one k-step ---------------------------------------------------------------
// hipcc's scheduler is not latency-aware here: it puts the v_cndmask that zeroes a fragment right behind the ds_read
// that fetches it (sched_group_barrier / sched_barrier variants all ended with an exposed LDS latency per k-step:
// 78-90 % of the MFMA-issue bound for a wavefront that has its SIMD to itself).  So the stream is written out:
//   ds_read (next k-step's fragments) ... the TM*TN MFMAs of this k-step ... s_waitcnt lgkmcnt(0) (in the shadow of
//   the last MFMA: the reads are ~1000 cycles old) ... masks of the next k-step.
// Every piece is an `asm volatile`; the wait statement takes the freshly read registers as in/out operands, so no use
// of them can be scheduled above it, and the compiler's own code (LDS-DMA issue, loop control) can only fall between
// pieces.  Accumulators live in AGPRs ("+a").
// (no wait states needed in front of the MFMAs: their A / B registers are written by ds_read only -- the zero column
// replaced the v_cndmask masks -- and s_waitcnt covers that)
#ifndef GZ2_EXP_NOP
#define GZ2_NOP ""
#else
#define GZ2_NOP "s_nop 1\n\t"
#endif
template <int OFF>
__device__ __forceinline__ float lds_rd(uint32_t byte_addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void mfma_row(f32x16 (&c)[4], float a, const float (&b)[4]) {
    // s_nop 1: a VALU write (the zeroing v_cndmask) needs two wait states before an MFMA reads the register, and the
    // hazard recognizer does not look inside asm statements
    asm volatile(GZ2_NOP
                 "v_mfma_f32_32x32x2_f32 %0, %5, %4, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %1, %6, %4, %1\n\t"
                 "v_mfma_f32_32x32x2_f32 %2, %7, %4, %2\n\t"
                 "v_mfma_f32_32x32x2_f32 %3, %8, %4, %3"
                 : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3])
                 : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
__device__ __forceinline__ void mfma_row(f32x16 (&c)[2], float a, const float (&b)[2]) {
    asm volatile(GZ2_NOP
                 "v_mfma_f32_32x32x2_f32 %0, %3, %2, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %1, %4, %2, %1"
                 : "+a"(c[0]), "+a"(c[1])
                 : "v"(a), "v"(b[0]), "v"(b[1]));
}
__device__ __forceinline__ void mfma_row(f32x16 (&c)[1], float a, const float (&b)[1]) {
    asm volatile(GZ2_NOP
                 "v_mfma_f32_32x32x2_f32 %0, %2, %1, %0"
                 : "+a"(c[0])
                 : "v"(a), "v"(b[0]));
}
__device__ __forceinline__ void mfma_one(f32x16& c, float a, float b) {      // dual mode, part B: accumulator block 0 only
    asm volatile(GZ2_NOP
                 "v_mfma_f32_32x32x2_f32 %0, %2, %1, %0"
                 : "+a"(c)
                 : "v"(a), "v"(b));
}
// D[m][n] order (lanes along n): row-major outputs (weight gradient slabs)
__device__ __forceinline__ void mfma_row_mn(f32x16 (&c)[4], float a, const float (&b)[4]) {
    asm volatile(GZ2_NOP
                 "v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %1, %4, %6, %1\n\t"
                 "v_mfma_f32_32x32x2_f32 %2, %4, %7, %2\n\t"
                 "v_mfma_f32_32x32x2_f32 %3, %4, %8, %3"
                 : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3])
                 : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
__device__ __forceinline__ void mfma_row_mn(f32x16 (&c)[2], float a, const float (&b)[2]) {
    asm volatile(GZ2_NOP
                 "v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\t"
                 "v_mfma_f32_32x32x2_f32 %1, %2, %4, %1"
                 : "+a"(c[0]), "+a"(c[1])
                 : "v"(a), "v"(b[0]), "v"(b[1]));
}
__device__ __forceinline__ void lgkm_done(float (&a)[4], float (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ void lgkm_done(float (&a)[4], float (&b)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]));
}
__device__ __forceinline__ void lgkm_done(float (&a)[2], float (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ void lgkm_done(float (&a)[4], float (&b)[1]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]));
}

template <class AL, class = void>
struct a_extra_of { static constexpr int value = 0; };
template <class AL>
struct a_extra_of<AL, std::void_t<decltype(AL::EXTRA)>> { static constexpr int value = AL::EXTRA; };
template <class AL>
constexpr int igemm2_a_extra() { return a_extra_of<AL>::value; }
template <class AL, class = void>
struct fwd_ow_of { static constexpr int value = 1; };
template <class AL>
struct fwd_ow_of<AL, std::void_t<decltype(AL::OW_C)>> { static constexpr int value = AL::OW_C; };
template <class AL>
constexpr int fwd_ow() { return fwd_ow_of<AL>::value; }

// multi-mode loaders (ConvDg5A2): traits with defaults for every other loader
template <class AL>
constexpr int mode_count_of() {
    if constexpr (is_dualmode<AL>::value) return AL::MODES;
    else return 1;
}
template <class AL>
constexpr bool mode_plain_of(int m, int S) {       // k-step S of a mode-m chunk: plain (half-wave = next LDS row) or row-shared
    if constexpr (is_dualmode<AL>::value) return AL::step_plain(m, S);
    else return false;
}
template <class AL>
constexpr unsigned mode_mask_of(int m, int S) {    // ... and the accumulator blocks its MFMAs feed
    if constexpr (is_dualmode<AL>::value) return AL::step_mask(m, S);
    else return 0xFu;
}
template <class AL>
constexpr int ksteps_of() {                        // k-steps per chunk (the folded four-phase form: 8 row-shared + 4 plain)
    if constexpr (is_dualmode<AL>::value) return AL::KSTEPS;
    else return BK / 2;
}
template <class AL>
constexpr int step_arow_of(int m, int S) {         // first LDS row of k-step S (plain steps: two rows per step)
    if constexpr (is_dualmode<AL>::value) return AL::step_arow(m, S);
    else return S;
}
template <class AL>
constexpr int step_brow_of(int m, int S) {         // first row of the B image k-step S multiplies
    if constexpr (is_dualmode<AL>::value) return AL::step_brow(m, S);
    else return 2 * S;
}

// KG = 2 (round 5): TWO wave groups of four wavefronts in one 512-thread workgroup work on the SAME output tile, each on
// half of the workgroup's reduction range with an LDS ring of its own, and meet in LDS at the end (group 1 parks its
// accumulators, group 0 adds them and runs the epilogue).  For launches whose tiles x reduction splits give a CU only
// one workgroup: the CU then still holds two wavefronts per SIMD -- the second one is what hides the non-MFMA
// instructions of the k-step -- without the slabs + finish pass that a global split of the reduction costs.
template <class Cfg, class AL, class BL, int KG>
constexpr int igemm2_ring_floats() {
    return ((STAGES2 * (AL::ROWS * AL::LD + igemm2_a_extra<AL>() + BL::ROWS * BL::LD) + 63) / 64) * 64;
}

template <class Cfg, class AL, class BL, class Epi, int KG = 1>
__global__ __launch_bounds__(NT * KG, KG == 1 ? Cfg::OCC : 2) void igemm2_kernel(typename AL::Params pa, typename BL::Params pb,
                                                                              typename Epi::Params pe, GridMap gm) {
    constexpr int LDA = AL::LD, LDB = BL::LD;
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    static_assert((TM == 4 || (TM == 2 && is_dualmode<AL>::value)) && (TN == 1 || TN == 2 || TN == 4),
                  "fragment registers of the hand-ordered k-step");
    static_assert(KG == 1 || (KG == 2 && TN <= 2), "two wave groups: the parked accumulators must fit the LDS");
    constexpr bool RS = is_rowshare<AL>::value;
    constexpr bool FR = is_fwdrows<AL>::value;
    // (ConvDg5A2) MODES consecutive chunk ranges, each with its own k-step form: row-shared or plain fragment addressing
    // and the accumulator blocks (output phases) its MFMAs feed -- AL::mode_plain(m), AL::mode_mask(m), al.bound[]
    constexpr bool DM = is_dualmode<AL>::value;
    static_assert(!DM || (RS && KG == 1 && (TN == 2 || TN == 4)), "multi-mode: row-shared loader, phases in the accumulator blocks");
    constexpr int A_EXTRA = igemm2_a_extra<AL>();
    constexpr int A_ELEMS = AL::ROWS * LDA + A_EXTRA, B_ELEMS = BL::ROWS * LDB;
    constexpr int STAGE = A_ELEMS + B_ELEMS;
    constexpr int PAD = 16;
    extern __shared__ __attribute__((aligned(16))) float smem2[];
    const int kg = KG == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);      // wave group
    float* const ring = smem2 + PAD + kg * igemm2_ring_floats<Cfg, AL, BL, KG>();   // stage i: [B image][A image]
    GZ2_STAMP(st0);
#ifdef GZ2_STAMPS
    const unsigned long long sr0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st1 = st0, st2 = st0;
#endif

    const int tid = KG == 1 ? (int)threadIdx.x : ((int)threadIdx.x & (NT - 1));     // inside the wave group
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    // (experiment, off by default -- see launch_igemm2)  The first-round workgroup in the CU's odd thread-group slot
    // starts a fraction of a tile late; the offset persists, because every later workgroup starts when its predecessor
    // in that slot ends.
    if (gm.stagger > 0 && bid < 512 && blockIdx.z == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (16 << 6) | 4);      // HW_ID.TG_ID
        if (hw & 1u) {
            const unsigned long long t_end = __builtin_amdgcn_s_memtime() + (unsigned long long)gm.stagger;
            while (__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(64);
        }
    }
    int y;
    if (gm.var_chunks) {      // phases of unequal length: longest first over all tiles (see igemm_kernel)
        const int tiles = gm.tiles_m * gm.tiles_n;
        y = gm.phase_order[bid / tiles];
        bid %= tiles;
    } else {
        if (!gm.no_swizzle) {
            const int q = nwg >> 3, rr = nwg & 7, x = bid & 7, i = bid >> 3;
            bid = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + i;
        }
        y = bid % gm.ny;
        bid /= gm.ny;
    }
    const int tile_n = bid % gm.tiles_n;
    const int tile_m = bid / gm.tiles_n;
    const int z = blockIdx.z;
    const int kcA = z * gm.chunks_per_split;                 // the workgroup's reduction range [kcA, kcB)
    const int kcB = min(gm.var_chunks ? gm.phase_chunks[y] : gm.chunks, kcA + gm.chunks_per_split);
    if (gm.slab && gm.var_chunks && kcA >= kcB) return;      // past this phase's last slab (uniform per workgroup)
    // this wave group's share: chunks [kc0, kc1) are live; the loop runs to kend in BOTH groups (same barrier count),
    // the chunks past kc1 arrive as zeros (out-of-range LDS-DMA) and add nothing
    int kc0 = kcA, kc1 = kcB, kend = kcB;
    if constexpr (KG == 2) {
        const int h = (kcB - kcA + 1) >> 1;
        kc0 = kcA + kg * h;
        kend = kc0 + h;
        kc1 = min(kcB, kend);
    }

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

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int half = lane >> 5, l32 = lane & 31;
    // byte addresses (LDS) of this lane's fragment columns inside stage 0
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)ring;
    uint32_t a_addr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
        a_addr[i] = lds0 + (uint32_t)(B_ELEMS + half * (RS ? 0 : LDA) + wm * TM * 32 + i * 32 + l32) * 4u;
    if constexpr (RS) {
        const int sh = al.frag_shift(half);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int b = (tile_m * Cfg::BM + wm * TM * 32 + i * 32 + l32) % pa.AW;
            const bool zero = (sh < 0 && b == 0) || (sh > 0 && b == pa.AW - 1);
            a_addr[i] = zero ? lds0 + (uint32_t)(B_ELEMS + AL::ZERO_COL) * 4u : a_addr[i] + (uint32_t)(sh * 4);
        }
        // the pad columns of every row of every stage
        if (tid < STAGES2 * AL::ROWS * 4) {
            const int st = tid / (AL::ROWS * 4), q = tid % (AL::ROWS * 4);
            ring[st * STAGE + B_ELEMS + (q >> 2) * LDA + AL::ZERO_COL + (q & 3)] = 0.f;
        }
    }
    uint32_t a_addrB[TM];          // dual mode, part B: half-wave = next LDS row, ONE shift (-1) for both
    if constexpr (DM) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int b = (tile_m * Cfg::BM + wm * TM * 32 + i * 32 + l32) % pa.AW;
            a_addrB[i] = lds0 + (uint32_t)(B_ELEMS + half * LDA + (b == 0 ? AL::ZERO_COL : wm * TM * 32 + i * 32 + l32 - 1)) * 4u;
        }
    }
    uint32_t a_odd[TM];            // forward-row images: the odd k-steps' addresses (a_addr: the even ones)
    if constexpr (FR) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int e, o;
            al.frag(wm * TM * 32 + i * 32 + l32, half, e, o);
            a_addr[i] = lds0 + (uint32_t)(B_ELEMS + e) * 4u;
            a_odd[i] = lds0 + (uint32_t)(B_ELEMS + o) * 4u;
        }
        for (int q = tid; q < STAGES2 * A_EXTRA; q += NT)
            ring[(q / A_EXTRA) * STAGE + B_ELEMS + AL::ZERO + q % A_EXTRA] = 0.f;
    }
    const uint32_t b_addr = lds0 + (uint32_t)(half * LDB + wn * TN * 32 + l32) * 4u;

    constexpr int NPA = AL::PIECES, NPB = BL::PIECES, NP = NPA + NPB, STEPS = ksteps_of<AL>();
    static_assert(STEPS == 8 || STEPS == 12, "k-steps per chunk");
    constexpr int PPS = (NP + STEPS - 1) / STEPS;          // LDS-DMA pieces per k-step (1 for the 16-byte loaders)
    static_assert(PPS <= 4 && NP < 64, "pieces are spread over the k-step's four MFMA rows; vmcnt is 6 bits");
    auto issue_piece = [&](int kc, int st, int p, bool live) {
#ifdef GZ2_EXP_NODMA       // timing experiments only (wrong results)
        if (kc >= kc0 + 2) return;
#endif
#ifdef GZ2_EXP_SAMECHUNK
        if (kc >= kc0 + 2) kc = kc0;
#endif
        float* sb = ring + st * STAGE;
        if (p < NPA) al.issue_piece(kc, sb + B_ELEMS, p, live);
        else bl.issue_piece(kc, sb, p - NPA, live);
    };
    // k-step S of the stage whose byte offset is `so`: raw fragments
    auto fetch_m = [&](auto Sc, auto Mc, uint32_t so, float (&af)[TM], float (&bf)[TN]) {      // multi-mode loaders
        constexpr int S = decltype(Sc)::value, MD = decltype(Mc)::value;
        constexpr bool PL = mode_plain_of<AL>(MD, S);
        constexpr unsigned MK = mode_mask_of<AL>(MD, S);
        constexpr int AO = LDA * step_arow_of<AL>(MD, S) * 4, BO = step_brow_of<AL>(MD, S) * LDB * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = lds_rd<AO>((PL ? a_addrB[i] : a_addr[i]) + so);
        if constexpr ((MK & 1u) != 0) bf[0] = lds_rd<BO>(b_addr + so);
        if constexpr (TN >= 2 && (MK & 2u) != 0) bf[1] = lds_rd<BO + 128>(b_addr + so);
        if constexpr (TN == 4 && (MK & 4u) != 0) bf[2] = lds_rd<BO + 256>(b_addr + so);
        if constexpr (TN == 4 && (MK & 8u) != 0) bf[3] = lds_rd<BO + 384>(b_addr + so);
    };
    // the first fragments of chunk `kc`, in that chunk's mode
    auto fetch_first = [&](int kc, uint32_t so, float (&af)[TM], float (&bf)[TN]) {
        if constexpr (DM) {
            if (al.mode_of(kc) == 0) fetch_m(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, so, af, bf);
            else fetch_m(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, so, af, bf);
        }
    };
    auto fetch = [&](auto Sc, uint32_t so, float (&af)[TM], float (&bf)[TN]) {
        constexpr int S = decltype(Sc)::value;
        constexpr int AO = (RS ? LDA : 2 * LDA) * S * 4, BO = 2 * S * LDB * 4;
#ifdef GZ2_EXP_NOFETCH
        if (so != 0xFFFFFFFFu) return;
#endif
        // immediate offsets only: no VALU address arithmetic inside the loop (`a_addr[i] + so` is per-chunk)
        if constexpr (FR) {
            constexpr int FO = (S >> 1) * 2 * fwd_ow<AL>() * 4;
            af[0] = lds_rd<FO>(((S & 1) ? a_odd[0] : a_addr[0]) + so);
            af[1] = lds_rd<FO>(((S & 1) ? a_odd[1] : a_addr[1]) + so);
            af[2] = lds_rd<FO>(((S & 1) ? a_odd[2] : a_addr[2]) + so);
            af[3] = lds_rd<FO>(((S & 1) ? a_odd[3] : a_addr[3]) + so);
        } else {
            af[0] = lds_rd<AO>(a_addr[0] + so);
            af[1] = lds_rd<AO>(a_addr[1] + so);
            af[2] = lds_rd<AO>(a_addr[2] + so);
            af[3] = lds_rd<AO>(a_addr[3] + so);
        }
        {
        bf[0] = lds_rd<BO>(b_addr + so);
        if constexpr (TN >= 2) bf[1] = lds_rd<BO + 128>(b_addr + so);
        if constexpr (TN == 4) {
            bf[2] = lds_rd<BO + 256>(b_addr + so);
            bf[3] = lds_rd<BO + 384>(b_addr + so);
        }
        }
    };
    auto mask = [&](float (&)[TM]) {};      // (row-shared images: the zero column replaces the register masks)

    if (kcA < kcB) {
#pragma unroll
        for (int p = 0; p < NP; ++p) issue_piece(kc0, 0, p, kc0 < kc1);
#pragma unroll
        for (int p = 0; p < NP; ++p) issue_piece(kc0 + 1, 1, p, kc0 + 1 < kc1);
        // lgkmcnt(0) as well: the zeroed pad columns / A_EXTRA region above were plain ds_writes, and gfx950's back-off
        // barrier does not wait for a wavefront's outstanding LDS operations by itself
#ifdef GZ2_EXP_NO_LGKM_PROLOGUE
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP) : "memory");
#endif
        __builtin_amdgcn_s_barrier();
#ifdef GZ2_STAMPS
        st1 = __builtin_amdgcn_s_memtime();
#endif
        float af[2][TM], bf[2][TN];
#ifdef GZ2_STEP_STAMPS
        unsigned long long step_cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, step_t = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (DM) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[q][j] = 0.f;
            fetch_first(kc0, 0u, af[0], bf[0]);
        } else {
            fetch(std::integral_constant<int, 0>{}, 0u, af[0], bf[0]);
        }
        lgkm_done(af[0], bf[0]);
        mask(af[0]);
        int stage = 0;
        // (dual mode: the row-shared chunks and the plain chunks are TWO loops, one after the other -- a per-chunk branch
        // between the two k-step forms inside one loop made the register allocator spill accumulators in every iteration)
        auto run_chunks = [&](int k_from, int k_to, auto Mc) {
        for (int kc = k_from; kc < k_to; ++kc) {
            int s1 = stage + 1; if (s1 >= STAGES2) s1 -= STAGES2;
            int s2 = s1 + 1; if (s2 >= STAGES2) s2 -= STAGES2;
            const uint32_t so = (uint32_t)(stage * STAGE * 4), sno = (uint32_t)(s1 * STAGE * 4);
            const bool more = kc + 2 < kc1;
            auto kstep = [&](auto Sc) {
                constexpr int S = decltype(Sc)::value;
                constexpr int MD = decltype(Mc)::value;          // multi-mode loaders: this loop's k-step form
                constexpr int c = S & 1, n = c ^ 1;
                auto mfma_row = [&](f32x16 (&cc)[TN], float a, const float (&b)[TN]) {
                    if constexpr (DM) {           // the accumulator blocks (output phases) this mode's taps feed
                        constexpr unsigned MK = mode_mask_of<AL>(MD, S);
                        if constexpr (MK == (TN == 4 ? 15u : 3u)) gz::mfma_row(cc, a, b);
                        else {
                            if constexpr ((MK & 1u) != 0) gz::mfma_one(cc[0], a, b[0]);
                            if constexpr (TN >= 2 && (MK & 2u) != 0) gz::mfma_one(cc[1], a, b[1]);
                            if constexpr (TN == 4 && (MK & 4u) != 0) gz::mfma_one(cc[2], a, b[2]);
                            if constexpr (TN == 4 && (MK & 8u) != 0) gz::mfma_one(cc[3], a, b[3]);
                        }
                    } else gz::mfma_row(cc, a, b);
                };
                auto fetch_same = [&](auto S2c, uint32_t o, float (&fa)[TM], float (&fb)[TN]) {
                    if constexpr (DM) fetch_m(S2c, Mc, o, fa, fb);
                    else fetch(S2c, o, fa, fb);
                };
                // chunk kc+2's pieces go out in the FIRST k-steps, PPS per step, one behind each MFMA row (the stage
                // they overwrite was last read in chunk kc-1, behind its barrier), so that by the last k-step exactly
                // NP are in flight
                auto pieces = [&](auto Rowc) {        // behind MFMA row `Rowc` (a single piece per k-step: behind row 1)
                    constexpr int row = decltype(Rowc)::value;
                    constexpr int r = PPS == 1 ? (row == 1 ? 0 : -1) : row;
                    constexpr int q = S * PPS + r;
                    if constexpr (r >= 0 && r < PPS && q < NP) issue_piece(kc + 2, s2, q, more);
                };
                if constexpr (S + 1 < STEPS) {
                    fetch_same(std::integral_constant<int, S + 1>{}, so, af[n], bf[n]);
                    mfma_row(acc[0], af[c][0], bf[c]);
                    pieces(std::integral_constant<int, 0>{});
                    mfma_row(acc[1], af[c][1], bf[c]);
                    pieces(std::integral_constant<int, 1>{});
                    if constexpr (TM == 4) mfma_row(acc[2], af[c][2], bf[c]);
                    pieces(std::integral_constant<int, 2>{});
                    if constexpr (TM == 4) mfma_row(acc[3], af[c][3], bf[c]);
                    pieces(std::integral_constant<int, 3>{});
                } else {
                    // last k-step: half of its MFMAs, then chunk kc+1 must have landed (all but the NP pieces of chunk
                    // kc+2) and every wavefront must be done with this stage's fragments; the first fragments of
                    // chunk kc+1 are fetched under the other half
                    mfma_row(acc[0], af[c][0], bf[c]);
                    pieces(std::integral_constant<int, 0>{});
                    if constexpr (TM == 4) mfma_row(acc[1], af[c][1], bf[c]);
                    pieces(std::integral_constant<int, 1>{});
                    pieces(std::integral_constant<int, 2>{});
                    pieces(std::integral_constant<int, 3>{});
#ifndef GZ2_EXP_NOVMWAIT
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
#endif
#ifndef GZ2_EXP_NOBARRIER
                    __builtin_amdgcn_s_barrier();
#endif
                    if constexpr (DM) fetch_first(kc + 1, sno, af[n], bf[n]);      // the next chunk's, in ITS mode
                    else fetch(std::integral_constant<int, 0>{}, sno, af[n], bf[n]);
                    if constexpr (TM == 4) {
                        mfma_row(acc[2], af[c][2], bf[c]);
                        mfma_row(acc[3], af[c][3], bf[c]);
                    } else {
                        mfma_row(acc[1], af[c][1], bf[c]);
                    }
                }
#ifdef GZ2_STEP_STAMPS      // diagnostic: cycles per k-step position, summed over the chunks
                {
                    const unsigned long long t = __builtin_amdgcn_s_memtime();
                    step_cyc[S] += t - step_t;
                    step_t = t;
                }
#endif
                lgkm_done(af[n], bf[n]);
                mask(af[n]);
            };
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
            kstep(std::integral_constant<int, 2>{});
            kstep(std::integral_constant<int, 3>{});
            kstep(std::integral_constant<int, 4>{});
            kstep(std::integral_constant<int, 5>{});
            kstep(std::integral_constant<int, 6>{});
            kstep(std::integral_constant<int, 7>{});
            if constexpr (STEPS == 12) {
                kstep(std::integral_constant<int, 8>{});
                kstep(std::integral_constant<int, 9>{});
                kstep(std::integral_constant<int, 10>{});
                kstep(std::integral_constant<int, 11>{});
            }
            stage = s1;
        }
        };
        if constexpr (DM) {          // one loop per mode, in order (a mode switch inside ONE loop spilled accumulators)
            run_chunks(kc0, min(kend, al.bound[0]), std::integral_constant<int, 0>{});
            run_chunks(max(kc0, al.bound[0]), kend, std::integral_constant<int, 1>{});
        } else {
            run_chunks(kc0, kend, std::integral_constant<int, 0>{});
        }
        // drain the LDS-DMA queue; MFMA results must have retired before the epilogue's v_accvgpr_read (the hazard
        // recognizer does not look inside the asm statements)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#ifdef GZ2_STAMPS
        st2 = __builtin_amdgcn_s_memtime();
#endif
#ifdef GZ2_STEP_STAMPS
        if (tid == 0 && blockIdx.x < 64)
            for (int q = 0; q < 8; ++q) gz2_stamps[(size_t)(8192 - 64 + blockIdx.x) * 8 + q] = step_cyc[q];
#endif
    }

    if constexpr (KG == 2) {
        // the two halves of the reduction meet: group 1 parks its accumulators in LDS (the rings are done with:
        // [register][thread], conflict-free), group 0 adds them -- a + b, the same bits whichever group held which half
        // -- and alone runs the epilogue
        __syncthreads();
        float* const park = smem2;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) park[((i * TN + j) * 16 + r) * NT + tid] = acc[i][j][r];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += park[((i * TN + j) * 16 + r) * NT + tid];
    }
#ifdef GZ2_EXP_NOSTORE   // timing experiment: keep one store so the accumulators stay live
    if (acc[0][0][0] == 123456.789f)
#endif
    if (gm.slab)
        store_slab<Epi::SWAP, TM, TN>(gm, acc, tile_m * Cfg::BM + wm * TM * 32, tile_n * Cfg::BN + wn * TN * 32, lane,
                                      gm.var_chunks ? gm.phase_slab0[y] + z : y * (int)gridDim.z + z);
    else
        Epi::template store<TM, TN>(pe, acc, tile_m * Cfg::BM + wm * TM * 32, tile_n * Cfg::BN + wn * TN * 32,
                                    lane, y, z);
#ifdef GZ2_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long* o = gz2_stamps + (size_t)blockIdx.x * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime();
        o[4] = sr0; o[5] = __builtin_amdgcn_s_memrealtime(); o[6] = kc1 - kc0; o[7] = TM * TN;
    }
#endif
}

// The same skeleton for operands that cannot arrive by LDS-DMA (weight gradient: both operands are
// reduction-contiguous in memory and are transposed on their way into LDS): loaders with the igemm_kernel interface
// issue() (global -> registers, at the top of a chunk) / commit() (registers -> LDS, late in the chunk), TWO stages
// (the stage written in chunk kc was last read in chunk kc-1, behind its barrier).  The compiler places the global
// loads and ds_writes (and the vmcnt wait between them) between the hand-ordered MFMA clusters; the wavefront drains
// its LDS queue (its own ds_writes) in front of the chunk's barrier.  Epilogue: D[m][n], lanes along n.
template <class Cfg, class AL, class BL, class Epi>
__global__ __launch_bounds__(NT, Cfg::OCC) void igemm2r_kernel(typename AL::Params pa, typename BL::Params pb,
                                                               typename Epi::Params pe, GridMap gm) {
    constexpr int LDA = AL::LD, LDB = BL::LD;
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    static_assert(TM == 4 && (TN == 2 || TN == 4) && !Epi::SWAP, "fragment registers of the hand-ordered k-step");
    constexpr int A_ELEMS = BK * LDA, B_ELEMS = BK * LDB;
    constexpr int STAGE = A_ELEMS + B_ELEMS;
    extern __shared__ __attribute__((aligned(16))) float smem2[];
    float* const ring = smem2;                           // stage i: [B image][A image]

    const int tid = threadIdx.x;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if (!gm.no_swizzle) {
        const int q = nwg >> 3, rr = nwg & 7, x = bid & 7, i = bid >> 3;
        bid = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + i;
    }
    const int y = bid % gm.ny;
    bid /= gm.ny;
    const int tile_n = bid % gm.tiles_n;
    const int tile_m = bid / gm.tiles_n;
    const int z = blockIdx.z;
    const int kc0 = z * gm.chunks_per_split;
    const int kc1 = min(gm.chunks, kc0 + gm.chunks_per_split);

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

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int half = lane >> 5, l32 = lane & 31;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)ring;
    const uint32_t a_addr = lds0 + (uint32_t)(B_ELEMS + half * LDA + wm * TM * 32 + l32) * 4u;
    const uint32_t b_addr = lds0 + (uint32_t)(half * LDB + wn * TN * 32 + l32) * 4u;
    constexpr int STEPS = BK / 2;
    auto fetch = [&](auto Sc, uint32_t so, float (&af)[TM], float (&bf)[TN]) {
        constexpr int S = decltype(Sc)::value;
        constexpr int AO = 2 * LDA * S * 4, BO = 2 * LDB * S * 4;
        af[0] = lds_rd<AO>(a_addr + so);
        af[1] = lds_rd<AO + 128>(a_addr + so);
        af[2] = lds_rd<AO + 256>(a_addr + so);
        af[3] = lds_rd<AO + 384>(a_addr + so);
        bf[0] = lds_rd<BO>(b_addr + so);
        if constexpr (TN >= 2) bf[1] = lds_rd<BO + 128>(b_addr + so);
        if constexpr (TN == 4) {
            bf[2] = lds_rd<BO + 256>(b_addr + so);
            bf[3] = lds_rd<BO + 384>(b_addr + so);
        }
    };

    if (kc0 < kc1) {
        al.issue(kc0);
        bl.issue(kc0);
        al.commit(ring + B_ELEMS);
        bl.commit(ring);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float af[2][TM], bf[2][TN];
        fetch(std::integral_constant<int, 0>{}, 0u, af[0], bf[0]);
        lgkm_done(af[0], bf[0]);
        int stage = 0;
        for (int kc = kc0; kc < kc1; ++kc) {
            const uint32_t so = (uint32_t)(stage * STAGE * 4), sno = (uint32_t)((stage ^ 1) * STAGE * 4);
            float* const nxt = ring + (stage ^ 1) * STAGE;
            const bool more = kc + 1 < kc1;
            auto kstep = [&](auto Sc) {
                constexpr int S = decltype(Sc)::value;
                constexpr int c = S & 1, n = c ^ 1;
                if constexpr (S == 0) {
                    if (more) {
                        al.issue(kc + 1);
                        bl.issue(kc + 1);
                    }
                }
                if constexpr (S + 1 < STEPS) {
                    fetch(std::integral_constant<int, S + 1>{}, so, af[n], bf[n]);
                    mfma_row_mn(acc[0], af[c][0], bf[c]);
                    mfma_row_mn(acc[1], af[c][1], bf[c]);
                    if constexpr (S == STEPS - 3) { if (more) al.commit(nxt + B_ELEMS); }
                    if constexpr (S == STEPS - 2) { if (more) bl.commit(nxt); }
                    mfma_row_mn(acc[2], af[c][2], bf[c]);
                    mfma_row_mn(acc[3], af[c][3], bf[c]);
                    lgkm_done(af[n], bf[n]);
                } else {
                    mfma_row_mn(acc[0], af[c][0], bf[c]);
                    mfma_row_mn(acc[1], af[c][1], bf[c]);
                    // this wavefront's ds_writes of the next stage are done (lgkm_done of the previous k-step waited
                    // for the whole LDS queue); every wavefront is past its last fragment read of this stage
                    __builtin_amdgcn_s_barrier();
                    fetch(std::integral_constant<int, 0>{}, sno, af[n], bf[n]);
                    mfma_row_mn(acc[2], af[c][2], bf[c]);
                    mfma_row_mn(acc[3], af[c][3], bf[c]);
                    lgkm_done(af[n], bf[n]);
                }
            };
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
            kstep(std::integral_constant<int, 2>{});
            kstep(std::integral_constant<int, 3>{});
            kstep(std::integral_constant<int, 4>{});
            kstep(std::integral_constant<int, 5>{});
            kstep(std::integral_constant<int, 6>{});
            kstep(std::integral_constant<int, 7>{});
            stage ^= 1;
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }
    Epi::template store<TM, TN>(pe, acc, tile_m * Cfg::BM + wm * TM * 32, tile_n * Cfg::BN + wn * TN * 32, lane, y, z);
}

// ---- weight gradient of the k4 s2 p1 convolution with BOTH operands arriving by LDS-DMA (igemm2w) -----------------------
// dW[ko][(c, ky, kx)] = sum over pixels (n, oy, ox) of  dy[n][ko][oy][ox] * x[n][c][2*oy - 1 + ky][2*ox - 1 + kx].
// Both operands are reduction-contiguous in memory, which is why igemm2r_kernel transposes them through registers
// (16 + 8 dword loads, 12 ds_writes and their address / mask VALU per wavefront and chunk, all paid in MFMA issue
// slots).  Here nothing is transposed on the way in:
//   A  the chunk's 16 consecutive pixels of a dy channel are 64 contiguous bytes: the LDS image is [m][16 k] (four
//      16-byte quads per row, the quad order XOR-swizzled with (m >> 2) & 3 so that the 16 lanes of a ds_read_b128
//      group hit 16 different slots), and the lane (m, half) takes its operands with ONE ds_read_b128 per four
//      k-steps.  The reduction index inside a chunk is enumerated so that this works: k-step s = 4*g + t of
//      half-wave h multiplies pixel k = 8*g + 4*h + t (the order of a sum's terms is free as long as A and B agree);
//   B  the raw input rows the chunk's pixels touch -- per input channel 2*R + 2 rows of 2*CW + 8 columns (R x CW =
//      the chunk's pixel rectangle: 1 x 16, 2 x 8 or 4 x 4; four columns either side so that every 16-byte quad is
//      entirely inside or entirely outside the image, outside = out-of-range lanes = zeros) -- land as they are, and
//      the lane (c, ky, kx) reads  image[c][2*r + ky][2*x + kx + 3]  with r, x from k: row / channel pitches are chosen
//      so that the 32 (c, ky, kx) lanes of a half-wave fall on 32 different banks, and every k-step's offset is an
//      immediate.
// No VALU and no ds_write in the k-steps (per chunk: a dozen VALU instructions for the stage offsets and the halo
// flags of the two B pieces); 24 LDS reads per chunk and wavefront instead of 48.  Ring, counted vmcnt,
// barrier placement and the epilogue (D[m][n], lanes along n, split-K slabs through the epilogue) as above.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ f32x4 lds_rd4(uint32_t byte_addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ void lgkm_done(float (&b)[2]) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1])); }
__device__ __forceinline__ void lgkm_done(f32x4 (&a)[4], float (&b)[2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]));
}

struct Wg2Params {
    const float* x;          // image side  [N, C, H, W], H = 2*OH, W = 2*OW
    const float* y;          // feature side [N, K, OH, OW]
    ConvShape s;
    FastDiv div_ohw, div_ow;
};

template <int BM>
struct WgDyA2 {
    static constexpr int PIECES = BM / 64;                 // per wavefront and chunk (BM rows x 4 quads / 64 lanes / 4)
    static constexpr int ELEMS = BM * BK;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[PIECES];
    int wave;
    __device__ __forceinline__ void init(const Wg2Params& p, int tile, int tid) {
        const ConvShape& s = p.s;
        rsrc = make_rsrc(p.y, (uint32_t)s.N * s.K * s.OH * s.OW * 4u);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int gq = (wave * PIECES + j) * 64 + lane;       // LDS quad
            const int ml = gq >> 2, qk = (gq & 3) ^ ((ml >> 2) & 3);
            const int m = tile * BM + ml;
            voff[j] = m < s.K ? (uint32_t)(m * s.OH * s.OW + 4 * qk) * 4u : OOB;
        }
    }
    __device__ __forceinline__ void issue_piece(float* stage, int j, uint32_t soff) {
        bload_lds16(rsrc, stage + (wave * PIECES + j) * 256, voff[j], soff);
    }
};

template <int BN, int CW>
struct WgImgB2 {
    static_assert(CW == 4 || CW == 8 || CW == 16, "chunk rectangles 4 x 4, 2 x 8, 1 x 16");
    static constexpr int R = 16 / CW, IMG_ROWS = 2 * R + 2;
    static constexpr int RQ = CW == 4 ? 5 : (2 * CW + 8) / 4;              // quads per image row (CW 4: one pad quad)
    static constexpr int CHQ = CW == 16 ? 41 : CW == 8 ? 37 : 52;           // quads per channel (pad quads at its end)
    static constexpr int RP = RQ * 4, CHP = CHQ * 4;                        // pitches in floats: (8*ky + 4*c + kx) mod 32,
                                                                            // (24*ky + 20*c + kx), (20*ky + 16*c + kx)
                                                                            // are 32 different banks
    static constexpr int NCH = BN / 16;
    static constexpr int PIECES = (NCH * CHQ + 255) / 256;                  // per wavefront and chunk
    static constexpr int ELEMS = PIECES * 4 * 256;
    // float offsets of the fragment read of k-step s = 4*g + t, half-wave h: pixel k = 8*g + 4*h + t = (r, x)
    static constexpr int HOFF = CW == 4 ? 2 * RP : 8, GOFF = CW == 16 ? 16 : CW == 8 ? 2 * RP : 4 * RP, TOFF = 2;
    static constexpr int ROWS_PER_CHUNK = R, TAPS = 16;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[PIECES], flags[PIECES];
    int wave;
    // float offset (inside the B image) of the lane that owns column `col` of the tile: (c, ky, kx) at pixel (0, 0)
    __device__ __forceinline__ int lane_base(int tile, int col) const {
        return (col >> 4) * CHP + ((col >> 2) & 3) * RP + (col & 3) + 3;
    }
    __device__ __forceinline__ void init(const Wg2Params& p, int tile, int tid) {
        const ConvShape& s = p.s;
        // per-lane offsets are relative to the chunk's first input pixel (2*oy0, 2*ox0) and reach one row up and four
        // columns left: the descriptor's base is moved back by that much (never dereferenced there: flagged lanes)
        const int shift = s.W + 4;
        rsrc = make_rsrc(p.x - shift, (uint32_t)(s.N * s.C * s.H * s.W + shift) * 4u);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int gq = (wave * PIECES + j) * 64 + lane;
            const int cl = gq / CHQ, rem = gq - cl * CHQ;
            const int row = rem / RQ, q = rem - row * RQ;
            const int c = tile * NCH + cl;
            const int col0 = 4 * q - 4;
            const bool inside = cl < NCH && c < s.C && row < IMG_ROWS && col0 <= 2 * CW;
            voff[j] = inside ? (uint32_t)((c * s.H + row - 1) * s.W + col0 + shift) * 4u : OOB;
            flags[j] = (row == 0 ? 1u : 0u) | (row == IMG_ROWS - 1 ? 2u : 0u) | (col0 < 0 ? 4u : 0u) |
                       (col0 >= 2 * CW ? 8u : 0u);
        }
    }
    __device__ __forceinline__ void issue_piece(float* stage, int j, uint32_t soff, uint32_t cond) {
        bload_lds16(rsrc, stage + (wave * PIECES + j) * 256, (flags[j] & cond) ? OOB : voff[j], soff);
    }
};

// The same raw-row image for any geometry with W = S * OW, H = S * OH (5x5 s2 p2, 3x3 s1 p1): S * (R - 1) + KH rows per
// channel, the column origin LP = P rounded up to a quad left of the chunk's first input column.  A 32-column block is
// no longer a whole number of channels (25 or 9 taps each), so a lane's (c, ky, kx) differs from block to block: the
// kernel keeps one base address per block.  Pitches from tools/wg_banks.py: 25 + 7 taps of two channels cannot tile the
// 32 banks with quad-aligned rows, the best layouts are 2-way on a few lanes.
template <int BN, int CW, int KH, int KW, int S, int P>
struct WgImgBG {
    static_assert(CW == 4 || CW == 8 || CW == 16, "chunk rectangles 4 x 4, 2 x 8, 1 x 16");
    static constexpr int R = 16 / CW, TAPS = KH * KW;
    static constexpr int LP = (P + 3) / 4 * 4;
    static constexpr int IMG_ROWS = S * (R - 1) + KH;
    static constexpr int RQ0 = (S * (CW - 1) + KW - 1 - P + LP) / 4 + 1;
    static constexpr bool K5 = KH == 5 && KW == 5 && S == 2 && P == 2, K3 = KH == 3 && KW == 3 && S == 1 && P == 1;
    static_assert(K5 || K3, "pitches are tabulated per geometry (tools/wg_banks.py)");
    static constexpr int RQ = RQ0 + ((K5 && CW == 4) ? 1 : 0);
    static constexpr int RP = RQ * 4;
    static constexpr int CHP = IMG_ROWS * RP + 4 * (K5 ? (CW == 16 ? 0 : CW == 8 ? 4 : 2) : (CW == 16 ? 3 : CW == 8 ? 1 : 0));
    static constexpr int CHQ = CHP / 4;
    static constexpr int NCH = (BN % TAPS == 0) ? BN / TAPS : (BN - 1) / TAPS + 2;      // channels a tile's columns can touch
    static constexpr int PIECES = (NCH * CHQ + 255) / 256;
    static constexpr int ELEMS = PIECES * 4 * 256;
    static constexpr int TOFF = S, HOFF = CW == 4 ? S * RP : 4 * S, GOFF = CW == 16 ? 8 * S : CW == 8 ? S * RP : 2 * S * RP;
    static constexpr int ROWS_PER_CHUNK = R;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff[PIECES], flags[PIECES];
    int wave;
    __device__ __forceinline__ int lane_base(int tile, int col) const {
        const int n = tile * BN + col;
        const int c = n / TAPS, tap = n - c * TAPS;
        int cl = c - (tile * BN) / TAPS;
        if (cl > NCH - 1) cl = NCH - 1;                  // columns past N (masked by the epilogue) stay inside the image
        const int ky = tap / KW, kx = tap - ky * KW;
        return cl * CHP + ky * RP + kx - P + LP;
    }
    __device__ __forceinline__ void init(const Wg2Params& p, int tile, int tid) {
        const ConvShape& s = p.s;
        const int shift = P * s.W + LP;
        rsrc = make_rsrc(p.x - shift, (uint32_t)(s.N * s.C * s.H * s.W + shift) * 4u);
        const int lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int c_lo = (tile * BN) / TAPS;
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int gq = (wave * PIECES + j) * 64 + lane;
            const int cl = gq / CHQ, rem = gq - cl * CHQ;
            const int row = rem / RQ, q = rem - row * RQ;
            const int c = c_lo + cl;
            const int col0 = 4 * q - LP;
            const bool inside = cl < NCH && c < s.C && row < IMG_ROWS && col0 <= S * CW;
            voff[j] = inside ? (uint32_t)((c * s.H + row - P) * s.W + col0 + shift) * 4u : OOB;
            flags[j] = (row < P ? 1u : 0u) | (row >= S * R + P ? 2u : 0u) | (col0 < 0 ? 4u : 0u) |
                       (col0 >= S * CW ? 8u : 0u);
        }
    }
    __device__ __forceinline__ void issue_piece(float* stage, int j, uint32_t soff, uint32_t cond) {
        bload_lds16(rsrc, stage + (wave * PIECES + j) * 256, (flags[j] & cond) ? OOB : voff[j], soff);
    }
};

template <class Cfg, class BL, class Epi>
__global__ __launch_bounds__(NT, Cfg::OCC) void igemm2w_kernel(Wg2Params p, typename Epi::Params pe, GridMap gm) {
    using AL = WgDyA2<Cfg::BM>;
    constexpr int CW = 16 / BL::ROWS_PER_CHUNK;
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    static_assert(TM == 4 && TN == 2 && !Epi::SWAP, "fragment registers of the hand-ordered k-step");
    constexpr int A_ELEMS = AL::ELEMS, B_ELEMS = BL::ELEMS, STAGE = A_ELEMS + B_ELEMS;
    extern __shared__ __attribute__((aligned(16))) float smem2[];
    float* const ring = smem2;                           // stage i: [B image][A image]

    const int tid = threadIdx.x;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if (!gm.no_swizzle) {
        const int q = nwg >> 3, rr = nwg & 7, x = bid & 7, i = bid >> 3;
        bid = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + i;
    }
    const int tile_n = bid % gm.tiles_n;
    const int tile_m = bid / gm.tiles_n;
    const int z = blockIdx.z;
    const int kc0 = z * gm.chunks_per_split;
    const int kc1 = min(gm.chunks, kc0 + gm.chunks_per_split);

    AL al;
    BL bl;
    al.init(p, tile_m, tid);
    bl.init(p, tile_n, tid);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int half = lane >> 5, l32 = lane & 31;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)ring;
    // A: row m = 16 floats, quad (2*g + half) ^ swizzle; the swizzle (m >> 2) & 3 only depends on the lane
    const int swz = (l32 >> 2) & 3;
    uint32_t a_addr[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
        a_addr[g] = lds0 + (uint32_t)(B_ELEMS + (wm * TM * 32 + l32) * BK + (((2 * g + half) ^ swz) << 2)) * 4u;
    // B: this lane's (c, ky, kx) in each of its two 32-column blocks
    uint32_t b_addr[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
        b_addr[j] = lds0 + (uint32_t)(bl.lane_base(tile_n, wn * TN * 32 + j * 32 + l32) + half * BL::HOFF) * 4u;

    constexpr int NPA = AL::PIECES, NPB = BL::PIECES, NP = NPA + NPB, STEPS = BK / 2;
    static_assert(NP <= STEPS, "at most one LDS-DMA piece per k-step");
    const ConvShape& s = p.s;
    const int OHW = s.OH * s.OW;
    uint32_t soff_a = 0, soff_b = 0, cond = 0;          // of the chunk being issued
    auto locate = [&](int kc, bool live) {
        const uint32_t p0 = (uint32_t)kc * BK;
        const uint32_t n = fdiv(p0, p.div_ohw);
        const uint32_t rem = p0 - n * (uint32_t)OHW;
        const uint32_t oy0 = fdiv(rem, p.div_ow);
        const uint32_t ox0 = rem - oy0 * (uint32_t)s.OW;
        soff_a = live ? (n * (uint32_t)(s.K * OHW) + rem) * 4u : SOFF_OOB;
        const uint32_t st = (uint32_t)(s.H / s.OH);           // stride (H = S * OH)
        soff_b = live ? (n * (uint32_t)(s.C * s.H * s.W) + st * oy0 * (uint32_t)s.W + st * ox0) * 4u : SOFF_OOB;
        cond = (oy0 == 0 ? 1u : 0u) | ((int)oy0 + BL::ROWS_PER_CHUNK == s.OH ? 2u : 0u) | (ox0 == 0 ? 4u : 0u) |
               ((int)ox0 + CW == s.OW ? 8u : 0u);
    };
    auto issue_piece = [&](int st, int q) {
        float* sb = ring + st * STAGE;
        if (q < NPA) al.issue_piece(sb + B_ELEMS, q, soff_a);
        else bl.issue_piece(sb, q - NPA, soff_b, cond);
    };
    auto fetch_a = [&](auto Gc, uint32_t so, f32x4 (&af)[TM]) {
        constexpr int G = decltype(Gc)::value;
        af[0] = lds_rd4<0>(a_addr[G] + so);
        af[1] = lds_rd4<32 * BK * 4>(a_addr[G] + so);
        af[2] = lds_rd4<64 * BK * 4>(a_addr[G] + so);
        af[3] = lds_rd4<96 * BK * 4>(a_addr[G] + so);
    };
    auto fetch_b = [&](auto Sc, uint32_t so, float (&bf)[TN]) {
        constexpr int S = decltype(Sc)::value;
        constexpr int BO = ((S >> 2) * BL::GOFF + (S & 3) * BL::TOFF) * 4;
        bf[0] = lds_rd<BO>(b_addr[0] + so);
        bf[1] = lds_rd<BO>(b_addr[1] + so);
    };

    if (kc0 < kc1) {
        locate(kc0, true);
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_piece(0, q);
        locate(kc0 + 1, kc0 + 1 < kc1);
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_piece(1, q);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        __builtin_amdgcn_s_barrier();
        f32x4 af[2][TM];
        float bf[2][TN];
        fetch_a(std::integral_constant<int, 0>{}, 0u, af[0]);
        fetch_b(std::integral_constant<int, 0>{}, 0u, bf[0]);
        lgkm_done(af[0], bf[0]);
        int stage = 0;
        for (int kc = kc0; kc < kc1; ++kc) {
            int s1 = stage + 1; if (s1 >= STAGES2) s1 -= STAGES2;
            int s2 = s1 + 1; if (s2 >= STAGES2) s2 -= STAGES2;
            const uint32_t so = (uint32_t)(stage * STAGE * 4), sno = (uint32_t)(s1 * STAGE * 4);
            locate(kc + 2, kc + 2 < kc1);
            auto kstep = [&](auto Sc) {
                constexpr int S = decltype(Sc)::value;
                constexpr int c = S & 1, n = c ^ 1, g = S >> 2, t = S & 3;
                if constexpr (S + 1 < STEPS) {
                    fetch_b(std::integral_constant<int, S + 1>{}, so, bf[n]);
                    if constexpr (S == 0) fetch_a(std::integral_constant<int, 1>{}, so, af[1]);
                    mfma_row_mn(acc[0], af[g][0][t], bf[c]);
                    mfma_row_mn(acc[1], af[g][1][t], bf[c]);
                    if constexpr (S < NP) issue_piece(s2, S);
                    mfma_row_mn(acc[2], af[g][2][t], bf[c]);
                    mfma_row_mn(acc[3], af[g][3][t], bf[c]);
                    if constexpr (S == 0) lgkm_done(af[1], bf[n]);
                    else lgkm_done(bf[n]);
                } else {
                    mfma_row_mn(acc[0], af[g][0][t], bf[c]);
                    mfma_row_mn(acc[1], af[g][1][t], bf[c]);
                    if constexpr (S < NP) issue_piece(s2, S);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
                    __builtin_amdgcn_s_barrier();
                    fetch_b(std::integral_constant<int, 0>{}, sno, bf[n]);
                    fetch_a(std::integral_constant<int, 0>{}, sno, af[0]);
                    mfma_row_mn(acc[2], af[g][2][t], bf[c]);
                    mfma_row_mn(acc[3], af[g][3][t], bf[c]);
                    lgkm_done(af[0], bf[n]);
                }
            };
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
            kstep(std::integral_constant<int, 2>{});
            kstep(std::integral_constant<int, 3>{});
            kstep(std::integral_constant<int, 4>{});
            kstep(std::integral_constant<int, 5>{});
            kstep(std::integral_constant<int, 6>{});
            kstep(std::integral_constant<int, 7>{});
            stage = s1;
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    }
    Epi::template store<TM, TN>(pe, acc, tile_m * Cfg::BM + wm * TM * 32, tile_n * Cfg::BN + wn * TN * 32, lane, 0, z);
}

template <class Cfg, class BL>
constexpr size_t igemm2w_lds_bytes() {
    return (size_t)STAGES2 * (WgDyA2<Cfg::BM>::ELEMS + BL::ELEMS) * 4;
}

template <class Cfg, class BL, class Epi>
inline int launch_igemm2w(const Wg2Params& p, const typename Epi::Params& pe, int M, int N, int K, int splits,
                          hipStream_t stream) {
    GridMap gm;
    gm.no_swizzle = knobs().no_xcd_swizzle;
    gm.var_chunks = 0;
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
    gm.ny = 1;
    gm.stagger = 0;
    for (int i = 0; i < 8; ++i) gm.phase_nz[i] = gm.phase_slab0[i] = 0;
    dim3 grid(gm.tiles_m * gm.tiles_n, 1, nz);
    constexpr size_t lds = igemm2w_lds_bytes<Cfg, BL>();
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    auto kern = igemm2w_kernel<Cfg, BL, Epi>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return launch_status();
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, p, pe, gm);
    return launch_status();
}

// split-K through the epilogue's own slabs (pe.slab_stride), as launch_igemm does for the weight gradient
template <class Cfg, class AL, class BL, class Epi>
inline int launch_igemm2r(const typename AL::Params& pa, const typename BL::Params& pb, const typename Epi::Params& pe,
                          int M, int N, int K, int splits, hipStream_t stream) {
    GridMap gm;
    gm.no_swizzle = knobs().no_xcd_swizzle;
    gm.var_chunks = 0;
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
    gm.ny = 1;
    gm.stagger = 0;
    for (int i = 0; i < 8; ++i) gm.phase_nz[i] = gm.phase_slab0[i] = 0;
    dim3 grid(gm.tiles_m * gm.tiles_n, 1, nz);
    constexpr size_t lds = (size_t)2 * BK * (AL::LD + BL::LD) * 4;
    auto kern = igemm2r_kernel<Cfg, AL, BL, Epi>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return launch_status();
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, pa, pb, pe, gm);
    return launch_status();
}

template <class Cfg, class AL, class BL>
constexpr size_t igemm2_lds_bytes() {
    return (size_t)(STAGES2 * (AL::ROWS * AL::LD + igemm2_a_extra<AL>() + BL::ROWS * BL::LD) + 32) * 4;
}
template <class Cfg, class AL, class BL>
constexpr size_t igemm2_lds_bytes_kg2() {
    const size_t rings = (size_t)(2 * igemm2_ring_floats<Cfg, AL, BL, 2>() + 32) * 4;
    const size_t park = (size_t)Cfg::TM * Cfg::TN * 16 * NT * 4;
    return rings > park ? rings : park;
}

// Which launches run two wave groups per workgroup (igemm2_kernel<.., KG = 2>): the k4 s2 p1 loaders of the DCGAN
// layers, 128- or 64-wide tiles, when the grid gives a CU at most ONE workgroup and each group still has a reduction
// worth pipelining.  Pure host logic: gz_conv2d_plan reports it, tests/golden/dispatch_plan.json pins it.
template <class T, class = void>
struct is_tapgather : std::false_type {};
template <class T>
struct is_tapgather<T, std::void_t<decltype(T::TAPGATHER)>> : std::true_type {};
template <class Cfg, class AL>
constexpr bool igemm2_kg2_built() {
    return Cfg::TN <= 2 && Cfg::WM * Cfg::WN == 4 && Cfg::WN == 2 && !is_dualmode<AL>::value &&
           (is_rowshare<AL>::value || is_fwdrows<AL>::value || is_tapgather<AL>::value);
}
inline bool igemm2_use_kg2(long long workgroups, int chunks_per_workgroup) {
    if (knobs().no_kg2) return false;
    return workgroups <= cus() && chunks_per_workgroup >= knobs().kg2_min_chunks;
}

// same contract as launch_igemm
template <class Cfg, class AL, class BL, class Epi>
inline int launch_igemm2(const typename AL::Params& pa, const typename BL::Params& pb, const typename Epi::Params& pe,
                         int M, int N, int K, int ny, int splits, hipStream_t stream, float* slab = nullptr,
                         const int* phase_chunks = nullptr) {
    static_assert(Epi::SWAP, "transposed accumulators (lanes along m)");
    GridMap gm;
    gm.no_swizzle = knobs().no_xcd_swizzle;
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
    // stagger (see the kernel), an experiment that stays OFF: GZ_IGEMM2_STAGGER=<percent of a tile>.  Measured (in-kernel
    // stamps): co-resident workgroups drift apart by themselves -- the average workgroup's loop takes 469 k cycles where
    // two in lockstep would take 524 k -- and a forced half-tile offset changes neither the layer (143.3 -> 142.8
    // TFLOP/s) nor the step (21.43 vs 21.40 ms).  What a launch does lose is its ramp: ~19 us until the first
    // round's prologues are through plus the last round's tail, ~6 % of a 1 ms launch.
    const int stagger_pct = knobs().igemm2_stagger;
    gm.stagger = (Cfg::OCC == 2 && nz == 1 && grid.x >= 1536)
                     ? (int)((long long)gm.chunks * 8 * Cfg::TM * Cfg::TN * 64 * 2 * stagger_pct / 100) : 0;
    if (slab && nz > 1) gm.slab = slab;
    SlabMap sm;
    sm.var = gm.slab && gm.var_chunks;
    for (int i = 0, at = 0; i < 8; ++i) {
        int n = (sm.var && i < ny) ? (gm.phase_chunks[i] + gm.chunks_per_split - 1) / gm.chunks_per_split : 0;
        gm.phase_nz[i] = sm.nz[i] = n;
        gm.phase_slab0[i] = sm.slab0[i] = at;
        at += n;
    }
    const size_t lds_extra = (size_t)knobs().igemm2_lds;   // experiment:
    bool launched = false;
    if constexpr (igemm2_kg2_built<Cfg, AL>()) {
        if (igemm2_use_kg2((long long)grid.x * nz, gm.chunks_per_split)) {
            const size_t lds2 = igemm2_lds_bytes_kg2<Cfg, AL, BL>();
            auto kern2 = igemm2_kernel<Cfg, AL, BL, Epi, 2>;
            static bool attr2_done = false;
            if (!attr2_done) {
                if (hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
                    return launch_status();
                attr2_done = true;
            }
            hipLaunchKernelGGL(kern2, grid, dim3(2 * NT), lds2, stream, pa, pb, pe, gm);
            launched = true;
        }
    }
    if (!launched) {
    const size_t lds = igemm2_lds_bytes<Cfg, AL, BL>() + lds_extra;             // throttles workgroups per CU
    auto kern = igemm2_kernel<Cfg, AL, BL, Epi>;
    static bool attr_done = false;       // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return launch_status();
        attr_done = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, stream, pa, pb, pe, gm);
    }
    if (gm.slab) {
        const int fm = (M + 31) / 32, fn = (N + 31) / 32;
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
