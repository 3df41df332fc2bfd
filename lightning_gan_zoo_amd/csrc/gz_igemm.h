// fp32 implicit-GEMM core for gfx950: v_mfma_f32_32x32x2_f32 tiles fed from LDS.
//
//   C[m][n] = sum_k A[m][k] * B[k][n]
//
// One 256-thread workgroup (4 wavefronts of 64) owns a BM x BN tile of C.  K is
// walked in chunks of BK=16.  Both operands live in LDS as [k][mn] images with
// the M/N index contiguous: lane l of a wave feeds the MFMA with
//   A[m = l&31][k = l>>5]   and   B[k = l>>5][n = l&31]
// (one dword each), so a fragment read is a conflict-free ds_read_b32 of 32
// consecutive words per half-wave and NCHW's pixel-contiguity maps straight onto
// the M (forward / dgrad) or K (wgrad) axis without any transposition in HBM.
//
// Staging is global -> VGPR -> LDS, software pipelined over two LDS buffers: the
// loads of chunk t+1 are issued before the MFMAs of chunk t and written to the
// other buffer after them; one barrier per chunk.  Loads are raw buffer loads:
// out-of-image taps / tails get a voffset >= num_records and read as 0 without
// a branch.  fp32-input MFMA runs at 256 FLOP/clk/CU (= 157 TF peak), i.e. a
// 128x128x16 chunk takes 2048 clk/CU, which leaves the memory pipeline ~8 B/clk/CU
// to fill -- the loaders are deliberately simple.
//
// "Loader" concept (A or B operand):
//   struct Params;  static constexpr int LD;        // LDS leading dimension
//   void init(const Params&, int tile, int y, int tid);
//   void issue(int chunk);                          // global -> registers
//   void commit(float* lds);                        // registers -> lds[k*LD + mn]
// "Epilogue" concept:
//   struct Params;
//   void store(const Params&, acc, m_base, n_base, lane, y, z);
//
// The core is split over four headers (round 5; one 3.7 k-line file before), each including the one before it:
//   gz_igemm_loaders.h    primitives (raw buffer loads, LDS-DMA), TileCfg, the round-1/2 operand loaders
//   gz_igemm_epilogues.h  epilogues of both skeletons
//   gz_igemm_core.h       igemm_kernel (rounds 1-2), launch_igemm, split-K finish
//   gz_igemm2.h           the igemm2 skeleton (rounds 3-5): igemm2 / igemm2w / igemm2r kernels, their loaders, launchers
#pragma once
#include "gz_igemm2.h"
