// igemm_launch.h -- host-side entry points of the implicit-GEMM kernel family.  The template instantiations are compiled in their
// own translation units (igemm2_inst.hip once per tile configuration, igemm_tiled_inst.hip) so that the library builds in parallel.
#pragma once
#include "igemm.hip.h"

namespace rvc {

// tile configurations of the register-direct kernel: index -> (MF, NF); every tile exists with KS in {1, 4, 8, 16}
static const int kMF[5] = {1, 1, 1, 2, 2}, kNF[5] = {1, 2, 4, 2, 4};

// lean kernel (igemm2): 2-D tile grid, LDS = offset table (none for LIN layers) + the KS partial tiles.  ea / eb: optional start /
// stop events of hipExtLaunchKernelGGL (the dispatch's own begin / end timestamps: what rocprofv3 --kernel-trace reports).
void launch_igemm2(int cfg, int ks, bool pre, bool lin, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
void launch_igemm2_cfg0(int ks, bool pre, bool lin, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_igemm2_cfg1(int ks, bool pre, bool lin, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_igemm2_cfg2(int ks, bool pre, bool lin, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_igemm2_cfg3(int ks, bool pre, bool lin, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_igemm2_cfg4(int ks, bool pre, bool lin, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
// the first-generation kernel remains for the two-stage grid-level split-K (offset table too long for LDS): 16x16 tiles only
void launch_igemm_v1(bool pre, const IgemmP &p, dim3 grid, hipStream_t s);
// workgroup-tiled throughput kernels: lc 0-2 / 6 = igemm_lds_kernel (128x128, 64x256, 32x256, 48x256), 3-5 / 7 / 8 = igemm32_kernel
void launch_igemm_tiled(int lc, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);

void launch_igemm_tiled_p0(int lc, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_igemm_tiled_p1(int lc, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_igemm_tiled_p2(int lc, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_igemm_tiled_p3(int lc, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);

// conv_tile_kernel (conv_tile.hip.h) tiles: 0 = 128 x 16 (four waves stacked in M), 1 = 64 x 32, 2 = 32 x 64 (2 x 2 waves), each with one or two K shares.
// WF = waves per K share x MF x NF.
static const int kTileBM[3] = {128, 64, 32}, kTileBN[3] = {16, 32, 64}, kTileWF[3] = {8, 8, 8};
void conv_tile_prepare_device();
void launch_conv_tile(int tile, int kshares, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);

// igemm2w_kernel (register-direct 32x32x2, table-free 1x1 layers at a few streams): tile 0 = 32 x 32 per wave, 1 = 64 x 32, 2 = 64 x 64; ks waves split K
static const int kG2wMT[3] = {1, 2, 2}, kG2wNT[3] = {1, 1, 2};
void igemm2w_prepare_device();
void launch_igemm2w(int tile, int ks, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
void launch_igemm2w_ln(int ks, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);      // LayerNorm-consumer variant: tile 0, ks 4 / 8
void launch_igemm2w_t0(int ks, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_igemm2w_t1(int ks, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_igemm2w_t2(int ks, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);

// igemm32l_kernel (igemm32l.hip.h): igemm32_kernel's tiles 3 / 7 / 8 for table-free 1x1 layers, buffer loads with scalar row offsets
void launch_igemm32l(int lc, int mode, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);      // mode 0 table-free, 1 table, 2 table + fused input LeakyReLU
void launch_igemm32l_p0(int mode, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_igemm32l_p1(int lc, int mode, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);

// conv32s_kernel (conv32s.hip.h: stride-1 1-D convolutions at many streams, input staged once per workgroup and 32-channel block, 32x32x2 MFMAs).
// Workgroup tiles BM x BN (every wave 32 x 64): 0 = 32 x 256, 1 = 64 x 128, 2 = 128 x 64
static const int kC32sBM[3] = {32, 64, 128}, kC32sBN[3] = {256, 128, 64};
static const int kC32sCB = 32, kC32sCS = 36;        // channels staged per block, LDS column stride (floats)
void launch_conv32s(int tile, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
void launch_conv32s_p0(const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_conv32s_p1(int tile, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);
void launch_conv32s_p2(const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb);

// exploratory split-bf16 GEMM (igemm_bf3_kernel): 128 x 128 workgroup tile; bf3_pack builds the weight panels it reads
void launch_igemm_bf3(bool lin, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr);
void bf3_pack(const float *wfrag, int M, int nchunks, void *out, hipStream_t s);

template <typename K> static inline void launch_k(K kern, const IgemmP &p, dim3 grid, dim3 block, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (ea) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, s, ea, eb, 0, p);
    else hipLaunchKernelGGL(kern, grid, block, lds, s, p);
}

}  // namespace rvc
