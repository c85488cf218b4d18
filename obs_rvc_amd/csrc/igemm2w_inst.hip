// igemm2w_inst.hip (-DRVC_G2W_PART=0..2: one unit per wave tile, compiled in parallel) -- instantiations of igemm2w_kernel, the register-direct
// 32x32x2 kernel of the table-free 1x1 layers at a few streams (igemm.hip.h): wave tiles 32 x 32, 64 x 32, 64 x 64, each with 1 / 2 / 3 / 4 / 6 / 8
// waves per workgroup splitting K.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "igemm_launch.h"

#ifndef RVC_G2W_PART
#define RVC_G2W_PART 0
#endif

namespace rvc {

#if RVC_G2W_PART == 0
#define G2W_MT 1
#define G2W_NT 1
#define G2W_FN launch_igemm2w_t0
#elif RVC_G2W_PART == 1
#define G2W_MT 2
#define G2W_NT 1
#define G2W_FN launch_igemm2w_t1
#else
#define G2W_MT 2
#define G2W_NT 2
#define G2W_FN launch_igemm2w_t2
#endif

template <int KS> static void g2w_attr()
{
    // the K-split reduction keeps KS partial tiles in LDS: up to 128 KB (64 x 64 tile, 8 waves)
    (void)hipFuncSetAttribute((const void *)igemm2w_kernel<G2W_MT, G2W_NT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
}
void G2W_FN(int ks, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (ks < 0) {      // once per device (igemm2w_prepare_device)
        g2w_attr<2>(); g2w_attr<3>(); g2w_attr<4>(); g2w_attr<6>(); g2w_attr<8>();
#if RVC_G2W_PART == 0
        g2w_attr<12>(); g2w_attr<16>();
#endif
        return;
    }
    switch (ks) {
#if RVC_G2W_PART == 0
    // round 6 (one-stream experiment, VERDICT r5 #6b): 12 / 16 waves splitting K on the 32 x 32 tile -- 96 tiles of a 768-row panel at one stream are
    // 768 waves with eight shares, fewer than the 896 SIMDs of the ContentVec partition
    case 12: launch_k(igemm2w_kernel<G2W_MT, G2W_NT, 12>, p, grid, dim3(768), lds, s, ea, eb); return;
    case 16: launch_k(igemm2w_kernel<G2W_MT, G2W_NT, 16>, p, grid, dim3(1024), lds, s, ea, eb); return;
#endif
    case 1: launch_k(igemm2w_kernel<G2W_MT, G2W_NT, 1>, p, grid, dim3(64), lds, s, ea, eb); return;
    case 2: launch_k(igemm2w_kernel<G2W_MT, G2W_NT, 2>, p, grid, dim3(128), lds, s, ea, eb); return;
    case 3: launch_k(igemm2w_kernel<G2W_MT, G2W_NT, 3>, p, grid, dim3(192), lds, s, ea, eb); return;
    case 4: launch_k(igemm2w_kernel<G2W_MT, G2W_NT, 4>, p, grid, dim3(256), lds, s, ea, eb); return;
    case 6: launch_k(igemm2w_kernel<G2W_MT, G2W_NT, 6>, p, grid, dim3(384), lds, s, ea, eb); return;
    default: launch_k(igemm2w_kernel<G2W_MT, G2W_NT, 8>, p, grid, dim3(512), lds, s, ea, eb); return;
    }
}

#if RVC_G2W_PART == 0
// LayerNorm-consumer instantiations (32 x 32 wave tile, four or eight waves splitting K): the one-stream QKV / first FFN projections of ContentVec
void launch_igemm2w_ln(int ks, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (ks < 0) {
        (void)hipFuncSetAttribute((const void *)igemm2w_kernel<1, 1, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        (void)hipFuncSetAttribute((const void *)igemm2w_kernel<1, 1, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        return;
    }
    if (ks == 4) launch_k(igemm2w_kernel<1, 1, 4, true>, p, grid, dim3(256), lds, s, ea, eb);
    else launch_k(igemm2w_kernel<1, 1, 8, true>, p, grid, dim3(512), lds, s, ea, eb);
}
void launch_igemm2w(int tile, int ks, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (tile == 0) launch_igemm2w_t0(ks, p, grid, lds, s, ea, eb);
    else if (tile == 1) launch_igemm2w_t1(ks, p, grid, lds, s, ea, eb);
    else launch_igemm2w_t2(ks, p, grid, lds, s, ea, eb);
}
void igemm2w_prepare_device()
{
    IgemmP p{};
    for (int t = 0; t < 3; t++) launch_igemm2w(t, -1, p, dim3(1), 0, nullptr, nullptr, nullptr);
    launch_igemm2w_ln(-1, p, dim3(1), 0, nullptr, nullptr, nullptr);
}
#endif

}  // namespace rvc
