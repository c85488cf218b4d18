// conv_tile_inst.hip -- instantiations of conv_tile_kernel (its own translation unit: builds in parallel with the rest of the family).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "igemm_launch.h"
#include "conv_tile.hip.h"

namespace rvc {

#define RVC_CT_ALL(X) X(4, 1, 2, 1, 1) X(4, 1, 1, 2, 1) X(2, 2, 1, 2, 1) X(4, 1, 2, 1, 2) X(4, 1, 1, 2, 2) X(2, 2, 1, 2, 2)

// the tiles of the long-dilation phases pass 64 KB of LDS: raise the limit on the CURRENT device (called when an engine is created on it)
void conv_tile_prepare_device()
{
#define RVC_CT_ATTR(a, b, c, d, e) (void)hipFuncSetAttribute((const void *)conv_tile_kernel<a, b, c, d, e>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    RVC_CT_ALL(RVC_CT_ATTR)
#undef RVC_CT_ATTR
}

void launch_conv_tile(int tile, int kshares, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
#define RVC_CT_GO(a, b, c, d, e) { launch_k(conv_tile_kernel<a, b, c, d, e>, p, grid, dim3(64 * a * b * e), lds, s, ea, eb); return; }
    const int k2 = kshares == 2;
    switch (tile) {
    case 0: if (k2) RVC_CT_GO(4, 1, 2, 1, 2) else RVC_CT_GO(4, 1, 2, 1, 1)
    case 1: if (k2) RVC_CT_GO(4, 1, 1, 2, 2) else RVC_CT_GO(4, 1, 1, 2, 1)
    default: if (k2) RVC_CT_GO(2, 2, 1, 2, 2) else RVC_CT_GO(2, 2, 1, 2, 1)
    }
#undef RVC_CT_GO
#undef RVC_CT_ALL
}

}  // namespace rvc
