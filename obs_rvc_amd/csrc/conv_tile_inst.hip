// conv_tile_inst.hip -- instantiations of conv_tile_kernel (its own translation unit: builds in parallel with the rest of the family).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "igemm_launch.h"
#include "conv_tile.hip.h"

namespace rvc {

void launch_conv_tile(int tc, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    // tc = tile (0: 128 x 16, 1: 64 x 32, 2: 32 x 64) + 4 * (K shares - 1)
    static const bool big_lds = [] {        // tiles of the long-dilation phases pass 64 KB
        for (const void *f : {(const void *)conv_tile_kernel<4, 1, 2, 1, 1>, (const void *)conv_tile_kernel<4, 1, 1, 2, 1>, (const void *)conv_tile_kernel<2, 2, 1, 2, 1>,
                              (const void *)conv_tile_kernel<4, 1, 2, 1, 2>, (const void *)conv_tile_kernel<4, 1, 1, 2, 2>, (const void *)conv_tile_kernel<2, 2, 1, 2, 2>,
                              (const void *)conv_tile_kernel<2, 1, 2, 2, 2>})
            (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        return true;
    }();
    (void)big_lds;
    switch (tc) {
    case 0: launch_k(conv_tile_kernel<4, 1, 2, 1, 1>, p, grid, dim3(256), lds, s, ea, eb); return;
    case 1: launch_k(conv_tile_kernel<4, 1, 1, 2, 1>, p, grid, dim3(256), lds, s, ea, eb); return;
    case 2: launch_k(conv_tile_kernel<2, 2, 1, 2, 1>, p, grid, dim3(256), lds, s, ea, eb); return;
    case 3: launch_k(conv_tile_kernel<2, 1, 2, 2, 2>, p, grid, dim3(256), lds, s, ea, eb); return;        // 64 x 32: two waves stacked in M x two K shares, 2 x 2 fragments per wave
    case 4: launch_k(conv_tile_kernel<4, 1, 2, 1, 2>, p, grid, dim3(512), lds, s, ea, eb); return;
    case 5: launch_k(conv_tile_kernel<4, 1, 1, 2, 2>, p, grid, dim3(512), lds, s, ea, eb); return;
    default: launch_k(conv_tile_kernel<2, 2, 1, 2, 2>, p, grid, dim3(512), lds, s, ea, eb); return;
    }
}

}  // namespace rvc
