// conv2d32s_inst.hip -- instantiations of conv2d32s_kernel (its own translation units, -DRVC_C2D_PART=0..1).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "igemm_launch.h"
#include "conv2d32s.hip.h"

namespace rvc {

#ifndef RVC_C2D_PART
#error "compile with -DRVC_C2D_PART=0..1"
#endif

// tile 0 = 32 x 256 (four waves side by side, every wave 32 x 64), 1 = 64 x 128 (2 x 2 waves), 2 = 128 x 64 (four waves stacked in M), 3 = 32 x 128 (four waves side by
// side, every wave 32 x 32: short images, where a 256-column tile would be mostly padding); 192 staged columns beyond the tile: images up to 93 columns wide
#if RVC_C2D_PART == 0
void launch_conv2d32s_a(int tile, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (tile == 0) launch_k(conv2d32s_kernel<1, 4, 1, 2, 192>, p, grid, dim3(256), lds, s, ea, eb);
    else launch_k(conv2d32s_kernel<1, 4, 1, 1, 192>, p, grid, dim3(256), lds, s, ea, eb);
}
void launch_conv2d32s(int tile, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (tile == 0 || tile == 3) launch_conv2d32s_a(tile, p, grid, lds, s, ea, eb);
    else launch_conv2d32s_b(tile, p, grid, lds, s, ea, eb);
}
#else
void launch_conv2d32s_b(int tile, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (tile == 1) launch_k(conv2d32s_kernel<2, 2, 1, 2, 192>, p, grid, dim3(256), lds, s, ea, eb);
    else launch_k(conv2d32s_kernel<4, 1, 1, 2, 192>, p, grid, dim3(256), lds, s, ea, eb);
}
#endif

}  // namespace rvc
