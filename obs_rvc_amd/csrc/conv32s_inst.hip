// conv32s_inst.hip -- instantiations of conv32s_kernel (its own translation units, -DRVC_C32S_PART=0..2: they build in parallel with the rest of the family).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "igemm_launch.h"
#include "conv32s.hip.h"

namespace rvc {

#ifndef RVC_C32S_PART
#error "compile with -DRVC_C32S_PART=0..2"
#endif

// tile 0 = 32 x 256 (four waves side by side), 1 = 64 x 128 (2 x 2 waves; | 4 = conv32s_buf_kernel), 2 = 128 x 64 (four waves stacked in M; conv32s_buf_kernel); every wave owns 32 x 64 outputs
#if RVC_C32S_PART == 0
void launch_conv32s_p0(const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb) { launch_k(conv32s_kernel<1, 4, 1, 2>, p, grid, dim3(256), lds, s, ea, eb); }
void launch_conv32s(int tile, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (tile == 0) launch_conv32s_p0(p, grid, lds, s, ea, eb);
    else if ((tile & 3) == 1) launch_conv32s_p1(tile, p, grid, lds, s, ea, eb);
    else launch_conv32s_p2(p, grid, lds, s, ea, eb);
}
#elif RVC_C32S_PART == 1
void launch_conv32s_p1(int tile, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (tile & 4) launch_k(conv32s_buf_kernel<2, 2, 1, 2>, p, grid, dim3(256), lds, s, ea, eb);       // (below 24 streams: plan.hip)
    else launch_k(conv32s_kernel<2, 2, 1, 2>, p, grid, dim3(256), lds, s, ea, eb);
}
#else
void launch_conv32s_p2(const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb) { launch_k(conv32s_buf_kernel<4, 1, 1, 2>, p, grid, dim3(256), lds, s, ea, eb); }
#endif

}  // namespace rvc
