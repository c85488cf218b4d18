// igemm32l_inst.hip -- instantiations of igemm32l_kernel (-DRVC_G32L_PART=0..1: two translation units that build in parallel with the rest of the family).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "igemm_launch.h"
#include "igemm32l.hip.h"

namespace rvc {

#ifndef RVC_G32L_PART
#error "compile with -DRVC_G32L_PART=0..1"
#endif

// lc = the tile ids of launch_igemm_tiled: 3 = 128 x 128 (2 x 2 waves of 64 x 64), 7 = 128 x 64 (four waves stacked in M), 8 = 64 x 64
// mode 0 = table-free, 1 = offset table, 2 = offset table + fused input LeakyReLU
#define RVC_G32L_GO(a, b, c, d) { if (mode == 0) launch_k(igemm32l_kernel<a, b, c, d, false, false>, p, grid, dim3(256), lds, s, ea, eb); \
                                  else if (mode == 1) launch_k(igemm32l_kernel<a, b, c, d, true, false>, p, grid, dim3(256), lds, s, ea, eb); \
                                  else launch_k(igemm32l_kernel<a, b, c, d, true, true>, p, grid, dim3(256), lds, s, ea, eb); }
#if RVC_G32L_PART == 0
void launch_igemm32l_p0(int mode, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb) { RVC_G32L_GO(2, 2, 2, 2) }
void launch_igemm32l(int lc, int mode, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (lc == 3) launch_igemm32l_p0(mode, p, grid, lds, s, ea, eb);
    else launch_igemm32l_p1(lc, mode, p, grid, lds, s, ea, eb);
}
#else
void launch_igemm32l_p1(int lc, int mode, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (lc == 7) RVC_G32L_GO(4, 1, 1, 2)
    else RVC_G32L_GO(2, 2, 1, 1)
}
#endif
#undef RVC_G32L_GO

}  // namespace rvc
