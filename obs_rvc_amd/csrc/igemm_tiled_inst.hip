// igemm_tiled_inst.hip (-DRVC_TILED_PART=0..3: four units, compiled in parallel) -- the dispatcher over the five igemm2 tile configurations, the first-generation kernel (grid-level split-K
// fallback) and the instantiations of the workgroup-tiled throughput kernels (igemm_lds_kernel, igemm32_kernel).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "igemm_launch.h"

#ifndef RVC_TILED_PART
#define RVC_TILED_PART 0
#endif

namespace rvc {

#if RVC_TILED_PART == 0 || defined(RVC_UNITY)
void launch_igemm2(int cfg, int ks, bool pre, bool lin, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    switch (cfg) {
    case 0: launch_igemm2_cfg0(ks, pre, lin, p, grid, lds, s, ea, eb); return;
    case 1: launch_igemm2_cfg1(ks, pre, lin, p, grid, lds, s, ea, eb); return;
    case 2: launch_igemm2_cfg2(ks, pre, lin, p, grid, lds, s, ea, eb); return;
    case 3: launch_igemm2_cfg3(ks, pre, lin, p, grid, lds, s, ea, eb); return;
    default: launch_igemm2_cfg4(ks, pre, lin, p, grid, lds, s, ea, eb); return;
    }
}

void launch_igemm_v1(bool pre, const IgemmP &p, dim3 grid, hipStream_t s)
{
    const size_t lds = (size_t)p.chunks_per_split * 16 * sizeof(int);
    if (pre) hipLaunchKernelGGL((igemm_kernel<1, 1, 12, 1, true>), grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL((igemm_kernel<1, 1, 12, 1, false>), grid, dim3(256), lds, s, p);
}

void launch_igemm_tiled(int lc, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (lc == 0 || lc == 1) launch_igemm_tiled_p0(lc, pre, p, grid, lds, s, ea, eb);
    else if (lc == 2 || lc == 6 || lc == 8) launch_igemm_tiled_p1(lc, pre, p, grid, lds, s, ea, eb);
    else if (lc == 3 || lc == 4) launch_igemm_tiled_p2(lc, pre, p, grid, lds, s, ea, eb);
    else launch_igemm_tiled_p3(lc, pre, p, grid, lds, s, ea, eb);
}
#endif

#define RVC_LG(WM, WN, MF, NF) { if (pre) launch_k(igemm_lds_kernel<WM, WN, MF, NF, true>, p, grid, dim3(256), lds, s, ea, eb); else launch_k(igemm_lds_kernel<WM, WN, MF, NF, false>, p, grid, dim3(256), lds, s, ea, eb); }
#define RVC_LG32(WM, WN, MT, NT) { if (pre) launch_k(igemm32_kernel<WM, WN, MT, NT, true>, p, grid, dim3(256), lds, s, ea, eb); else launch_k(igemm32_kernel<WM, WN, MT, NT, false>, p, grid, dim3(256), lds, s, ea, eb); }
#if RVC_TILED_PART == 0 || defined(RVC_UNITY)
void launch_igemm_tiled_p0(int lc, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb) { if (lc == 0) RVC_LG(2, 2, 4, 4) else RVC_LG(1, 4, 4, 4) }
#endif
#if RVC_TILED_PART == 1 || defined(RVC_UNITY)
void launch_igemm_tiled_p1(int lc, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb) { if (lc == 2) RVC_LG(1, 4, 2, 4) else if (lc == 8) RVC_LG32(2, 2, 1, 1) else RVC_LG(1, 4, 3, 4) }
#endif
#if RVC_TILED_PART == 2 || defined(RVC_UNITY)
void launch_igemm_tiled_p2(int lc, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb) { if (lc == 3) RVC_LG32(2, 2, 2, 2) else RVC_LG32(1, 4, 2, 2) }
#endif
#if RVC_TILED_PART == 3 || defined(RVC_UNITY)
void launch_igemm_tiled_p3(int lc, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb) { if (lc == 7) RVC_LG32(4, 1, 1, 2) else RVC_LG32(1, 4, 1, 2) }
#endif
#undef RVC_LG32
#undef RVC_LG

}  // namespace rvc
