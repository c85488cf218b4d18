// igemm2_inst.hip -- instantiations of igemm2_kernel for ONE tile configuration (-DRVC_IGEMM2_CFG=0..4): compiled five times, in parallel.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "igemm_launch.h"

#ifndef RVC_IGEMM2_CFG
#error "compile with -DRVC_IGEMM2_CFG=<0..4>"
#endif

namespace rvc {

#if RVC_IGEMM2_CFG == 0
#define RVC_FN launch_igemm2_cfg0
#define RVC_MF 1
#define RVC_NF 1
#define RVC_D 12
#elif RVC_IGEMM2_CFG == 1
#define RVC_FN launch_igemm2_cfg1
#define RVC_MF 1
#define RVC_NF 2
#define RVC_D 8
#elif RVC_IGEMM2_CFG == 2
#define RVC_FN launch_igemm2_cfg2
#define RVC_MF 1
#define RVC_NF 4
#define RVC_D 5
#elif RVC_IGEMM2_CFG == 3
#define RVC_FN launch_igemm2_cfg3
#define RVC_MF 2
#define RVC_NF 2
#define RVC_D 6
#else
#define RVC_FN launch_igemm2_cfg4
#define RVC_MF 2
#define RVC_NF 4
#define RVC_D 4
#endif

void RVC_FN(int ks, bool pre, bool lin, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    // register-ring depth per K split: a wave with fewer than D chunks takes the kernel's short path (conditional requests, conservative waits), so the
    // deeper splits get shallower rings -- with 8 waves a 768-deep projection leaves 6 chunks per wave, with 16 waves a 2304-deep one 9
    constexpr int MF = RVC_MF, NF = RVC_NF, D = RVC_D, D16 = D > 8 ? 8 : D, D8 = D > 6 ? 6 : D;
    if (p.ln_wsum) {           // LayerNorm-consumer instantiations (checked at plan time: lin, ks > 1)
        switch (ks) {
        case 4: launch_k(igemm2_kernel<MF, NF, D, 4, false, true, true>, p, grid, dim3(256), lds, s, ea, eb); return;
        case 8: launch_k(igemm2_kernel<MF, NF, D8, 8, false, true, true>, p, grid, dim3(512), lds, s, ea, eb); return;
        default: launch_k(igemm2_kernel<MF, NF, (D16 > 4 ? 4 : D16), 16, false, true, true>, p, grid, dim3(1024), lds, s, ea, eb); return;      // (128 registers at 16 waves: a shallower ring instead of spills)
        }
    }
#define RVC_KS2(PRE, LIN)                                                                                              \
    switch (ks) {                                                                                                      \
    case 1: launch_k(igemm2_kernel<MF, NF, D, 1, PRE, LIN>, p, grid, dim3(256), lds, s, ea, eb); return;               \
    case 4: launch_k(igemm2_kernel<MF, NF, D, 4, PRE, LIN>, p, grid, dim3(256), lds, s, ea, eb); return;               \
    case 8: launch_k(igemm2_kernel<MF, NF, D8, 8, PRE, LIN>, p, grid, dim3(512), lds, s, ea, eb); return;              \
    default: launch_k(igemm2_kernel<MF, NF, D16, 16, PRE, LIN>, p, grid, dim3(1024), lds, s, ea, eb); return;          \
    }
    if (lin) { RVC_KS2(false, true) } else if (pre) { RVC_KS2(true, false) } else { RVC_KS2(false, false) }
#undef RVC_KS2
}

}  // namespace rvc
