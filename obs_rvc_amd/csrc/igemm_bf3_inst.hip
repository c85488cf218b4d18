// igemm_bf3_inst.hip -- instantiation of the exploratory split-bf16 GEMM (igemm_bf3_kernel, igemm.hip.h) and of its weight packer.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include "igemm_launch.h"

namespace rvc {

void launch_igemm_bf3(bool lin, bool pre, const IgemmP &p, dim3 grid, size_t lds, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    if (lin) launch_k(igemm_bf3_kernel<2, 2, 2, 2, true, false>, p, grid, dim3(256), lds, s, ea, eb);
    else if (pre) launch_k(igemm_bf3_kernel<2, 2, 2, 2, false, true>, p, grid, dim3(256), lds, s, ea, eb);
    else launch_k(igemm_bf3_kernel<2, 2, 2, 2, false, false>, p, grid, dim3(256), lds, s, ea, eb);
}

// fp32 fragment-major panel [ceil(M / 16)][nchunks][64][4] -> split panels [ceil(M / 32)][nchunks][hi | lo][64][8 bf16]; bytes of the result = M32 * nchunks * 2048
void bf3_pack(const float *wfrag, int M, int nchunks, void *out, hipStream_t s)
{
    const long long total = (long long)((M + 31) / 32) * nchunks * 128;
    hipLaunchKernelGGL(bf3_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wfrag, M, nchunks, reinterpret_cast<bf16x8 *>(out), total);
}

}  // namespace rvc
