// Floor of one "step" of the persistent synthesizer front end's hand-off protocol (csrc/synth_front.hip) without any arithmetic:
// G workgroups of 512 threads; per step the 256 stager threads of every workgroup sweep ROWS x 16 tagged granules of the previous
// step's output (agent-scope relaxed atomic loads, NB in flight per thread), write them to LDS, two workgroup barriers, and the first
// ROWS/16 x 2 workgroups publish 256 granules each (agent-scope relaxed atomic stores).  Prints us per step.
//   hipcc --offload-arch=gfx950 -O3 step_floor_probe.hip -o step_floor_probe && ./step_floor_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ u64 gload(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gstore(u64 *p, unsigned tag, float v) { __hip_atomic_store(p, ((u64)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NB>
__global__ __launch_bounds__(512) void steps_kernel(u64 *bufs, int rows, int nsteps, u64 *t_out, int *status, int sleep_n)
{
    extern __shared__ float X[];
    const int g = blockIdx.x, tid = threadIdx.x, nf = g & 1, col0 = nf * 16, mt = g >> 1;
    const int units = rows / 16 * 2;
    const size_t bsz = (size_t)rows * 32;
    bool dead = false;
    const u64 t0 = wall_clock64();
    for (int s = 1; s <= nsteps; s++) {
        const u64 *src = bufs + (size_t)((s - 1) & 63) * bsz;       // step 0's buffer was filled by the host with tag 0 + 1... see main
        u64 *dst = bufs + (size_t)(s & 63) * bsz;
        if (tid < 256) {
            const int total = rows * 16;
            for (int base = tid; base < total; base += 256 * NB) {
                u64 v[NB]; unsigned pend = 0;
#pragma unroll
                for (int i = 0; i < NB; i++) { const int idx = base + i * 256; if (idx < total) { v[i] = gload(src + (size_t)(idx >> 4) * 32 + col0 + (idx & 15)); pend |= 1u << i; } }
                for (unsigned spins = 0; pend;) {
                    unsigned still = 0;
#pragma unroll
                    for (int i = 0; i < NB; i++)
                        if (pend & (1u << i)) {
                            const int idx = base + i * 256;
                            if ((unsigned)(v[i] >> 32) == (unsigned)s) X[(idx >> 4) * 24 + 4 + (idx & 15)] = __uint_as_float((unsigned)v[i]); else still |= 1u << i;
                        }
                    pend = still;
                    if (!pend || dead) break;
                    if (++spins > (1u << 16)) { dead = true; *status = 7; break; }
                    if (sleep_n) __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int i = 0; i < NB; i++) if (pend & (1u << i)) { const int idx = base + i * 256; v[i] = gload(src + (size_t)(idx >> 4) * 32 + col0 + (idx & 15)); }
                }
            }
        }
        __syncthreads();
        float acc = 0.f;
        if (tid >= 256) { for (int k = 0; k < 8; k++) acc += X[((tid - 256) & 15) * 24 + 4 + k]; X[rows * 24 + tid] = acc; }
        __syncthreads();
        if (tid < 256 && g < units) {
            const int l = tid & 63, m = mt * 16 + (l >> 4) * 4 + (tid >> 6), n = col0 + (l & 15);
            gstore(dst + (size_t)m * 32 + n, (unsigned)(s + 1), X[rows * 24 + 256 + (tid & 255)] * 0.5f + 1.0f);
        }
    }
    const u64 t1 = wall_clock64();
    if (tid == 0) t_out[g] = t1 - t0;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    CHK(hipFuncSetAttribute((const void *)steps_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    CHK(hipFuncSetAttribute((const void *)steps_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    for (int rows : {96, 192, 384, 768}) for (int G : {24, 48, 96}) for (int nb : {16, 32}) for (int sl : {0, 1}) {
        if (G < rows / 16 * 2) continue;
        const size_t bsz = (size_t)rows * 32;
        u64 *bufs, *t; int *st;
        CHK(hipMalloc(&bufs, 64 * bsz * 8)); CHK(hipMalloc(&t, 256 * 8)); CHK(hipMalloc(&st, 4));
        CHK(hipMemset(bufs, 0, 64 * bsz * 8)); CHK(hipMemset(st, 0, 4));
        std::vector<u64> h0(bsz);
        for (size_t i = 0; i < bsz; i++) h0[i] = ((u64)1 << 32) | 0x3f800000u;      // step 0's output: tag 1
        CHK(hipMemcpy(bufs, h0.data(), bsz * 8, hipMemcpyHostToDevice));
        const int nsteps = 60;      // < 64 buffers: no reuse inside the launch
        const size_t lds = (size_t)(rows * 24 + 512 + 64) * 4;
        if (nb == 16) hipLaunchKernelGGL(steps_kernel<16>, dim3(G), dim3(512), lds, 0, bufs, rows, nsteps, t, st, sl);
        else hipLaunchKernelGGL(steps_kernel<32>, dim3(G), dim3(512), lds, 0, bufs, rows, nsteps, t, st, sl);
        CHK(hipDeviceSynchronize());
        std::vector<u64> ht(G); int hs = 0;
        CHK(hipMemcpy(ht.data(), t, G * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hs, st, 4, hipMemcpyDeviceToHost));
        u64 mx = 0; for (auto x : ht) mx = x > mx ? x : mx;
        printf("rows %3d (granules per stager thread %2d)  G %2d  in flight %2d  sleep %d: %.2f us per step, status %d\n", rows, rows * 16 / 256, G, nb, sl, (double)mx / 100.0 / nsteps, hs);
        hipFree(bufs); hipFree(t); hipFree(st);
    }
    return 0;
}
