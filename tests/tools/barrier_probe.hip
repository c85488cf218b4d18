// Cost of one grid-wide barrier (+ agent-scope fence and a dependent exchange) inside a persistent kernel on MI355X.
// hipcc --offload-arch=gfx950 -O3 barrier_probe.hip -o barrier_probe && ./barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(1024) void probe(unsigned *bar, float *buf, int rounds, int G, int mode, unsigned long long *t_out, int *status)
{
    const int wg = blockIdx.x;
    unsigned long long t0 = wall_clock64();
    float v = (float)wg;
    for (int r = 0; r < rounds; r++) {
        // "work": every workgroup writes one value, after the barrier reads its neighbour's
        if (threadIdx.x == 0) buf[(r & 1) * G + wg] = v;
        if (mode >= 1) __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(r + 1) * (unsigned)G;
            unsigned spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { *status = 7; break; }
            }
        }
        __syncthreads();
        if (mode >= 1) __threadfence();
        if (threadIdx.x == 0) v = 0.5f * v + __builtin_nontemporal_load(&buf[(r & 1) * G + (wg + 1) % G]);
    }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { t_out[wg] = t1 - t0; buf[2 * G + wg] = v; }
}

int main()
{
    for (int G : {16, 48, 96, 256}) for (int threads : {256, 1024}) for (int mode : {0, 1}) {
        unsigned *bar; float *buf; unsigned long long *t; int *st;
        CHK(hipMalloc(&bar, 4)); CHK(hipMalloc(&buf, 3 * G * 4)); CHK(hipMalloc(&t, G * 8)); CHK(hipMalloc(&st, 4));
        CHK(hipMemset(bar, 0, 4)); CHK(hipMemset(st, 0, 4)); CHK(hipMemset(buf, 0, 3 * G * 4));
        const int rounds = 200;
        hipLaunchKernelGGL(probe, dim3(G), dim3(threads), 0, 0, bar, buf, rounds, G, mode, t, st);
        CHK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(G); int hs = 0;
        CHK(hipMemcpy(h.data(), t, G * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hs, st, 4, hipMemcpyDeviceToHost));
        unsigned long long mx = 0; for (auto x : h) mx = x > mx ? x : mx;
        printf("G=%3d threads=%4d fence=%d: %.2f us per round (status %d)\n", G, threads, mode, (double)mx / 100.0 / rounds, hs);
        hipFree(bar); hipFree(buf); hipFree(t); hipFree(st);
    }
    return 0;
}
