// XCD-local persistent kernels: what does a barrier between the workgroups of ONE XCD cost when it is kept at that XCD's L2
// (no agent-scope cache maintenance), is the data exchanged through L2 visible, and how fast can one XCD stream weights?
//   hipcc --offload-arch=gfx950 -O3 xcd_probe.hip -o xcd_probe && ./xcd_probe
// Every spin is bounded (status 7 instead of a hang).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }       // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }        // HW_REG_HW_ID

__global__ void where_kernel(unsigned *out)
{
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc_id(); out[2 * blockIdx.x + 1] = hw_id(); }
}

// L2-level (sc0) load / L1 invalidate: the two primitives an XCD-local exchange needs
__device__ __forceinline__ unsigned load_sc0(const unsigned *p)
{
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ float loadf_sc0(const float *p)
{
    float v;
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned l2_read_rmw(unsigned *p)
{
    unsigned v, z = 0;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");     // returning RMW: executes at the L2
    return v;
}
__device__ __forceinline__ void inv_l1() { asm volatile("buffer_inv sc0" ::: "memory"); }

// mode 0: agent-scope release add / acquire spin + threadfence (the portable form)
// mode 1: relaxed agent-scope add + relaxed agent-scope spin, exchange through nontemporal loads
// mode 2: workgroup-scope relaxed add (executes at the L2), spin with sc0 loads, s_waitcnt before the add, exchange read with sc0 load
// mode 3: as 2, exchange read with a plain load after buffer_inv sc0
// mode 4: as 2, spin with an atomic RMW (fetch_max 0) instead of a load
__global__ __launch_bounds__(1024) void bar_kernel(unsigned *bar, float *buf, int rounds, int G, int mode, unsigned long long *t_out, int *status, unsigned *bad, int stride, unsigned *xcc_seen)
{
    if (blockIdx.x % stride) return;
    const int wg = blockIdx.x / stride;
    if (threadIdx.x == 0) atomicOr(xcc_seen, 1u << xcc_id());
    float v = (float)wg;
    unsigned nbad = 0;
    __shared__ int dead;
    if (threadIdx.x == 0) dead = 0;
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; r++) {
        float *slot = buf + (size_t)(mode == 6 ? r : (r & 1)) * G * 64;      // mode 6: a fresh region every round (never read before in this launch)
        if (threadIdx.x < 64) slot[wg * 64 + threadIdx.x] = v + (float)threadIdx.x;          // a cache line per workgroup
        if (mode == 0) __threadfence();
        else __builtin_amdgcn_s_waitcnt(0);                                                   // stores acknowledged by the L2
        __syncthreads();
        if (threadIdx.x == 0 && !dead) {
            const unsigned want = (unsigned)(r + 1) * (unsigned)G;
            unsigned spins = 0;
            if (mode == 0) {
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 16)) { *status = 7; dead = 1; break; } }
            } else if (mode == 1) {
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 16)) { *status = 7; dead = 1; break; } }
            } else if (mode >= 5) {
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (l2_read_rmw(bar) < want) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 16)) { *status = 7; dead = 1; break; } }
            } else if (mode == 4) {
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (__hip_atomic_fetch_max(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 16)) { *status = 7; dead = 1; break; } }
            } else {
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                while (load_sc0(bar) < want) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 16)) { *status = 7; dead = 1; break; } }
            }
        }
        __syncthreads();
        if (mode == 0) __threadfence();
        if (mode == 3) inv_l1();
        const int nb = (wg + 1) % G;
        float got = 0.f;
        if (threadIdx.x < 64) {
            const float *src = slot + nb * 64 + threadIdx.x;
            if (mode == 0 || mode == 3 || mode == 6 || mode == 7) got = *src;
            else if (mode == 1 || mode == 5) got = __builtin_nontemporal_load(src);
            else got = loadf_sc0(src);
        }
        // expected: neighbour's v of this round + lane
        // (every workgroup evolves v identically from its own index: v_r(wg) is recomputed below)
        if (threadIdx.x < 64) {
            float e = (float)nb;
            for (int q = 0; q < r; q++) e = 0.5f * e + 1.0f;
            if (got != e + (float)threadIdx.x) nbad++;
        }
        v = 0.5f * v + 1.0f;
    }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) t_out[wg] = t1 - t0;
    if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(1024) void stream_kernel(const float4 *src, size_t n4, float *sink, int stride)
{
    if (blockIdx.x % stride) return;
    float4 a = make_float4(0, 0, 0, 0);
    for (size_t i = (size_t)(blockIdx.x / stride) * blockDim.x + threadIdx.x; i < n4; i += (size_t)(gridDim.x / stride) * blockDim.x) { const float4 v = src[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (a.x + a.y + a.z + a.w == 12345.678f) sink[0] = a.x;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    // masked streams: ms[0] = mask bits b with b % 8 == 0 (one XCD if the mask is XCD-interleaved), ms[1] = the first ncu/8 bits
    hipStream_t ms[2];
    for (int k = 0; k < 2; k++) {
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int c = 0; c < ncu; c++) if (k == 0 ? (c % 8 == 0) : (c < ncu / 8)) mask[c / 32] |= 1u << (c % 32);
        CHK(hipExtStreamCreateWithCUMask(&ms[k], (uint32_t)mask.size(), mask.data()));
    }
    hipStream_t plain; CHK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    unsigned *d_where; CHK(hipMalloc(&d_where, 2 * 512 * 4));
    for (int which = 0; which < 3; which++) {
        hipStream_t s = which < 2 ? ms[which] : plain;
        const int G = which < 2 ? 64 : 64;
        CHK(hipMemset(d_where, 0xff, 2 * 512 * 4));
        hipLaunchKernelGGL(where_kernel, dim3(G), dim3(256), 0, s, d_where);
        CHK(hipStreamSynchronize(s));
        std::vector<unsigned> h(2 * G); CHK(hipMemcpy(h.data(), d_where, h.size() * 4, hipMemcpyDeviceToHost));
        int cnt[16] = {0}, rr = 0;
        for (int i = 0; i < G; i++) { cnt[h[2 * i] & 15]++; if ((h[2 * i] & 15) == (unsigned)(i % 8)) rr++; }
        printf("[blockIdx %% 8 == XCC_ID for %d of %d] ", rr, G);
        printf("%s: workgroups per XCC_ID:", which == 0 ? "mask bits b%8==0" : (which == 1 ? "mask first ncu/8 bits" : "plain stream"));
        for (int x = 0; x < 16; x++) if (cnt[x]) printf(" [%d]=%d", x, cnt[x]);
        printf("   first hw_ids: %08x %08x %08x %08x\n", h[1], h[3], h[5], h[7]);
    }
    // barrier cost
    for (int where = 0; where < 2; where++)
        for (int G : {8, 16, 32}) for (int threads : {256, 1024}) for (int mode = 0; mode < 8; mode++) {
            if (mode >= 2 && mode <= 4) continue;
            if (where == 1 && mode >= 2) continue;       // L2-local forms are only valid on one XCD
            hipStream_t s = plain; const int stride = where == 0 ? 8 : 1;
            unsigned *xs; CHK(hipMalloc(&xs, 4)); CHK(hipMemset(xs, 0, 4));
            unsigned *bar, *bad; float *buf; unsigned long long *t; int *st;
            CHK(hipMalloc(&bar, 256)); CHK(hipMalloc(&bad, 4)); CHK(hipMalloc(&buf, (size_t)256 * G * 64 * 4)); CHK(hipMalloc(&t, G * 8)); CHK(hipMalloc(&st, 4));
            CHK(hipMemset(bar, 0, 256)); CHK(hipMemset(bad, 0, 4)); CHK(hipMemset(st, 0, 4)); CHK(hipMemset(buf, 0, (size_t)256 * G * 64 * 4));
            CHK(hipDeviceSynchronize());
            const int rounds = 200;
            hipLaunchKernelGGL(bar_kernel, dim3(G * stride), dim3(threads), 0, s, bar, buf, rounds, G, mode, t, st, bad, stride, xs);
            CHK(hipStreamSynchronize(s));
            std::vector<unsigned long long> h(G); int hs = 0; unsigned hb = 0, hx = 0; CHK(hipMemcpy(&hx, xs, 4, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(h.data(), t, G * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hs, st, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            unsigned long long mx = 0; for (auto x : h) mx = x > mx ? x : mx;
            printf("%s G=%2d threads=%4d mode=%d: %.2f us per round, status %d, wrong exchanges %u, XCC mask %02x\n", where == 0 ? "one-XCD" : "whole-chip", G, threads, mode, (double)mx / 100.0 / rounds, hs, hb, hx);
            hipFree(bar); hipFree(bad); hipFree(buf); hipFree(t); hipFree(st);
        }
    // streaming bandwidth of one XCD vs the chip (cold: 512 MB buffer, read 64 MB windows that were not touched recently)
    {
        const size_t total = (size_t)1 << 30;
        float4 *src; float *sink; CHK(hipMalloc(&src, total)); CHK(hipMalloc(&sink, 4));
        CHK(hipMemset(src, 0, total));
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        for (int where = 0; where < 2; where++)
            for (size_t mb : {8, 32, 128}) {
                hipStream_t s = plain; const int stride = where == 0 ? 8 : 1;
                const int G = ncu;
                float best = 1e9f;
                for (int it = 0; it < 4; it++) {
                    const float4 *p = src + ((size_t)it * 192 << 20) / 16;
                    CHK(hipEventRecord(a, s));
                    hipLaunchKernelGGL(stream_kernel, dim3(G * 2), dim3(1024), 0, s, p, (mb << 20) / 16, sink, stride);
                    CHK(hipEventRecord(b, s));
                    CHK(hipStreamSynchronize(s));
                    float ms_; CHK(hipEventElapsedTime(&ms_, a, b));
                    best = ms_ < best ? ms_ : best;
                }
                printf("%s stream %zu MB: %.1f us -> %.2f TB/s\n", where == 0 ? "one-XCD" : "whole-chip", mb, best * 1e3, (double)(mb << 20) / (best * 1e-3) / 1e12);
            }
    }
    return 0;
}
