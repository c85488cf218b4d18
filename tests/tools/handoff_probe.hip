// One-way hand-off latency between two workgroups through a tagged 8-byte granule ("the data is the flag"), by protocol and placement.
//   hipcc --offload-arch=gfx950 -O3 handoff_probe.hip -o handoff_probe && ./handoff_probe
// Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }
__device__ __forceinline__ u64 load_nt(const u64 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ u64 load_sc1(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 load_sc0sc1(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void store_plain(u64 *p, u64 v) { *(volatile u64 *)p = v; }
__device__ __forceinline__ void store_sc1(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_nt(u64 *p, u64 v) { __builtin_nontemporal_store(v, p); }

// mode: 0 = store sc1 / load sc1 (agent-scope relaxed atomics: the placement-independent form), 1 = plain store / nt load,
//       2 = plain store / sc1 load, 3 = nt store / nt load, 4 = store sc1 / nt load
// wgA = blockIdx a, wgB = blockIdx b take part; 64 lanes each exchange 64 granules (one 512-byte row)
__global__ __launch_bounds__(64) void pingpong(u64 *slots, int a, int b, int rounds, int mode, u64 *t_out, int *status, unsigned *xcc)
{
    const int me = (int)blockIdx.x == a ? 0 : ((int)blockIdx.x == b ? 1 : -1);
    if (me < 0) return;
    const int lane = threadIdx.x;
    if (lane == 0) xcc[me] = xcc_id();
    u64 *mine = slots + (size_t)me * 4096, *theirs = slots + (size_t)(1 - me) * 4096;
    bool dead = false;
    const u64 t0 = wall_clock64();
    for (int r = 1; r <= rounds; r++) {
        // fresh 512-byte row per round (r & 63): rows are reused every 64 rounds, all accesses to them are cache-bypassing forms
        u64 *w = mine + (size_t)(r & 63) * 64 + lane; const u64 *rd = theirs + (size_t)(r & 63) * 64 + lane;
        const u64 val = ((u64)(unsigned)r << 32) | (unsigned)(lane + r);
        if (me == 0) {
            if (mode == 0 || mode == 4) store_sc1(w, val); else if (mode == 3) store_nt(w, val); else store_plain(w, val);
        }
        u64 x = 0; unsigned spins = 0;
        while (!dead) {
            x = (mode == 0 || mode == 2) ? load_sc1(rd) : load_nt(rd);
            if ((unsigned)(x >> 32) == (unsigned)r) break;
            if (++spins > (1u << 18)) { dead = true; *status = 7; }
        }
        if (!dead && (unsigned)x != (unsigned)(lane + r)) *status = 9;
        if (me == 1) {
            if (mode == 0 || mode == 4) store_sc1(w, val); else if (mode == 3) store_nt(w, val); else store_plain(w, val);
        }
    }
    const u64 t1 = wall_clock64();
    if (lane == 0) t_out[me] = t1 - t0;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    u64 *slots, *t; int *st; unsigned *xcc;
    CHK(hipMalloc(&slots, 2 * 4096 * 8)); CHK(hipMalloc(&t, 16)); CHK(hipMalloc(&st, 4)); CHK(hipMalloc(&xcc, 8));
    const int rounds = 2000;
    for (int place = 0; place < 2; place++)
        for (int mode = 0; mode < 5; mode++) {
            const int a = 0, b = place == 0 ? 8 : 1;        // blockIdx 0 and 8 share an XCD (round-robin dispatch), 0 and 1 do not
            CHK(hipMemset(slots, 0, 2 * 4096 * 8)); CHK(hipMemset(st, 0, 4)); CHK(hipDeviceSynchronize());
            hipLaunchKernelGGL(pingpong, dim3(16), dim3(64), 0, 0, slots, a, b, rounds, mode, t, st, xcc);
            CHK(hipDeviceSynchronize());
            u64 ht[2]; int hs; unsigned hx[2];
            CHK(hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&hs, st, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost));
            printf("%s (XCC %u, %u) mode %d: one-way hand-off %.3f us, status %d\n", place == 0 ? "same XCD" : "other XCD", hx[0], hx[1], mode,
                   (double)(ht[0] > ht[1] ? ht[0] : ht[1]) / 100.0 / rounds / 2.0, hs);
        }
    return 0;
}
