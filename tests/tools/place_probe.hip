// Where do the workgroups of one launch land (XCD, SE, CU) and which of them share a CU?  Decides whether a launch can order
// its blocks so that long and short work items pair up on a CU.
//   hipcc --offload-arch=gfx950 -O3 place_probe.hip -o place_probe && ./place_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); }
__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }
__global__ __launch_bounds__(256) void where_kernel(unsigned *out, unsigned long long *t, int hold_us)
{
    extern __shared__ int lds[];
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc_id(); out[2 * blockIdx.x + 1] = hw_id(); t[blockIdx.x] = wall_clock64(); }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)hold_us * 100) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 1000) lds[0] = 1;
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int G : {474, 512, 316}) {
        for (int ldsb : {40 * 1024, 70 * 1024}) {
            unsigned *out; unsigned long long *t;
            CHK(hipMalloc(&out, G * 8)); CHK(hipMalloc(&t, G * 8));
            CHK(hipFuncSetAttribute((const void *)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
            for (int rep = 0; rep < 2; rep++) { where_kernel<<<G, 256, ldsb>>>(out, t, 20); CHK(hipDeviceSynchronize()); }
            std::vector<unsigned> h(2 * G); std::vector<unsigned long long> ht(G);
            CHK(hipMemcpy(h.data(), out, G * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(ht.data(), t, G * 8, hipMemcpyDeviceToHost));
            unsigned long long tmin = ~0ull; for (auto v : ht) tmin = v < tmin ? v : tmin;
            std::map<unsigned, std::vector<int>> cu;      // key: xcc, se, sh, cu
            for (int b = 0; b < G; b++) {
                const unsigned id = h[2 * b + 1];
                const unsigned key = (h[2 * b] << 16) | (((id >> 13) & 7) << 8) | (((id >> 12) & 1) << 4) | ((id >> 8) & 15);
                cu[key].push_back(b);
            }
            int n1 = 0, n2 = 0, n3 = 0, diff256 = 0, late = 0;
            for (auto &kv : cu) {
                if (kv.second.size() == 1) n1++; else if (kv.second.size() == 2) { n2++; if (kv.second[1] - kv.second[0] == 256) diff256++; } else n3++;
            }
            for (int b = 0; b < G; b++) if (ht[b] - tmin > 500) late++;
            printf("G=%d lds=%dK: %zu CUs used; with 1 / 2 / >2 workgroups: %d / %d / %d; pairs (b, b+256): %d; workgroups that started > 5 us late: %d\n", G, ldsb >> 10, cu.size(), n1, n2, n3, diff256, late);
            if (ldsb == 40 * 1024 && G == 474) {
                int shown = 0;
                for (auto &kv : cu) { if (shown++ >= 24) break; printf("   xcc %u se %u sh %u cu %2u:", kv.first >> 16, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 15); for (int b : kv.second) printf(" %d", b); printf("\n"); }
            }
            CHK(hipFree(out)); CHK(hipFree(t));
        }
    }
    return 0;
}
