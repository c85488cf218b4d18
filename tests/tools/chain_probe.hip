// What does a dependent-kernel boundary cost, and can a device-side flag replace it?
//   hipcc --offload-arch=gfx950 -O3 chain_probe.hip -o chain_probe && ./chain_probe
// A chain of N small "layers" (G workgroups x 256 threads; every output depends on the whole previous vector) launched
//   mode 0: in order on one stream (the engine's form today: barrier bit on every dispatch packet)
//   mode 1: one stream, hipExtAnyOrderLaunch (no barrier bit) + done-counter of the predecessor polled by every workgroup
//   mode 2: alternating between two streams + done-counter
//   modes 3 / 4: as 1 / 2 with write-through (sc1) stores in the producer instead of an L2 write-back fence
// Every spin is bounded; a time-out raises `dead`, which all later waits honour.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct LayerP {
    const float *x; float *y; const float *w; unsigned *done_prev, *done_me; unsigned target; int n, J, wait, sc1; int *dead;
};

__global__ __launch_bounds__(256) void layer(LayerP p)
{
    const int t = threadIdx.x, b = blockIdx.x;
    // weights first: independent of the predecessor (what a real layer can prefetch while it waits)
    float wr[8];
#pragma unroll
    for (int j = 0; j < 8; j++) wr[j] = p.w[((size_t)b * 256 + t) * 8 + j];
    if (p.wait) {
        if (t == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(p.done_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 16) || __hip_atomic_load(p.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(p.dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    float acc = 0.f;
    const int n = p.n;
    for (int j = 0; j < p.J; j++) {
        const int idx = (t + j * 257 + b * 31) & (n - 1);
        acc = fmaf(p.x[idx], wr[j & 7], acc);
    }
    const float v = acc * (1.0f / (float)p.J) + 0.001f * (float)(t & 7);
    if (p.sc1) __hip_atomic_store(&p.y[b * 256 + t], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else p.y[b * 256 + t] = v;
    if (p.wait) {
        __syncthreads();
        if (t == 0) {
            if (!p.sc1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(p.done_me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int N = 200;
    hipStream_t s[2]; CHK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int Gs[] = {32, 64, 128}, Js[] = {4, 32, 256};
    for (int gi = 0; gi < 3; gi++)
        for (int ji = 0; ji < 3; ji++) {
            const int G = Gs[gi], J = Js[ji], n = G * 256;
            float *buf[2], *w; unsigned *done; int *dead;
            CHK(hipMalloc(&buf[0], n * 4)); CHK(hipMalloc(&buf[1], n * 4)); CHK(hipMalloc(&w, (size_t)n * 8 * 4));
            CHK(hipMalloc(&done, (N + 1) * 4)); CHK(hipMalloc(&dead, 4));
            std::vector<float> hx(n), hw((size_t)n * 8), ref(n), out(n);
            for (int i = 0; i < n; i++) hx[i] = (float)((i * 37) % 101) * 0.01f;
            for (size_t i = 0; i < hw.size(); i++) hw[i] = 0.5f + (float)((i * 13) % 17) * 0.03f;
            CHK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
            printf("G=%d J=%d:", G, J);
            for (int mode = 0; mode < 5; mode++) {
                double best = 1e30; int bad = 0, hdead = 0;
                for (int rep = 0; rep < 4; rep++) {
                    CHK(hipMemcpy(buf[0], hx.data(), n * 4, hipMemcpyHostToDevice));
                    CHK(hipMemset(done, 0, (N + 1) * 4)); CHK(hipMemset(dead, 0, 4));
                    // done[0] stands for "the input is there"
                    unsigned g = (unsigned)G; CHK(hipMemcpy(done, &g, 4, hipMemcpyHostToDevice));
                    CHK(hipDeviceSynchronize());
                    CHK(hipEventRecord(e0, s[0]));
                    if (mode == 2 || mode == 4) { CHK(hipStreamWaitEvent(s[1], e0, 0)); }
                    for (int k = 0; k < N; k++) {
                        LayerP p{buf[k & 1], buf[(k + 1) & 1], w, done + k, done + k + 1, (unsigned)G, n, J, mode != 0, mode >= 3, dead};
                        void *args[] = {&p};
                        hipStream_t st = (mode == 2 || mode == 4) ? s[k & 1] : s[0];
                        const int flags = (mode == 1 || mode == 3) ? hipExtAnyOrderLaunch : 0;
                        CHK(hipExtLaunchKernel((const void *)layer, dim3(G), dim3(256), args, 0, st, nullptr, nullptr, flags));
                    }
                    if (mode == 2 || mode == 4) { CHK(hipEventRecord(e1, s[1])); CHK(hipStreamWaitEvent(s[0], e1, 0)); }
                    CHK(hipEventRecord(e1, s[0]));
                    CHK(hipEventSynchronize(e1));
                    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                    CHK(hipMemcpy(out.data(), buf[N & 1], n * 4, hipMemcpyDeviceToHost));
                    CHK(hipMemcpy(&hdead, dead, 4, hipMemcpyDeviceToHost));
                    if (mode == 0) ref = out;
                    else for (int i = 0; i < n; i++) if (out[i] != ref[i]) { bad++; break; }
                }
                printf("  m%d %.2f us/layer%s%s", mode, best * 1000.0 / N, bad ? " MISMATCH" : "", hdead ? " TIMEOUT" : "");
            }
            printf("\n");
            CHK(hipFree(buf[0])); CHK(hipFree(buf[1])); CHK(hipFree(w)); CHK(hipFree(done)); CHK(hipFree(dead));
        }
    return 0;
}
