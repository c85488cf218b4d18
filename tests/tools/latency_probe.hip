// Tuning aid: shader clock under a light load and dependent-load latency (L2 hit / HBM), single lane.
// Build: hipcc --offload-arch=gfx950 -O2 -Wno-unused-value tests/tools/latency_probe.hip -o /tmp/lp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
__global__ void chase(const int *next, int start, int steps, long long *out, int *sink)
{
    int p = start;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < steps; i++) p = next[p];
    long long c1 = clock64(), w1 = wall_clock64();
    out[0] = c1 - c0; out[1] = w1 - w0; *sink = p;
}
__global__ void spin(long long *out, int iters)
{
    long long c0 = clock64(), w0 = wall_clock64();
    float a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    if (a == 12345.f) out[2] = 1;
}
__global__ void empty_k(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
int main()
{
    long long *d_out; int *d_sink; hipMalloc(&d_out, 64); hipMalloc(&d_sink, 4);
    long long h[3];
    for (size_t n : {size_t(1) << 12, size_t(1) << 20, size_t(1) << 27}) {   // 16 KB, 4 MB, 512 MB of ints
        std::vector<int> nx(n);
        size_t stride = 4099 % n; if (stride == 0) stride = 1;
        for (size_t i = 0; i < n; i++) nx[i] = (int)((i + stride * 33) % n);
        int *d; hipMalloc(&d, n * 4); hipMemcpy(d, nx.data(), n * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d, 0, 2000, d_out, d_sink);
            hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        }
        double ns = (double)h[1] * 10.0 / 2000.0;   // wall_clock64 ticks at 100 MHz
        printf("chase %6.1f MB: %.0f shader cycles/load, %.0f ns/load, shader clock %.0f MHz\n", n * 4 / 1e6, (double)h[0] / 2000.0, ns, (double)h[0] / ((double)h[1] * 10.0) * 1e3);
        hipFree(d);
    }
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, d_out, 200000);
    hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
    printf("spin (1 wave): shader clock %.0f MHz\n", (double)h[0] / ((double)h[1] * 10.0) * 1e3);
    hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, 0, d_out, 200000);
    hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
    printf("spin (full chip): shader clock %.0f MHz\n", (double)h[0] / ((double)h[1] * 10.0) * 1e3);
    // back-to-back empty kernels: boundary cost
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, nullptr);
    hipEventRecord(a, s);
    for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, nullptr);
    hipEventRecord(b, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("1000 empty kernels eager: %.2f us each\n", ms);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, nullptr);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, a, b);
    printf("1000 empty kernels in a graph: %.2f us each\n", ms);
    return 0;
}
