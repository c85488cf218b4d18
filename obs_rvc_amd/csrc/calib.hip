// calib.hip -- what THIS box can do, measured in-run (row d of SURVEY.md section 8: measurement; no counterpart in the reference, which only prints
// `Instant` deltas, rvc/src/rvc.rs:217).  MI355X boxes of one pool differ by ~10 % on the matrix-core-bound legs of the same binary (round 5: 19.44
// vs 21.78 ms per 32-stream step); the chip clocks to its power budget, so a roofline fraction against the nominal 2.4 GHz peak mixes kernel
// quality with the box.  Two instruments, both behind the C ABI so that bench.py (and any host) can call them:
//   rvc_calibrate            a ~50 ms calibration: a bare v_mfma_f32_32x32x2_f32 stream on every SIMD (non-zero operands: zero-filled inputs clock
//                            higher) -> achieved fp32-MFMA TF/s and the shader clock it ran at; a float4 read stream over 1 GiB -> HBM TB/s
//   rvc_clock_monitor_start  eight sleeping waves (one per XCD) that count shader cycles (s_memtime) against the 100 MHz real-time counter
//   / _stop                  (s_memrealtime) WHILE the engine works: the effective shader clock of a leg under its real load
#include "../../include/rvc_mi355x.h"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

namespace rvc {
namespace {

typedef float f32x16_c __attribute__((ext_vector_type(16)));

// bare MFMA stream: four independent accumulators per wave, operands differ per lane and per accumulator (power depends on the data)
__global__ __launch_bounds__(256) void calib_mfma_kernel(float *sink, long long *clk, int iters)
{
    const int lane = threadIdx.x & 63;
    const long long c0 = clock64(), r0 = wall_clock64();
    f32x16_c a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float va = 1.0f + lane * 0.0137f, vb = 0.731f - lane * 0.0071f, vc = -0.37f + lane * 0.0213f, vd = 1.91f - lane * 0.0049f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vc, vd, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb, vc, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(vd, va, a3, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) s += a0[r] + a1[r] + a2[r] + a3[r];
    if (s == 1.2345e37f) sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - r0; }
}

// read stream: every thread sums float4s of its stripe; nothing is written but one partial per workgroup that cannot be proven dead
__global__ __launch_bounds__(256) void calib_read_kernel(const float4 *x, size_t n4, float *sink, long long *clk)
{
    const long long c0 = clock64(), r0 = wall_clock64();
    float4 s = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
        s.x += a.x + b.x + c.x + d.x; s.y += a.y + b.y + c.y + d.y; s.z += a.z + b.z + c.z + d.z; s.w += a.w + b.w + c.w + d.w;
    }
    for (; i < n4; i += stride) { const float4 a = x[i]; s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
    const float t = s.x + s.y + s.z + s.w;
    if (t == 1.2345e37f) sink[blockIdx.x * 256 + threadIdx.x] = t;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = clock64() - c0; clk[3] = wall_clock64() - r0; }
}

__global__ void calib_fill_kernel(float4 *x, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float v = (float)((i * 2654435761u) & 0xffff) * (1.0f / 65536.0f) - 0.5f;
        x[i] = float4{v, -v, v * 0.5f, 1.0f - v};
    }
}

// clock monitor: workgroup b (one wave) sleeps and samples until the host raises *stop or `limit` real-time ticks (100 MHz) have passed;
// out[b] = {shader cycles, real-time ticks, XCC id}
__global__ __launch_bounds__(64) void clock_monitor_kernel(const volatile int *stop, long long *out, long long limit)
{
    if (threadIdx.x != 0) return;
    const long long c0 = clock64(), r0 = wall_clock64();
    long long c1 = c0, r1 = r0;
    while (true) {
        __builtin_amdgcn_s_sleep(127);
        c1 = clock64(); r1 = wall_clock64();
        if (r1 - r0 > limit) break;
        if (__hip_atomic_load(const_cast<const int *>(stop), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
    }
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x * 3 + 0] = c1 - c0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (long long)(xcc & 0xf);
}

struct Monitor {
    hipStream_t stream = nullptr;
    int *h_stop = nullptr;            // host-mapped
    long long *d_out = nullptr;
    bool running = false;
};
std::mutex g_mon_mu;
std::map<int, Monitor> g_mon;

}  // namespace
}  // namespace rvc

using namespace rvc;

extern "C" {

rvc_status rvc_calibrate(int device, rvc_calibration *out)
{
    if (!out) return RVC_SHAPE;
    memset(out, 0, sizeof(*out));
    try {
        if (device < 0) HIPCHK(hipGetDevice(&device));
        HIPCHK(hipSetDevice(device));
        hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, device));
        const int ncu = prop.multiProcessorCount;
        float *sink = nullptr; long long *clk = nullptr; float4 *buf = nullptr;
        const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
        HIPCHK(hipMalloc(&sink, (size_t)ncu * 4 * 256 * sizeof(float)));
        HIPCHK(hipMalloc(&clk, 4 * sizeof(long long)));
        HIPCHK(hipMalloc(&buf, bytes));
        hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        const auto t_begin = std::chrono::steady_clock::now();
        // fp32 MFMA: 4 workgroups of 4 waves per CU = 4 waves per SIMD; 16 MFMAs per iteration and wave; ~20 ms per launch, the third launch counts
        const int iters = 12000, grid = ncu * 4;
        float ms = 0.f;
        for (int rep = 0; rep < 3; rep++) {
            HIPCHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(calib_mfma_kernel, dim3(grid), dim3(256), 0, 0, sink, clk, iters);
            HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        }
        long long h[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpy(h, clk, 2 * sizeof(long long), hipMemcpyDeviceToHost));
        const double flops = (double)grid * 4 * (double)iters * 16 * 4096.0;       // 32 x 32 x 2 x 2 flops per MFMA
        out->mfma_f32_tflops = flops / (ms * 1e-3) / 1e12;
        out->mfma_ms = ms;
        out->mfma_sclk_mhz = h[1] > 0 ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
        // HBM read: 1 GiB (four times the 256 MB memory-side cache), filled with non-trivial data; best of 6 passes
        hipLaunchKernelGGL(calib_fill_kernel, dim3(ncu * 8), dim3(256), 0, 0, buf, n4);
        double best = 0.0;
        for (int rep = 0; rep < 6; rep++) {
            HIPCHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(calib_read_kernel, dim3(ncu * 8), dim3(256), 0, 0, (const float4 *)buf, n4, sink, clk);
            HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            best = std::max(best, (double)bytes / (ms * 1e-3) / 1e12);
        }
        HIPCHK(hipMemcpy(h, clk, 4 * sizeof(long long), hipMemcpyDeviceToHost));
        out->hbm_read_tbs = best;
        out->hbm_sclk_mhz = h[3] > 0 ? (double)h[2] / (double)h[3] * 100.0 : 0.0;
        out->compute_units = ncu;
        out->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        (void)hipFree(buf); (void)hipFree(clk); (void)hipFree(sink);
        return RVC_OK;
    } catch (const std::exception &) { return RVC_BACKEND; }
}

rvc_status rvc_clock_monitor_start(int device)
{
    try {
        if (device < 0) HIPCHK(hipGetDevice(&device));
        HIPCHK(hipSetDevice(device));
        std::lock_guard<std::mutex> lk(g_mon_mu);
        Monitor &m = g_mon[device];
        if (m.running) return RVC_BACKEND;
        if (!m.stream) {
            HIPCHK(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
            HIPCHK(hipHostMalloc((void **)&m.h_stop, sizeof(int), hipHostMallocMapped));
            HIPCHK(hipMalloc(&m.d_out, 8 * 3 * sizeof(long long)));
        }
        *m.h_stop = 0;
        int *d_stop = nullptr;
        HIPCHK(hipHostGetDevicePointer((void **)&d_stop, m.h_stop, 0));
        HIPCHK(hipMemsetAsync(m.d_out, 0, 8 * 3 * sizeof(long long), m.stream));
        // bounded: the waves leave by themselves after 20 s of real time whatever the host does
        hipLaunchKernelGGL(clock_monitor_kernel, dim3(8), dim3(64), 0, m.stream, (const volatile int *)d_stop, m.d_out, (long long)20 * 100000000LL);
        HIPCHK(hipGetLastError());
        m.running = true;
        return RVC_OK;
    } catch (const std::exception &) { return RVC_BACKEND; }
}

rvc_status rvc_clock_monitor_stop(int device, double *sclk_mhz_mean, double *sclk_mhz_min, double *seconds)
{
    try {
        if (device < 0) HIPCHK(hipGetDevice(&device));
        HIPCHK(hipSetDevice(device));
        std::lock_guard<std::mutex> lk(g_mon_mu);
        auto it = g_mon.find(device);
        if (it == g_mon.end() || !it->second.running) return RVC_BACKEND;
        Monitor &m = it->second;
        __atomic_store_n(m.h_stop, 1, __ATOMIC_SEQ_CST);
        HIPCHK(hipStreamSynchronize(m.stream));
        m.running = false;
        long long h[24];
        HIPCHK(hipMemcpy(h, m.d_out, sizeof h, hipMemcpyDeviceToHost));
        double sum = 0.0, mn = 1e30, secs = 0.0; int n = 0;
        for (int b = 0; b < 8; b++) {
            if (h[b * 3 + 1] <= 0) continue;
            const double mhz = (double)h[b * 3] / (double)h[b * 3 + 1] * 100.0;
            sum += mhz; mn = std::min(mn, mhz); n++;
            secs = std::max(secs, (double)h[b * 3 + 1] * 1e-8);
        }
        if (sclk_mhz_mean) *sclk_mhz_mean = n ? sum / n : 0.0;
        if (sclk_mhz_min) *sclk_mhz_min = n ? mn : 0.0;
        if (seconds) *seconds = secs;
        return RVC_OK;
    } catch (const std::exception &) { return RVC_BACKEND; }
}

}  // extern "C"
