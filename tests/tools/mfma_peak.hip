// Tuning aid: what one CU can sustain on v_mfma_f32_32x32x2_f32 with the igemm32 step skeleton added piece by piece.
//   hipcc --offload-arch=gfx950 -O3 tests/tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE bits: 1 B operands from LDS, 2 barrier per step, 4 B staging (8 dword loads -> ds_write), 8 sched_barrier pins loads before / writes
// after the MFMAs, 16 A loads (4 float4), 32 two steps per iteration with ping-pong A registers, 64 ds_write without loads
template <int MODE>
__global__ __launch_bounds__(256) void k(const float *w, const float *x, float *y, int steps)
{
    extern __shared__ float lds[];
    constexpr int RS = 132;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c32 = lane & 31, ks = lane >> 5;
    const long long t0 = clock64(), w0 = wall_clock64();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    for (int i = threadIdx.x; i < 2 * 16 * RS; i += 256) lds[i] = (float)(i & 7);
    __syncthreads();
    f32x4 a[2][2];
    for (int mt = 0; mt < 2; mt++) for (int u = 0; u < 2; u++) a[mt][u] = *reinterpret_cast<const f32x4 *>(w + (mt * 2 + u) * 256 + lane * 4);
    const float *br = lds + (wave & 1) * 64 + c32 + ks * 4 * RS;
    const float *wp = w + (long long)blockIdx.x % 64 * 4096 + lane * 4;
    const float *xp = x + (long long)blockIdx.x % 64 * 4096 + threadIdx.x;
    float sb[8]; f32x4 an[2][2];
    auto step = [&](int c, f32x4 (&ac)[2][2], f32x4 (&nx)[2][2]) {
        const float *bcur = br + (c & 1) * 16 * RS;
        if (MODE & 4) for (int i = 0; i < 8; i++) sb[i] = xp[(c & 31) * 128 + i * 8192];
        if (MODE & 16) for (int mt = 0; mt < 2; mt++) for (int u = 0; u < 2; u++) nx[mt][u] = *reinterpret_cast<const f32x4 *>(wp + (c & 31) * 256 + (mt * 2 + u) * 65536);
        if (MODE & 128) {          // B staging straight into LDS: wave w writes 64 columns of rows (w >> 1) + 2i
            float *bn = lds + ((c + 1) & 1) * 16 * RS;
            for (int i = 0; i < 8; i++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xp + (c & 31) * 128 + i * 8192),
                                                 (__attribute__((address_space(3))) void *)(bn + ((wave >> 1) + i * 2) * RS + (wave & 1) * 64), 4, 0, 0);
        }
        if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            float bv[2][4];
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int j = 0; j < 4; j++) bv[nt][j] = (MODE & 1) ? bcur[(u * 8 + j) * RS + nt * 32] : ac[nt][u][j] + 1.f;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[mt][u][j], bv[nt][j], acc[mt][nt], 0, 0, 0);
        }
        if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
        if (MODE & (4 | 64)) {
            float *bn = lds + ((c + 1) & 1) * 16 * RS;
            for (int i = 0; i < 8; i++) bn[((threadIdx.x >> 7) + i * 2) * RS + (threadIdx.x & 127)] = (MODE & 4) ? sb[i] : ac[0][0][0];
        }
        if (MODE & 128) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the direct-to-LDS loads have landed
        if (MODE & 2) __syncthreads();
    };
    if (MODE & 32) {
        for (int c = 0; c < steps; c += 2) { step(c, a, an); step(c + 1, an, a); }
    } else {
        for (int c = 0; c < steps; c++) {
            step(c, a, an);
            if (MODE & 16) for (int mt = 0; mt < 2; mt++) for (int u = 0; u < 2; u++) a[mt][u] = an[mt][u];
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
    if (s == 12345.678f) y[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { reinterpret_cast<long long *>(y)[0] = clock64() - t0; reinterpret_cast<long long *>(y)[1] = wall_clock64() - w0; }
}

template <int MODE> static int run(const char *name, const float *w, const float *x, float *y, int wgs_per_cu)
{
    const int steps = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const size_t sh = 2 * 16 * 132 * 4;
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), sh, 0, w, x, y, steps);
        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1));
    }
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    long long ck[2]; CHK(hipMemcpy(ck, y, 16, hipMemcpyDeviceToHost));
    const double fl = (double)grid * 4 * steps * 32 * 4096.0;
    printf("%-44s wg/cu=%d  %8.3f ms  %6.1f TF/s  (%.1f%% of 157.3)  shader clock %.0f MHz\n", name, wgs_per_cu, ms, fl / ms * 1e-9, fl / ms * 1e-9 / 157.3 * 100, (double)ck[0] / (double)ck[1] * 100.0);
    return 0;
}
int main()
{
    float *w, *x, *y;
    CHK(hipMalloc(&w, 64 << 20)); CHK(hipMalloc(&x, 64 << 20)); CHK(hipMalloc(&y, 64 << 20));
    CHK(hipMemset(w, 0, 64 << 20)); CHK(hipMemset(x, 0, 64 << 20));
    for (int n = 1; n <= 3; n++) {
        if (run<0>("mfma only", w, x, y, n)) return 1;
        if (run<1>("mfma + B from LDS", w, x, y, n)) return 1;
        if (run<1 + 64>("  + ds_write x8 (no global loads)", w, x, y, n)) return 1;
        if (run<1 + 4>("  + B staging (8 dword loads, ds_write)", w, x, y, n)) return 1;
        if (run<1 + 16>("  + A loads (4 x float4, reg copy)", w, x, y, n)) return 1;
        if (run<1 + 16 + 32>("  + A loads (4 x float4, ping-pong regs)", w, x, y, n)) return 1;
        if (run<1 + 128>("  + B staging by global_load_lds (8 dword)", w, x, y, n)) return 1;
        if (run<1 + 2 + 128 + 16 + 32>("full skeleton (load_lds B, ping-pong A)", w, x, y, n)) return 1;
        if (run<1 + 2 + 4 + 16>("full skeleton (copy)", w, x, y, n)) return 1;
        if (run<1 + 2 + 4 + 16 + 32>("full skeleton (ping-pong)", w, x, y, n)) return 1;
        if (run<1 + 2 + 4 + 16 + 32 + 8>("full skeleton (ping-pong, pinned)", w, x, y, n)) return 1;
    }
    return 0;
}
