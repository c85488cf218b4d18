// Vector-memory instruction cost on one CU by access pattern (operands resident in L2): what bounds the register-direct GEMM at one stream.
//   hipcc --offload-arch=gfx950 -O3 ta_probe.hip -o ta_probe && ./ta_probe
// pattern 0: dword per lane, 16 lanes contiguous x 4 rows          (the B operand today: x[c][t], 4 loads per MFMA fragment)
// pattern 1: dwordx4 per lane, 4 lanes contiguous (64 B) x 16 rows (the B operand time-major: x[t][c], 1 load per fragment)
// pattern 2: dwordx4 per lane, 64 lanes contiguous (1 KB)          (the A operand: fragment-major weights)
// pattern 3: dwordx2 per lane, 8 lanes contiguous x 8 rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PAT>
__global__ __launch_bounds__(1024) void probe(const float *x, int row_stride, int iters, float *sink, unsigned long long *t_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *base = x + (size_t)blockIdx.x * 65536 + wave * 4096;      // 256 KB per workgroup, 16 KB per wave: L2 resident, L1 misses
    float acc = 0.f;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        const float *b = base + (it & 7) * 64;     // walk along the rows (k direction): fresh columns, same lines re-touched like a K loop
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (PAT == 0) { acc += b[((lane >> 4) + u * 4) * row_stride + (lane & 15)]; }
            else if (PAT == 1) { f32x4 v = *reinterpret_cast<const f32x4 *>(b + ((lane & 15) + (u & 1) * 16) * row_stride + (lane >> 4) * 4 + (u >> 1) * 16); acc += v[0] + v[1] + v[2] + v[3]; }
            else if (PAT == 2) { f32x4 v = *reinterpret_cast<const f32x4 *>(b + u * 256 + lane * 4); acc += v[0] + v[1] + v[2] + v[3]; }
            else { f32x2 v = *reinterpret_cast<const f32x2 *>(b + ((lane >> 3) + u * 8) * row_stride + (lane & 7) * 2); acc += v[0] + v[1]; }
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (acc == 12345.678f) sink[0] = acc;
    if (threadIdx.x == 0) t_out[blockIdx.x] = t1 - t0;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *x, *sink; unsigned long long *t;
    const size_t n = (size_t)256 * 65536 + (1 << 20);
    CHK(hipMalloc(&x, n * 4)); CHK(hipMemset(x, 0, n * 4)); CHK(hipMalloc(&sink, 4)); CHK(hipMalloc(&t, 256 * 8));
    const int iters = 200;
    for (int waves : {4, 8, 16}) {
        for (int pat = 0; pat < 4; pat++) {
            for (int rs : {128, 768}) {
                if (pat == 2 && rs != 128) continue;
                for (int rep = 0; rep < 2; rep++) {
                    if (pat == 0) probe<0><<<256, waves * 64>>>(x, rs, iters, sink, t);
                    else if (pat == 1) probe<1><<<256, waves * 64>>>(x, rs, iters, sink, t);
                    else if (pat == 2) probe<2><<<256, waves * 64>>>(x, rs, iters, sink, t);
                    else probe<3><<<256, waves * 64>>>(x, rs, iters, sink, t);
                    CHK(hipDeviceSynchronize());
                }
                std::vector<unsigned long long> h(256);
                CHK(hipMemcpy(h.data(), t, 256 * 8, hipMemcpyDeviceToHost));
                double s = 0; for (auto v : h) s += (double)v; s /= 256;        // 100 MHz ticks
                const double ns = s * 10.0, per_instr_cu = ns / ((double)iters * 8 * waves);
                const int bytes = pat == 0 ? 256 : (pat == 3 ? 512 : 1024);
                printf("waves/CU %2d  pattern %d  row stride %4d floats: %.1f ns per load instruction per CU (%.1f clk at 2.4 GHz), %.1f B/clk\n",
                       waves, pat, rs, per_instr_cu, per_instr_cu * 2.4, bytes / (per_instr_cu * 2.4));
            }
        }
    }
    return 0;
}
