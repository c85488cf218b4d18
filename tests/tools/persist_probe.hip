// persist_probe.hip -- decides whether a PERSISTENT kernel can beat per-layer launches for the one-stream ContentVec chain.
// A chain of 1x1 "layers" with ContentVec's shapes (M x K x N=111: 2304x768, 768x768, 3072x768, 768x3072, repeated) is run
//   A: one launch per layer (the shape of today's engine: ~15 us per layer in the chain),
//   B: ONE launch of G resident workgroups that walk the layers, separated by a grid barrier made of a flag array
//      (every workgroup publishes its epoch, one wave per workgroup polls all G flags), in two coherence variants:
//        B1  plain stores + agent-scope release fence / acquire fence around the barrier (L2 write-back + invalidate)
//        B2  activations written with agent-scope (write-through) stores and read with agent-scope loads: no cache maintenance
//        B3  write-through stores, PLAIN loads (every buffer written once per launch: nothing stale can be cached)
//        B4  the barrier alone (no tiles): its floor
// Same tile code in all variants (16 x 32 tile per workgroup of 8 waves, K split over the waves, LDS reduction), so the
// difference is the boundary.  Prints us per layer and checks B against A.
//   hipcc --offload-arch=gfx950 -O3 persist_probe.hip -o persist_probe && ./persist_probe [G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Layer { const float *w; const float *x; float *y; int M, K, N, ld; };      // w: fragment order [M/16][K/16][64][4]; x: [K][ld]; y: [M][ld]

typedef __attribute__((address_space(1))) float gfloat;          // global address space: pointers read from memory would otherwise be flat
typedef __attribute__((address_space(1))) const float cgfloat;
template <int MODE> __device__ __forceinline__ float ldx(const float *p)
{
    if (MODE == 2) return __hip_atomic_load((cgfloat *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *(cgfloat *)p;          // MODE 3: plain loads are safe when every buffer is written ONCE per launch and never read before (no stale line can exist)
}
template <int MODE> __device__ __forceinline__ void sty(float *p, float v)
{
    if (MODE >= 2) __hip_atomic_store((gfloat *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(gfloat *)p = v;
}

// one 16 x 32 output tile by 8 waves (512 threads): wave w takes chunks [w * nch / 8, ...); partial tiles meet in LDS
template <int MODE>
__device__ __forceinline__ void tile16x32(const Layer &L, int tm, int tn, float *red)
{
    constexpr int KS = 8, D = 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    const int nch = L.K / 16, cpw = (nch + KS - 1) / KS, c0 = wave * cpw, c1 = min(nch, c0 + cpw), nc = max(0, c1 - c0);
    const float *wrow = L.w + ((size_t)tm * nch + c0) * 256 + lane * 4;
    int n0 = tn * 32 + li, n1 = n0 + 16;
    n0 = min(n0, L.N - 1); n1 = min(n1, L.N - 1);
    const float *x0 = L.x + (size_t)(c0 * 16 + kq * 4) * L.ld + n0, *x1 = L.x + (size_t)(c0 * 16 + kq * 4) * L.ld + n1;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    f32x4 a[D]; float b0[D][4], b1[D][4];
#pragma unroll
    for (int s = 0; s < D; s++)
        if (s < nc) {
            a[s] = *(const __attribute__((address_space(1))) f32x4 *)(wrow + s * 256);
#pragma unroll
            for (int j = 0; j < 4; j++) { b0[s][j] = ldx<MODE>(x0 + (size_t)(s * 16 + j) * L.ld); b1[s][j] = ldx<MODE>(x1 + (size_t)(s * 16 + j) * L.ld); }
        }
    for (int c = 0; c < nc; c += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            if (c + s < nc) {
                const f32x4 av = a[s];
                float v0[4], v1[4];
#pragma unroll
                for (int j = 0; j < 4; j++) { v0[j] = b0[s][j]; v1[j] = b1[s][j]; }
                if (c + s + D < nc) {
                    a[s] = *(const __attribute__((address_space(1))) f32x4 *)(wrow + (c + s + D) * 256);
#pragma unroll
                    for (int j = 0; j < 4; j++) { b0[s][j] = ldx<MODE>(x0 + (size_t)((c + s + D) * 16 + j) * L.ld); b1[s][j] = ldx<MODE>(x1 + (size_t)((c + s + D) * 16 + j) * L.ld); }
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], v0[j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], v1[j], acc1, 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { red[(wave * 8 + r) * 64 + lane] = acc0[r]; red[(wave * 8 + 4 + r) * 64 + lane] = acc1[r]; }
    __syncthreads();
    // 512 threads finish the 512 elements: element e = (frag f, reg r, lane l)
    {
        const int e = threadIdx.x, l = e & 63, r = (e >> 6) & 3, f = e >> 8;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < KS; w++) v += red[(w * 8 + f * 4 + r) * 64 + l];
        const int m = tm * 16 + (l >> 4) * 4 + r, n = tn * 32 + f * 16 + (l & 15);
        if (m < L.M && n < L.N) sty<MODE>(L.y + (size_t)m * L.ld + n, tanhf(v * 0.05f));      // a bounded "activation" keeps the chain finite
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void layer_kernel(Layer L)
{
    __shared__ float red[8 * 8 * 64];
    const int ntn = (L.N + 31) / 32;
    const int t = blockIdx.x;
    tile16x32<0>(L, t / ntn, t % ntn, red);
}

template <int MODE>
__global__ __launch_bounds__(512) void persistent_kernel(const Layer *layers, int nlayers, unsigned *flags, unsigned epoch0, int *status, unsigned long long *t_out)
{
    __shared__ float red[8 * 8 * 64];
    const int G = gridDim.x, g = blockIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int l = 0; l < nlayers; l++) {
        const Layer L = layers[l];
        const int ntn = (L.N + 31) / 32, ntiles = (L.M / 16) * ntn;
        if (MODE != 4) for (int t = g; t < ntiles; t += G) tile16x32<MODE>(L, t / ntn, t % ntn, red);
        // ---- grid barrier: publish, then one wave polls every workgroup's flag ----
        const unsigned epoch = epoch0 + (unsigned)l + 1u;
        if (MODE == 1) __threadfence();                                    // release: L2 write-back
        else __builtin_amdgcn_s_waitcnt(0);                                // write-through stores acknowledged
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flags + g, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 64) {
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                for (int i = threadIdx.x; i < G; i += 64) ok = ok && (int)(__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) >= 0;
                if (__all(ok)) break;
                if (++spins > (1u << 22)) { if (threadIdx.x == 0) *status = 7; break; }
            }
        }
        __syncthreads();
        if (MODE == 1) __threadfence();                                    // acquire: invalidate
    }
    if (threadIdx.x == 0) t_out[g] = wall_clock64() - t0;
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int N = 111, ld = 112, reps = 12;
    const int shapes[4][2] = {{2304, 768}, {768, 768}, {3072, 768}, {768, 3072}};      // (M, K): qkv (next reads its first 768 rows), o, ff1, ff2
    const int nl = 4 * reps;
    std::vector<Layer> hl(nl);
    std::vector<float *> bufs;
    float *x0; CHK(hipMalloc(&x0, (size_t)3072 * ld * 4));
    { std::vector<float> h((size_t)3072 * ld); for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f; CHK(hipMemcpy(x0, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
    // two output sets (A and B runs) so that B can be checked against A
    std::vector<float *> yA(nl), yB(nl);
    for (int l = 0; l < nl; l++) {
        const int M = shapes[l % 4][0], K = shapes[l % 4][1];
        float *w; CHK(hipMalloc(&w, (size_t)M * K * 4));
        std::vector<float> hw((size_t)M * K);
        for (size_t i = 0; i < hw.size(); i++) hw[i] = ((float)(((i + l * 7919u) * 2246822519u) % 2001) / 1000.f - 1.f) * (1.0f / sqrtf((float)K));
        CHK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        CHK(hipMalloc(&yA[l], (size_t)M * ld * 4)); CHK(hipMalloc(&yB[l], (size_t)M * ld * 4));
        CHK(hipMemset(yA[l], 0, (size_t)M * ld * 4)); CHK(hipMemset(yB[l], 0, (size_t)M * ld * 4));
        hl[l].w = w; hl[l].M = M; hl[l].K = K; hl[l].N = N; hl[l].ld = ld;
    }
    auto wire = [&](std::vector<float *> &y) { for (int l = 0; l < nl; l++) { hl[l].x = l == 0 ? x0 : y[l - 1]; hl[l].y = y[l]; } };
    hipEvent_t ea, eb; CHK(hipEventCreate(&ea)); CHK(hipEventCreate(&eb));
    // ---- A: per-layer launches ----
    wire(yA);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        CHK(hipEventRecord(ea, 0));
        for (int l = 0; l < nl; l++) hipLaunchKernelGGL(layer_kernel, dim3((hl[l].M / 16) * ((N + 31) / 32)), dim3(512), 0, 0, hl[l]);
        CHK(hipEventRecord(eb, 0)); CHK(hipEventSynchronize(eb));
        float ms; CHK(hipEventElapsedTime(&ms, ea, eb)); if (it >= 2) best = fminf(best, ms);
    }
    printf("A  per-layer launches            : %7.2f us per layer (%d layers)\n", best * 1e3f / nl, nl);
    std::vector<float> refA((size_t)768 * ld);
    CHK(hipMemcpy(refA.data(), yA[nl - 1], refA.size() * 4, hipMemcpyDeviceToHost));
    // ---- B: persistent ----
    wire(yB);
    Layer *dl; CHK(hipMalloc(&dl, nl * sizeof(Layer))); CHK(hipMemcpy(dl, hl.data(), nl * sizeof(Layer), hipMemcpyHostToDevice));
    unsigned *flags; int *st; unsigned long long *tt;
    CHK(hipMalloc(&flags, 1024 * 4)); CHK(hipMemset(flags, 0, 1024 * 4)); CHK(hipMalloc(&st, 4)); CHK(hipMemset(st, 0, 4)); CHK(hipMalloc(&tt, 1024 * 8));
    unsigned epoch = 0;
    std::vector<int> Gs = {128, 224, 256, 448, 512};
    if (argc > 1) Gs = {atoi(argv[1])};
    for (int mode = 2; mode <= 4; mode++)
        for (int G : Gs) {
            float bestb = 1e9f;
            for (int it = 0; it < 5; it++) {
                CHK(hipEventRecord(ea, 0));
                if (mode == 1) hipLaunchKernelGGL(persistent_kernel<1>, dim3(G), dim3(512), 0, 0, dl, nl, flags, epoch, st, tt);
                else if (mode == 2) hipLaunchKernelGGL(persistent_kernel<2>, dim3(G), dim3(512), 0, 0, dl, nl, flags, epoch, st, tt);
                else if (mode == 3) hipLaunchKernelGGL(persistent_kernel<3>, dim3(G), dim3(512), 0, 0, dl, nl, flags, epoch, st, tt);
                else hipLaunchKernelGGL(persistent_kernel<4>, dim3(G), dim3(512), 0, 0, dl, nl, flags, epoch, st, tt);
                if (mode == 3) for (int l = 0; l < nl; l++) CHK(hipMemsetAsync(yB[l], 0, 4, 0));      // (nothing: buffers are rewritten in full)
                CHK(hipEventRecord(eb, 0)); CHK(hipEventSynchronize(eb));
                epoch += (unsigned)nl;
                float ms; CHK(hipEventElapsedTime(&ms, ea, eb)); if (it >= 1) bestb = fminf(bestb, ms);
            }
            int hs = 0; CHK(hipMemcpy(&hs, st, 4, hipMemcpyDeviceToHost));
            std::vector<float> out((size_t)768 * ld);
            CHK(hipMemcpy(out.data(), yB[nl - 1], out.size() * 4, hipMemcpyDeviceToHost));
            double err = 0, ref = 0;
            for (int m = 0; m < 768; m++) for (int n = 0; n < N; n++) { const double d = out[(size_t)m * ld + n] - refA[(size_t)m * ld + n]; err += d * d; ref += (double)refA[(size_t)m * ld + n] * refA[(size_t)m * ld + n]; }
            printf("B%d persistent, G = %3d workgroups : %7.2f us per layer   status %d   rel err vs A %.2e\n", mode, G, bestb * 1e3f / nl, hs, sqrt(err / (ref + 1e-30)));
            CHK(hipMemset(st, 0, 4));
        }
    return 0;
}
