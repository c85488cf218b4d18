// kernels.hip.h -- hand-written gfx950 (CDNA4) kernels for the RVC per-chunk hot path (everything except the implicit-GEMM
// templates, which live in igemm.hip.h).  Layout convention and the GEMM formulation: see igemm.hip.h.
#pragma once
#include "igemm.hip.h"
#include "state.hip.h"

namespace rvc {

// second stage of a split-K launch: fixed-order (deterministic) sum of the partials + epilogue
static __global__ __launch_bounds__(256) void splitk_epilogue_kernel(IgemmP p)
{
    const int total = p.M * p.N;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int z = blockIdx.y;
    const int phase = z % p.nphase, b = z / p.nphase;
    const PhaseD ph = p.ph[phase];
    const float *pp = p.part + (long long)(b * p.nphase + phase) * p.ksplit * total + i;
    float acc = 0.f;
    for (int ks = 0; ks < p.ksplit; ks++) acc += pp[(long long)ks * total];
    const int m = i / p.N, n = i - m * p.N;
    epilogue_store(p, ph, b, m, n, acc);
}

// ------------------------------------------------------------------------------------
// wave / block reductions (64-wide wavefronts)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// blockDim.x must be a multiple of 64 and <= 1024; red must hold 16 floats
__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; i++) t += red[i];
    return t;
}

// ------------------------------------------------------------------------------------
// RMVPE front end: reflect pad + periodic Hann + 1024-pt FFT + magnitude + mel + log
// (reference: rvc/src/f0/rmvpe.rs:80-116, 159-205).  One workgroup per frame, everything
// LDS-resident; the mel reduction is a wavefront shuffle reduction.
// Output goes straight into the RMVPE input image [1][Tm(+halo)][128(+halo)] with the
// network's input BatchNorm affine applied; the raw log-mel is kept for taps.
// ------------------------------------------------------------------------------------
struct MelP {
    const float *audio;     // [B][n] 16 kHz input (device)
    long long audio_bs;
    int n;                  // samples per stream
    int frame;              // f0_extractor_frame: the last `frame` samples are analysed
    int Tm;
    const float *window;    // [1024]
    const float *twiddle;   // [512][2] cos,sin of -2*pi*j/1024
    const float *basis;     // [128][513]
    const int *band;        // [128][2] first / one-past-last non-zero bin of each mel filter
    float *mel;             // [B][128][Tm] raw log-mel (tap / parity)
    float *img;             // RMVPE input image interior pointer
    long long img_bs; int img_ld;
    float bn_scale, bn_shift;
};

// One workgroup = one frame.  rmvpe.rs:159-205 restated for the GPU: reflect pad + periodic Hann while loading, a 1024-point
// complex FFT as five radix-4 Stockham passes (every thread owns one 4-point butterfly per pass; ping-pong in LDS, five barriers
// instead of the ten of a radix-2 pass structure), magnitudes of the 513 kept bins, then the mel projection as WAVEFRONT-SHUFFLE
// reductions: a 16-lane group per filter strides over the filter's non-zero band and folds its partial sums with four xor-shuffles
// (four filters per wave at a time, 128 filters over the four waves), log, and the RMVPE input affine.
__device__ __forceinline__ void tw1024(const float *tw, int e, float &c, float &s)
{
    // W^e = exp(-2 pi i e / 1024) from the half table (e < 512); W^(e + 512) = -W^e
    const int h = e & 511;
    c = tw[2 * h]; s = tw[2 * h + 1];
    if (e & 512) { c = -c; s = -s; }
}
static __global__ __launch_bounds__(256) void mel_frontend_kernel(MelP p)
{
    __shared__ float bufr[2][1024], bufi[2][1024];
    __shared__ float mag[516];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float *sig = p.audio + (long long)b * p.audio_bs + (p.n - p.frame);
    const int L = p.frame;
    const int lane = tid & 63, wave = tid >> 6, sub = lane >> 4, l16 = lane & 15;
    // Round 6: EVERY table value this thread will need is requested here, next to the signal -- its twelve twiddles (they depend on tid only), the bands of its eight
    // mel filters and, behind those, the first four basis values per filter.  The tables are cold at every chunk (853 MB of weights pass between two uses) and
    // were read where they were needed: a memory round trip in front of each of four FFT passes and two per mel filter group -- most of the kernel's 20 us, which is
    // the head of the f0 branch, the critical path of the chunk's front.  Same arithmetic in the same order.
    float sg[4], wn[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int j = tid + r * 256;
        int q = t * 160 + j - 512;            // index into the unpadded signal
        if (q < 0) q = -q;                     // left reflect: padded[512-i-1] = sig[i+1]
        if (q >= L) q = 2 * L - 2 - q;         // right reflect: padded[L+512+i] = sig[L-i-2]
        sg[r] = sig[q]; wn[r] = p.window[j];
    }
    float twc[4][3], tws[4][3];                // passes Ns = 4, 16, 64, 256; r = 1..3
    {
        int pi = 0;
#pragma unroll
        for (int Ns = 4; Ns < 1024; Ns *= 4, pi++) {
            const int estep = (tid & (Ns - 1)) * (256 / Ns);
#pragma unroll
            for (int r = 1; r < 4; r++) tw1024(p.twiddle, estep * r, twc[pi][r - 1], tws[pi][r - 1]);
        }
    }
    int blo[8], bhi[8];
#pragma unroll
    for (int it = 0; it < 8; it++) { const int m = wave * 32 + it * 4 + sub; blo[it] = p.band[2 * m]; bhi[it] = p.band[2 * m + 1]; }
    float bpre[8][4];
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const float *br = p.basis + (wave * 32 + it * 4 + sub) * 513;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int k = blo[it] + l16 + 16 * j; bpre[it][j] = k < bhi[it] ? br[k] : 0.f; }
    }
    // frame t covers padded[t*160 .. t*160+1024), padded = reflect(sig, 512); natural order (the Stockham passes sort as they go)
#pragma unroll
    for (int r = 0; r < 4; r++) { const int j = tid + r * 256; bufr[0][j] = sg[r] * wn[r]; bufi[0][j] = 0.f; }
    __syncthreads();
    int cur = 0;
    int pass = -1;
#pragma unroll
    for (int Ns = 1; Ns < 1024; Ns *= 4, pass++) {
        const int k = tid & (Ns - 1);
        float vr[4], vi[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float xr = bufr[cur][tid + r * 256], xi = bufi[cur][tid + r * 256];
            if (r == 0 || Ns == 1) { vr[r] = xr; vi[r] = xi; }
            else { const float c = twc[Ns == 1 ? 0 : pass][r - 1], sn = tws[Ns == 1 ? 0 : pass][r - 1]; vr[r] = xr * c - xi * sn; vi[r] = xr * sn + xi * c; }
        }
        const float a0r = vr[0] + vr[2], a0i = vi[0] + vi[2], a1r = vr[0] - vr[2], a1i = vi[0] - vi[2];
        const float a2r = vr[1] + vr[3], a2i = vi[1] + vi[3];
        const float a3r = vi[1] - vi[3], a3i = -(vr[1] - vr[3]);     // (v1 - v3) * (-i)
        const int j0 = (tid / Ns) * Ns * 4 + k;
        float *orr = bufr[cur ^ 1], *oi = bufi[cur ^ 1];
        orr[j0] = a0r + a2r;          oi[j0] = a0i + a2i;
        orr[j0 + Ns] = a1r + a3r;     oi[j0 + Ns] = a1i + a3i;
        orr[j0 + 2 * Ns] = a0r - a2r; oi[j0 + 2 * Ns] = a0i - a2i;
        orr[j0 + 3 * Ns] = a1r - a3r; oi[j0 + 3 * Ns] = a1i - a3i;
        cur ^= 1;
        __syncthreads();
    }
    for (int k = tid; k < 513; k += 256) { const float xr = bufr[cur][k], xi = bufi[cur][k]; mag[k] = sqrtf(xr * xr + xi * xi); }
    __syncthreads();
    // mel projection: filter m = wave * 32 + it * 4 + (lane >> 4); its 16 lanes stride over the band [lo, hi), xor-shuffle fold
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int m = wave * 32 + it * 4 + sub;
        const int lo = blo[it], hi = bhi[it];
        const float *br = p.basis + m * 513;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int k = lo + l16 + 16 * j; if (k < hi) s += bpre[it][j] * mag[k]; }
        for (int k = lo + l16 + 64; k < hi; k += 16) s += br[k] * mag[k];          // (bands wider than 64 bins: none at 16 kHz / 1024 points)
        s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
        if (l16 == 0) {
            const float lm = logf(fmaxf(s, 1e-5f));
            p.mel[((long long)b * 128 + m) * p.Tm + t] = lm;
            p.img[(long long)b * p.img_bs + (long long)t * p.img_ld + m] = lm * p.bn_scale + p.bn_shift;
        }
    }
}

// ------------------------------------------------------------------------------------
// normalisation kernels
// ------------------------------------------------------------------------------------
// LayerNorm over channels of x[B][C][ld] for each time step (eps 1e-5), optional in-place.
// block = 4 time steps x 64 channel lanes (so T = 111 already spreads over 28 workgroups); each thread keeps
// its C/64 values in registers: one global read pass, two-pass mean/variance as in the reference definition.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_ct_kernel(const float *x, float *y, const float *g, const float *bta,
                                                           int C, int T, int x_cs, long long x_bs, int y_cs, long long y_bs)
{
    __shared__ float red[4][4];
    const int tx = threadIdx.x & 3, ty = threadIdx.x >> 2;      // ty = 0..63
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + tx, b = blockIdx.y;
    const bool ok = t < T;
    const float *xp = x + (long long)b * x_bs + (ok ? t : 0);
    float v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = ty + i * 64;
        v[i] = (ok && c < C) ? xp[(long long)c * x_cs] : 0.f;
        s += v[i];
    }
    float gv[NV], bv[NV];       // loaded now, consumed after the two reductions
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = ty + i * 64;
        gv[i] = c < C ? g[c] : 0.f; bv[i] = c < C ? bta[c] : 0.f;
    }
    // reduce over the 16 lanes of this wave that share tx (lane bits 2..5), then over the 4 waves through LDS
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
    if (lane < 4) red[wave][lane] = s;
    __syncthreads();
    const float mean = (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]) / (float)C;
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const int c = ty + i * 64;
        const float d = (c < C) ? v[i] - mean : 0.f;
        q += d * d;
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) q += __shfl_xor(q, o, 64);
    if (lane < 4) red[wave][lane] = q;
    __syncthreads();
    const float var = (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]) / (float)C;
    const float inv = 1.0f / sqrtf(var + 1e-5f);
    if (ok) {
        float *yp = y + (long long)b * y_bs + t;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            const int c = ty + i * 64;
            if (c < C) yp[(long long)c * y_cs] = (v[i] - mean) * inv * gv[i] + bv[i];
        }
    }
}

// Throughput-mode LayerNorm (many streams): a workgroup owns 32 time steps x ALL channels of one stream.  Rows are read and written as
// full 128-byte lines (the 4-step kernel above touches 16-byte slivers of lines that other workgroups -- on other XCDs -- fetch again:
// 156 MB of HBM/MALL reads per launch for 22 MB of data at 64 streams); the tile sits in LDS ([C][33]) for the two-pass statistics.
static __global__ __launch_bounds__(256) void layernorm_tile_kernel(const float *x, float *y, const float *g, const float *bta,
                                                             int C, int T, int x_cs, long long x_bs, int y_cs, long long y_bs)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [C][33]
    __shared__ float red[8][32], s_mean[32], s_inv[32];
    const int t0 = blockIdx.x * 32, b = blockIdx.y, tid = threadIdx.x, tx = tid & 31, part = tid >> 5;
    const float *xb = x + (long long)b * x_bs + t0;
    const bool ok = t0 + tx < T;
    for (int c0 = part; c0 < C; c0 += 8 * 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int c = c0 + 8 * u; v[u] = (ok && c < C) ? xb[(long long)c * x_cs + tx] : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int c = c0 + 8 * u; if (c < C) tile[c * 33 + tx] = v[u]; }
    }
    __syncthreads();
    float s = 0.f;
    for (int c = part; c < C; c += 8) s += tile[c * 33 + tx];
    red[part][tx] = s;
    __syncthreads();
    if (part == 0) { float m = 0.f; for (int q = 0; q < 8; q++) m += red[q][tx]; s_mean[tx] = m / (float)C; }
    __syncthreads();
    const float mean = s_mean[tx];
    float qv = 0.f;
    for (int c = part; c < C; c += 8) { const float d = tile[c * 33 + tx] - mean; qv += d * d; }
    red[part][tx] = qv;
    __syncthreads();
    if (part == 0) { float m = 0.f; for (int q = 0; q < 8; q++) m += red[q][tx]; s_inv[tx] = 1.0f / sqrtf(m / (float)C + 1e-5f); }
    __syncthreads();
    if (!ok) return;
    const float inv = s_inv[tx];
    float *yb = y + (long long)b * y_bs + t0;
    for (int c = part; c < C; c += 8) yb[(long long)c * y_cs + tx] = (tile[c * 33 + tx] - mean) * inv * g[c] + bta[c];
}

// Many streams: LayerNorm over channels on 16-column strips, registers only.  A workgroup owns 16 consecutive time steps of one stream
// (one 64-byte segment of every channel row): thread (rg = tid / 4, quad = tid % 4) loads the float4 of rows rg, rg + 64, ... up front
// (NR independent 16-byte loads in flight per thread, no LDS staging), the per-column sums over the 64 row groups go through one
// small LDS exchange, mean and variance are taken from the values still in registers (two-pass, as the reference definition), and
// the strip is written back as float4 (the ragged last quad of a row element-wise, so a halo behind it stays zero).
// The round-1 tile kernel staged [C][32] in 100 KB of LDS: one workgroup per CU, 54 us for 22 MB in + 22 MB out at 64 streams.
template <int NR>
__global__ __launch_bounds__(256) void layernorm_strip_kernel(const float *x, float *y, const float *g, const float *bta,
                                                              int C, int T, int x_cs, long long x_bs, int y_cs, long long y_bs)
{
    __shared__ float red[64][17];
    __shared__ float s_stat[2][16];
    const int t0 = blockIdx.y * 16, b = blockIdx.x, tid = threadIdx.x, quad = tid & 3, rg = tid >> 2;      // (grid: x = stream, y = strip)
    const float *xb = x + (long long)b * x_bs + t0 + quad * 4;
    f32x4 v[NR];
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int c = rg + r * 64;
        v[r] = c < C ? *reinterpret_cast<const f32x4 *>(xb + (long long)c * x_cs) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 sm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NR; r++) sm += v[r];
#pragma unroll
    for (int j = 0; j < 4; j++) red[rg][quad * 4 + j] = sm[j];
    __syncthreads();
    if (tid < 16) { float m = 0.f; for (int q = 0; q < 64; q++) m += red[q][tid]; s_stat[0][tid] = m / (float)C; }
    __syncthreads();
    f32x4 mean;
#pragma unroll
    for (int j = 0; j < 4; j++) mean[j] = s_stat[0][quad * 4 + j];
    f32x4 qv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NR; r++) { if (rg + r * 64 < C) { const f32x4 d = v[r] - mean; qv += d * d; } }
#pragma unroll
    for (int j = 0; j < 4; j++) red[rg][quad * 4 + j] = qv[j];
    __syncthreads();
    if (tid < 16) { float m = 0.f; for (int q = 0; q < 64; q++) m += red[q][tid]; s_stat[1][tid] = 1.0f / sqrtf(m / (float)C + 1e-5f); }
    __syncthreads();
    f32x4 inv;
#pragma unroll
    for (int j = 0; j < 4; j++) inv[j] = s_stat[1][quad * 4 + j];
    float *yb = y + (long long)b * y_bs + t0 + quad * 4;
    const int tq = t0 + quad * 4;
    if (tq >= T) return;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        const int c = rg + r * 64;
        if (c >= C) break;
        const float gg = g[c], bb = bta[c];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = (v[r][j] - mean[j]) * inv[j] * gg + bb;
        float *dst = yb + (long long)c * y_cs;
        if (tq + 4 <= T) *reinterpret_cast<f32x4 *>(dst) = o;
        else { for (int j = 0; j < 4 && tq + j < T; j++) dst[j] = o[j]; }
    }
}


// ContentVec's first layer in one kernel: Conv1d(1 -> C, k taps, stride st, no bias) + GroupNorm(C groups = per channel over time) +
// GELU.  One workgroup per (channel, stream): the k weights live in registers, every thread computes its outputs from the raw 16 kHz
// ring (k fused multiply-adds each, in tap order -- the same f32 chain the matrix core runs) and KEEPS them in registers, the mean /
// variance go through two block reductions (two-pass, as the reference definition), and the normalised, activated row is written
// once.  HBM traffic = one write of the [C][T] row per stream instead of write + read + read + write (at 64 streams: 0.94 GB instead
// of 3.8 GB), and no implicit-GEMM launch with K = 16 for a 10-tap filter.
template <int NT>
__global__ __launch_bounds__(256) void conv0_gn_gelu_kernel(const float *audio, long long audio_bs, const float *w, int ktaps, int stride,
                                                            const float *g, const float *bta, float *y, int T, int y_cs, long long y_bs)
{
    __shared__ float red[16];
    const int c = blockIdx.x, b = blockIdx.y;
    const float *xin = audio + (long long)b * audio_bs;
    float wk[16];
#pragma unroll
    for (int k = 0; k < 16; k++) wk[k] = k < ktaps ? w[c * ktaps + k] : 0.f;
    float v[NT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int t = threadIdx.x + i * 256;
        float a = 0.f;
        if (t < T) {
            const float *xp = xin + (long long)t * stride;
#pragma unroll
            for (int k = 0; k < 16; k++) if (k < ktaps) a = fmaf(wk[k], xp[k], a);
            s += a;
        }
        v[i] = a;
    }
    const float mean = block_sum(s, red) / (float)T;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NT; i++) { if ((int)threadIdx.x + i * 256 < T) { const float d = v[i] - mean; q += d * d; } }
    const float var = block_sum(q, red) / (float)T;
    const float inv = 1.0f / sqrtf(var + 1e-5f), gg = g[c], bb = bta[c];
    float *r = y + (long long)b * y_bs + (long long)c * y_cs;
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int t = threadIdx.x + i * 256;
        if (t < T) r[t] = apply_act((v[i] - mean) * inv * gg + bb, ACT_GELU, 0.f);
    }
}


// The same for `cpw` consecutive channels per workgroup (many streams): a thread's NT x KT input samples are loaded ONCE into
// registers (the stride-5 gathers are what the one-channel form spends its time on: 280 strided loads per thread and channel, 87 %
// of the wave cycles waiting) and reused for every channel; the k weights of a channel are wave-uniform.  1024 threads per workgroup
// keep the register copy at NT = ceil(T / 1024) samples per thread.  Same f32 chain per output as the one-channel kernel; the
// statistics are summed in a different grouping (1024 partial sums instead of 256), as every block-size choice does.
template <int NT, int KT>
__global__ __launch_bounds__(1024) void conv0_gn_gelu_multi_kernel(const float *audio, long long audio_bs, const float *w, int stride, const float *g,
                                                                   const float *bta, float *y, int T, int y_cs, long long y_bs, int cpw)
{
    __shared__ float red[16];
    const int b = blockIdx.y;
    const float *xin = audio + (long long)b * audio_bs;
    float xr[NT][KT];
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int t = threadIdx.x + i * 1024;
        const float *xp = xin + (long long)(t < T ? t : 0) * stride;
#pragma unroll
        for (int k = 0; k < KT; k++) xr[i][k] = xp[k];
    }
    for (int cc = 0; cc < cpw; cc++) {
        const int c = blockIdx.x * cpw + cc;
        float wk[KT];
#pragma unroll
        for (int k = 0; k < KT; k++) wk[k] = w[c * KT + k];
        float v[NT];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            float a = 0.f;
            if ((int)threadIdx.x + i * 1024 < T) {
#pragma unroll
                for (int k = 0; k < KT; k++) a = fmaf(wk[k], xr[i][k], a);
                s += a;
            }
            v[i] = a;
        }
        const float mean = block_sum(s, red) / (float)T;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NT; i++) { if ((int)threadIdx.x + i * 1024 < T) { const float d = v[i] - mean; q += d * d; } }
        const float var = block_sum(q, red) / (float)T;
        const float inv = 1.0f / sqrtf(var + 1e-5f), gg = g[c], bb = bta[c];
        float *r = y + (long long)b * y_bs + (long long)c * y_cs;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int t = threadIdx.x + i * 1024;
            if (t < T) r[t] = apply_act((v[i] - mean) * inv * gg + bb, ACT_GELU, 0.f);
        }
    }
}

// GroupNorm with one group per channel (= per-channel normalisation over time) + GELU, in place.
static __global__ __launch_bounds__(256) void groupnorm_gelu_kernel(float *x, const float *g, const float *bta, int T, int cs, long long bs)
{
    __shared__ float red[16];
    const int c = blockIdx.x, b = blockIdx.y;
    float *r = x + (long long)b * bs + (long long)c * cs;
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) s += r[t];
    const float mean = block_sum(s, red) / (float)T;
    float v = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) { float d = r[t] - mean; v += d * d; }
    const float var = block_sum(v, red) / (float)T;
    const float inv = 1.0f / sqrtf(var + 1e-5f), gg = g[c], bb = bta[c];
    for (int t = threadIdx.x; t < T; t += 256) r[t] = apply_act((r[t] - mean) * inv * gg + bb, ACT_GELU, 0.f);
}

// ------------------------------------------------------------------------------------
// attention (fp32 VALU; T <= 256, head_dim <= 128).  K and V of one head live in LDS.
// qkv: [B][3E][ld] (q rows 0..E, k rows E..2E, v rows 2E..3E), out: [B][E][ld]
// Optional relative-position terms (synth TextEncoder): rel_k/rel_v [2*window+1][hd]
// ------------------------------------------------------------------------------------
struct AttnP {
    const float *qkv; float *out;
    int E, T, heads, cs; long long bs;
    int o_cs; long long o_bs;
    float scale;
    const float *rel_k, *rel_v; int window;
    int qloop;      // attention_mfma_kernel: one workgroup per (head, stream) walks all query tiles, keeping its K / V fragments in registers
};

// One workgroup = one head x 16 query rows; each wave owns 4 query rows and keeps 4 accumulators
// live so that every K / V LDS read feeds 4 FMAs.  grid = (heads * ceil(T/16), B)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

static __global__ __launch_bounds__(256) void attention_kernel(AttnP p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int hd = p.E / p.heads, T = p.T, Tp = T | 1;
    const int qtiles = (T + 15) / 16;
    const int h = blockIdx.x / qtiles, qt = blockIdx.x - h * qtiles, b = blockIdx.y;
    float *KVs = smem;                                  // [hd][Tp]: K during the score pass, then V (one buffer: long windows fit)
    float *Ps = smem + ((hd * Tp + 3) & ~3);            // [4 waves][Tp][4]
    float *Qs = Ps + 16 * Tp;                           // [4 waves][hd][4]
    const float *base = p.qkv + (long long)b * p.bs;
    for (int d = threadIdx.x >> 6; d < hd; d += 4)
        for (int t = threadIdx.x & 63; t < T; t += 64) KVs[d * Tp + t] = base[(long long)(p.E + h * hd + d) * p.cs + t];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *P = Ps + wave * 4 * Tp, *Q = Qs + wave * 4 * hd;
    const int t1 = qt * 16 + wave * 4;
    for (int d = lane; d < hd; d += 64) {
        f32x4 qv;
#pragma unroll
        for (int r = 0; r < 4; r++) qv[r] = (t1 + r < T) ? base[(long long)(h * hd + d) * p.cs + t1 + r] * p.scale : 0.f;
        *reinterpret_cast<f32x4 *>(Q + d * 4) = qv;
    }
    __syncthreads();
    const bool active = t1 < T;                          // idle waves of the last tile still take part in the barriers below
    const int W = p.window;
    float inv[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int t2 = lane; t2 < T; t2 += 64) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            for (int d = 0; d < hd; d++) {
                const float kv = KVs[d * Tp + t2];
                const f32x4 qv = *reinterpret_cast<const f32x4 *>(Q + d * 4);
                a[0] += qv[0] * kv; a[1] += qv[1] * kv; a[2] += qv[2] * kv; a[3] += qv[3] * kv;
            }
            if (p.rel_k) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int rr = t2 - (t1 + r);
                    if (rr >= -W && rr <= W) {
                        float ra = 0.f;
                        const float *rk = p.rel_k + (rr + W) * hd;
                        for (int d = 0; d < hd; d++) ra += Q[d * 4 + r] * rk[d];
                        a[r] += ra;
                    }
                }
            }
            *reinterpret_cast<f32x4 *>(P + t2 * 4) = a;
#pragma unroll
            for (int r = 0; r < 4; r++) mx[r] = fmaxf(mx[r], a[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) mx[r] = wave_max(mx[r]);
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
        for (int t2 = lane; t2 < T; t2 += 64) {
            f32x4 e = *reinterpret_cast<const f32x4 *>(P + t2 * 4);
#pragma unroll
            for (int r = 0; r < 4; r++) { e[r] = expf(e[r] - mx[r]); sum[r] += e[r]; }
            *reinterpret_cast<f32x4 *>(P + t2 * 4) = e;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) inv[r] = 1.0f / wave_sum(sum[r]);
        if (p.rel_v) {
            // relative-value path uses normalised probabilities (same order as the reference definition)
            for (int t2 = lane; t2 < T; t2 += 64) {
                f32x4 e = *reinterpret_cast<const f32x4 *>(P + t2 * 4);
#pragma unroll
                for (int r = 0; r < 4; r++) e[r] *= inv[r];
                *reinterpret_cast<f32x4 *>(P + t2 * 4) = e;
            }
        }
    }
    __syncthreads();                                     // every wave is done with K
    for (int d = threadIdx.x >> 6; d < hd; d += 4)
        for (int t = threadIdx.x & 63; t < T; t += 64) KVs[d * Tp + t] = base[(long long)(2 * p.E + h * hd + d) * p.cs + t];
    __syncthreads();
    if (!active) return;
    for (int d = lane; d < hd; d += 64) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        const float *vr = KVs + d * Tp;
        for (int t2 = 0; t2 < T; t2++) {
            const float vv = vr[t2];
            const f32x4 pr = *reinterpret_cast<const f32x4 *>(P + t2 * 4);
            o[0] += pr[0] * vv; o[1] += pr[1] * vv; o[2] += pr[2] * vv; o[3] += pr[3] * vv;
        }
        if (p.rel_v) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int tq = t1 + r;
                int lo = tq - W < 0 ? 0 : tq - W, hi = tq + W >= T ? T - 1 : tq + W;
                float acc = o[r];
                for (int t2 = lo; t2 <= hi; t2++) acc += P[t2 * 4 + r] * p.rel_v[(t2 - tq + W) * hd + d];
                o[r] = acc;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] *= inv[r];
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (t1 + r < T) p.out[(long long)b * p.o_bs + (long long)(h * hd + d) * p.o_cs + t1 + r] = o[r];
    }
}

// Matrix-core attention for ContentVec (no relative terms, T <= 64*KF).  One workgroup = one head x 16 query rows.
// Latency-shaped for B = 1: every global operand (the Q tile, this wave's K fragments, this wave's V fragments) is loaded into
// registers up front in fully unrolled code, so the kernel pays ~one memory round trip instead of one per loop iteration.
// S = (Q*scale) K^T: A = Q tile, B = K (lanes along t, coalesced), the 4 waves split the key fragments.  Softmax over the
// D fragments (16-lane shuffles + a 4-wave LDS exchange).  O = P V: P goes through LDS into the A layout, V stays in registers
// in the B layout (lane (kq, li) holds V[d = dt*16 + li][t = 4c + kq]); the 4 waves split head_dim (HD/16 <= 4 fragments... one per wave
// for HD = 64; HD = 128 would need two passes and is handled by the VALU kernel).
template <int HD, int KF>
__global__ __launch_bounds__(256) void attention_mfma_kernel(AttnP p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NC = KF * 16;                // t-chunks of 4 covered by the V registers
    const int T = p.T;
    const int qtiles = (T + 15) / 16, kfr = (T + 15) / 16;
    // many streams (qloop): the K and V fragments of a head do not depend on the query tile, so one workgroup per (head, stream) loads them
    // once and walks the query tiles (7 at T = 111): the K / V re-reads -- 5376 workgroups x 57 KB per layer at 64 streams -- drop 7-fold
    const int h = p.qloop ? (int)blockIdx.x : (int)blockIdx.x / qtiles, qt_first = p.qloop ? 0 : (int)blockIdx.x - h * qtiles, b = blockIdx.y;
    const int qt_last = p.qloop ? qtiles : qt_first + 1;
    const int Tq = KF * 64 + 1;
    float *Ps = smem;                         // [16][Tq]
    float *red = Ps + 16 * Tq;                // [4 waves][16 rows] x 2
    const float *base = p.qkv + (long long)b * p.bs;
    const float *qb = base + (long long)(h * HD) * p.cs, *kb = base + (long long)(p.E + h * HD) * p.cs, *vb = base + (long long)(2 * p.E + h * HD) * p.cs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    float qa[HD / 4], kv[HD / 4][KF], vv[NC];
    {
#pragma unroll
        for (int f = 0; f < KF; f++) {
            int t2 = (wave + f * 4) * 16 + li; t2 = t2 < T ? t2 : T - 1;
#pragma unroll
            for (int c = 0; c < HD / 4; c++) kv[c][f] = kb[(long long)(c * 4 + kq) * p.cs + t2];
        }
        const float *vr = vb + (long long)(wave * 16 + li) * p.cs + kq;
#pragma unroll
        for (int c = 0; c < NC; c++) vv[c] = (wave * 16 < HD && c * 4 + kq < T) ? vr[c * 4] : 0.f;
    }
    {
        const int tq = qt_first * 16 + li < T ? qt_first * 16 + li : T - 1;
#pragma unroll
        for (int c = 0; c < HD / 4; c++) qa[c] = qb[(long long)(c * 4 + kq) * p.cs + tq];
    }
    for (int qt = qt_first; qt < qt_last; qt++) {
    const int t1 = qt * 16;
    if (qt != qt_first) __syncthreads();         // the previous tile's probabilities / statistics have been consumed
    f32x4 sacc[KF];
#pragma unroll
    for (int f = 0; f < KF; f++) sacc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < HD / 4; c++) {
        const float a = qa[c] * p.scale;
#pragma unroll
        for (int f = 0; f < KF; f++) sacc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, kv[c][f], sacc[f], 0, 0, 0);
    }
    // (round 6) the NEXT query tile's rows are requested as soon as this tile's products have consumed the registers: the walk over the query tiles of a
    // (head, stream) paid one memory round trip per tile in front of its first MFMA (7 per workgroup at T = 111)
    if (qt + 1 < qt_last) {
        const int tq = t1 + 16 + li < T ? t1 + 16 + li : T - 1;
#pragma unroll
        for (int c = 0; c < HD / 4; c++) qa[c] = qb[(long long)(c * 4 + kq) * p.cs + tq];
    }
    // row statistics: this lane holds rows kq*4 + r, column li of each of its key fragments
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int f = 0; f < KF; f++) {
        const int kf = wave + f * 4;
        const bool ok = kf < kfr && kf * 16 + li < T;
#pragma unroll
        for (int r = 0; r < 4; r++) { if (!ok) sacc[f][r] = -INFINITY; mx[r] = fmaxf(mx[r], sacc[f][r]); }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], o, 64));
        if (li == 0) red[wave * 16 + kq * 4 + r] = mx[r];
    }
    __syncthreads();
    float sum[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = kq * 4 + r;
        mx[r] = fmaxf(fmaxf(red[row], red[16 + row]), fmaxf(red[32 + row], red[48 + row]));
        sum[r] = 0.f;
    }
#pragma unroll
    for (int f = 0; f < KF; f++) {
        const int kf = wave + f * 4;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float e = expf(sacc[f][r] - mx[r]);      // exp(-inf) = 0 for the masked columns
            sum[r] += e;
            Ps[(kq * 4 + r) * Tq + kf * 16 + li] = e;       // every column of [0, 64*KF) is written (zeros beyond T)
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sum[r] += __shfl_xor(sum[r], o, 64);
        if (li == 0) red[64 + wave * 16 + kq * 4 + r] = sum[r];
    }
    __syncthreads();
    // O = P V, this wave's head_dim fragment dt = wave
    if (wave * 16 < HD) {
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
        const float *pr = Ps + li * Tq + kq;
#pragma unroll
        for (int c = 0; c < NC; c += 2) {
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[c * 4], vv[c], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pr[c * 4 + 4], vv[c + 1], o1, 0, 0, 0);
        }
        // D: row = kq*4 + r (query), col = li (d)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = kq * 4 + r, tq = t1 + row;
            const float inv = 1.0f / (red[64 + row] + red[64 + 16 + row] + red[64 + 32 + row] + red[64 + 48 + row]);
            if (tq < T) p.out[(long long)b * p.o_bs + (long long)(h * HD + wave * 16 + li) * p.o_cs + tq] = (o0[r] + o1[r]) * inv;
        }
    }
    }
}

// Small-T attention with relative-position terms (synthesizer TextEncoder: T = return_length <= 64, 2 heads x 96).
// grid = (heads * ceil(T/4), streams); a workgroup owns 4 query rows of one head (one per wave), lanes run along the key axis.
// K, V, both relative tables and the 4 Q rows are staged "all loads into registers, then all LDS stores" in unrolled batches
// (the kernel is a latency chain at B = 1); the dot products keep the sequential d / j order of the definition.
static __global__ __launch_bounds__(256) void relpos_attention_small_kernel(AttnP p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = 256;
    const int hd = p.E / p.heads, T = p.T, Tp = T | 1, W = p.window, NR = 2 * W + 1;
    const int qtiles = (T + 3) / 4;
    const int h = blockIdx.x / qtiles, qt = blockIdx.x - h * qtiles, b = blockIdx.y;
    float *Ks = smem, *Vs = Ks + hd * Tp;                          // [hd][Tp] each, contiguous: K, V
    float *Rk = Vs + hd * Tp, *Rv = Rk + NR * hd;                  // [NR][hd] each, contiguous
    float *Qs = Rv + NR * hd;                                      // [4][hd]
    float *S = Qs + 4 * hd;                                        // [4][64]
    const float *base = p.qkv + (long long)b * p.bs;
    // staging without integer divisions (the first version spent most of its 19 us on them: three per staged element): K / V rows
    // are walked as (d, t) with t padded to a power of two, the relative-position tables are one contiguous copy, Q one row per pass
    // All global loads are issued before the first LDS write (one memory round trip for the whole staging instead of one per loop
    // iteration: the kernel was bound by exactly that serialisation).
    const int tsh = T <= 32 ? 5 : 6, tmask = (1 << tsh) - 1;
    constexpr int KV_IT = 16, RT_IT = 12;
    const int kv_n = hd << tsh, rt_n = NR * hd;
    float kk[KV_IT], vv[KV_IT], rk[RT_IT], rv[RT_IT], qq[4];
#pragma unroll
    for (int u = 0; u < KV_IT; u++) {
        const int idx = threadIdx.x + u * NT, d = idx >> tsh, t = idx & tmask;
        const bool ok = idx < kv_n && t < T;
        kk[u] = ok ? base[(long long)(p.E + h * hd + d) * p.cs + t] : 0.f;
        vv[u] = ok ? base[(long long)(2 * p.E + h * hd + d) * p.cs + t] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < RT_IT; u++) {
        const int j = threadIdx.x + u * NT;
        rk[u] = j < rt_n ? p.rel_k[j] : 0.f;
        rv[u] = j < rt_n ? p.rel_v[j] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int tq = qt * 4 + r;
        qq[r] = (threadIdx.x < hd && tq < T) ? base[(long long)(h * hd + threadIdx.x) * p.cs + tq] * p.scale : 0.f;
    }
#pragma unroll
    for (int u = 0; u < KV_IT; u++) {
        const int idx = threadIdx.x + u * NT, d = idx >> tsh, t = idx & tmask;
        if (idx < kv_n && t < T) { Ks[d * Tp + t] = kk[u]; Vs[d * Tp + t] = vv[u]; }
    }
#pragma unroll
    for (int u = 0; u < RT_IT; u++) {
        const int j = threadIdx.x + u * NT;
        if (j < rt_n) { Rk[j] = rk[u]; Rv[j] = rv[u]; }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) if (threadIdx.x < hd) Qs[r * hd + threadIdx.x] = qq[r];
    // sizes beyond the unrolled staging (not reached by any official configuration: 2 heads x 96, window 10, T <= 64)
    for (int idx = threadIdx.x + KV_IT * NT; idx < kv_n; idx += NT) {
        const int d = idx >> tsh, t = idx & tmask;
        if (t < T) { Ks[d * Tp + t] = base[(long long)(p.E + h * hd + d) * p.cs + t]; Vs[d * Tp + t] = base[(long long)(2 * p.E + h * hd + d) * p.cs + t]; }
    }
    for (int j = threadIdx.x + RT_IT * NT; j < rt_n; j += NT) { Rk[j] = p.rel_k[j]; Rv[j] = p.rel_v[j]; }
    for (int d = threadIdx.x + NT; d < hd; d += NT)
        for (int r = 0; r < 4; r++) { const int tq = qt * 4 + r; Qs[r * hd + d] = tq < T ? base[(long long)(h * hd + d) * p.cs + tq] * p.scale : 0.f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = qt * 4 + wave;
    if (i >= T) return;
    const float *q = Qs + wave * hd;
    float *Sr = S + wave * 64;
    float sc = -INFINITY;
    if (lane < T) {
        const int j = lane;
        float a = 0.f;
#pragma unroll 8
        for (int d = 0; d < hd; d++) a += q[d] * Ks[d * Tp + j];
        const int r = j - i;
        if (r >= -W && r <= W) {
            float ra = 0.f;
            const float *rk = Rk + (r + W) * hd;
#pragma unroll 8
            for (int d = 0; d < hd; d++) ra += q[d] * rk[d];
            a += ra;
        }
        sc = a;
    }
    const float mx = wave_max(sc);
    const float ex = lane < T ? expf(sc - mx) : 0.f;
    const float inv = 1.0f / wave_sum(ex);
    if (lane < T) Sr[lane] = ex * inv;
    wave_lds_sync();
    const int lo = i - W < 0 ? 0 : i - W, hi = i + W >= T ? T - 1 : i + W;
    for (int d = lane; d < hd; d += 64) {
        float a = 0.f;
#pragma unroll 8
        for (int j = 0; j < T; j++) a += Sr[j] * Vs[d * Tp + j];
#pragma unroll 8
        for (int j = lo; j <= hi; j++) a += Sr[j] * Rv[(j - i + W) * hd + d];
        p.out[(long long)b * p.o_bs + (long long)(h * hd + d) * p.o_cs + i] = a;
    }
}

// The same attention on the matrix cores (one stream: the VALU kernel above is a latency chain of ~200 dependent LDS-read + FMA steps per lane,
// 12.4 us per layer).  grid = (heads * ceil(T / 16), streams): a workgroup owns 16 query columns of one head.  Every product is a 16x16x4 fp32
// MFMA (A[i = lane & 15][k = lane >> 4], B[k][j = lane & 15], D[row = (lane >> 4) * 4 + r][col = lane & 15]):
//   scores[i][j] = sum_d q[d][i] k[d][j]            P[i][r] = sum_d q[d][i] rel_k[r][d]        scores[i][j] += P[i][j - i + W] inside the window
//   out[c][i]    = sum_j v[c][j] S[i][j] + sum_r rel_v[r][c] Ssk[i][r]      with Ssk[i][r] = S[i][i + r - W] (zero outside [0, T))
// Padded k (j >= T, r >= NR) multiplies a ZEROED S / Ssk entry by a finite staged value; padded rows / columns of D are not stored.
static __global__ __launch_bounds__(256) void relpos_attention_mfma_kernel(AttnP p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = 256;
    const int kc = p.E / p.heads, T = p.T, TP = T | 1, Wd = p.window, NR = 2 * Wd + 1, NRP = (NR + 3) & ~3;
    const int JF = (T + 15) >> 4, SW = JF * 16, RF = (NR + 15) >> 4, PW = RF * 16;
    const int h = blockIdx.x / JF, qb = blockIdx.x - h * JF, b = blockIdx.y;
    const int col0 = qb * 16, nq = T - col0 < 16 ? T - col0 : 16;
    float *q = smem, *kk = q + kc * 16, *vv = kk + kc * TP, *rk = vv + kc * TP, *rv = rk + PW * kc;
    float *Sx = rv + NRP * kc, *P = Sx + 16 * SW, *Ssk = P + 16 * PW;
    const float *base = p.qkv + (long long)b * p.bs;
    // staging: every global load is issued before the first LDS write (one memory round trip)
    const int tsh = T <= 32 ? 5 : 6, tmask = (1 << tsh) - 1;
    constexpr int KV_IT = 16, RT_IT = 12, Q_IT = 6;
    const int kv_n = kc << tsh, rk_n = PW * kc, rv_n = NRP * kc, rt_n = NR * kc, q_n = kc * 16;
    float kr[KV_IT], vr[KV_IT], rkr[RT_IT], rvr[RT_IT], qr[Q_IT];
#pragma unroll
    for (int u = 0; u < Q_IT; u++) {
        const int idx = threadIdx.x + u * NT, d = idx >> 4, c = idx & 15;
        qr[u] = (idx < q_n && c < nq) ? base[(long long)(h * kc + d) * p.cs + col0 + c] * p.scale : 0.f;
    }
#pragma unroll
    for (int u = 0; u < KV_IT; u++) {
        const int idx = threadIdx.x + u * NT, d = idx >> tsh, t = idx & tmask;
        const bool ok = idx < kv_n && t < T;
        kr[u] = ok ? base[(long long)(p.E + h * kc + d) * p.cs + t] : 0.f;
        vr[u] = ok ? base[(long long)(2 * p.E + h * kc + d) * p.cs + t] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < RT_IT; u++) {
        const int j = threadIdx.x + u * NT;
        rkr[u] = j < rt_n ? p.rel_k[j] : 0.f;
        rvr[u] = j < rt_n ? p.rel_v[j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < Q_IT; u++) { const int idx = threadIdx.x + u * NT; if (idx < q_n) q[idx] = qr[u]; }
#pragma unroll
    for (int u = 0; u < KV_IT; u++) {
        const int idx = threadIdx.x + u * NT, d = idx >> tsh, t = idx & tmask;
        if (idx < kv_n && t < TP) { kk[d * TP + t] = kr[u]; vv[d * TP + t] = vr[u]; }       // (column T of the odd padding: zero)
    }
#pragma unroll
    for (int u = 0; u < RT_IT; u++) {
        const int j = threadIdx.x + u * NT;
        if (j < rk_n) rk[j] = rkr[u];
        if (j < rv_n) rv[j] = rvr[u];
    }
    // sizes beyond the unrolled staging (no official configuration: 2 heads x 96, window 10, T <= 32)
    for (int idx = threadIdx.x + Q_IT * NT; idx < q_n; idx += NT) { const int d = idx >> 4, c = idx & 15; q[idx] = c < nq ? base[(long long)(h * kc + d) * p.cs + col0 + c] * p.scale : 0.f; }
    for (int idx = threadIdx.x + KV_IT * NT; idx < kv_n; idx += NT) {
        const int d = idx >> tsh, t = idx & tmask;
        if (t < TP) { kk[d * TP + t] = t < T ? base[(long long)(p.E + h * kc + d) * p.cs + t] : 0.f; vv[d * TP + t] = t < T ? base[(long long)(2 * p.E + h * kc + d) * p.cs + t] : 0.f; }
    }
    for (int j = threadIdx.x + RT_IT * NT; j < rk_n; j += NT) rk[j] = j < rt_n ? p.rel_k[j] : 0.f;
    for (int j = threadIdx.x + RT_IT * NT; j < rv_n; j += NT) rv[j] = j < rt_n ? p.rel_v[j] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, kq = lane >> 4;
    // scores and P: items (16-column block of keys), then (16-row block of relative positions), one per wave and pass
    for (int it = wave; it < JF + RF; it += 4) {
        const bool is_p = it >= JF;
        const int f = is_p ? it - JF : it;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        const float *qa = q + kq * 16 + li;
        const float *bb = is_p ? rk + (f * 16 + li) * kc + kq : kk + kq * TP + f * 16 + li;
        const int bst = is_p ? 4 : 4 * TP;
        for (int ks = 0; ks + 1 < kc / 4; ks += 2) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[ks * 64], bb[ks * bst], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[(ks + 1) * 64], bb[(ks + 1) * bst], a1, 0, 0, 0);
        }
        if ((kc / 4) & 1) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[(kc / 4 - 1) * 64], bb[(kc / 4 - 1) * bst], a0, 0, 0, 0);
        a0 += a1;
        float *dst = is_p ? P + f * 16 : Sx + f * 16;
        const int dw = is_p ? PW : SW;
#pragma unroll
        for (int r = 0; r < 4; r++) dst[(kq * 4 + r) * dw + li] = a0[r];
    }
    __syncthreads();
    // softmax with the relative-position term: 16 lanes per query row, keys strided over the lanes
    {
        const int i = threadIdx.x >> 4, gi = col0 + i;
        float *Sr = Sx + i * SW, *Kr = Ssk + i * PW;
        const float *Pr = P + i * PW;
        float mx = -INFINITY;
        if (gi < T) {
            for (int j = li; j < T; j += 16) {
                float a = Sr[j];
                const int r = j - gi;
                if (r >= -Wd && r <= Wd) a += Pr[r + Wd];
                Sr[j] = a; mx = fmaxf(mx, a);
            }
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        float sum = 0.f;
        if (gi < T) for (int j = li; j < T; j += 16) { const float ex = expf(Sr[j] - mx); Sr[j] = ex; sum += ex; }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const float inv = 1.0f / sum;
        for (int j = li; j < SW; j += 16) Sr[j] = (gi < T && j < T) ? Sr[j] * inv : 0.f;
        // (the 16 lanes of a row run in lockstep inside one wave: Sr is complete before it is read back skewed)
        for (int r = li; r < PW; r += 16) { const int j = gi + r - Wd; Kr[r] = (gi < T && r < NR && j >= 0 && j < T) ? Sr[j] : 0.f; }
    }
    __syncthreads();
    // attention output of the own columns: items = 16-channel blocks of the head
    for (int cf = wave; cf < kc / 16; cf += 4) {
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        const float *va = vv + (cf * 16 + li) * TP + kq, *sb = Sx + li * SW + kq;
        for (int ks = 0; ks < (T + 3) / 4; ks++) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va[ks * 4], sb[ks * 4], a0, 0, 0, 0);
        const float *ra = rv + kq * kc + cf * 16 + li, *kb = Ssk + li * PW + kq;
        for (int ks = 0; ks < NRP / 4; ks++) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[ks * 4 * kc], kb[ks * 4], a1, 0, 0, 0);
        a0 += a1;
        // D: row = channel cf * 16 + kq * 4 + r, col = query li
        if (col0 + li < T) {
#pragma unroll
            for (int r = 0; r < 4; r++) p.out[(long long)b * p.o_bs + (long long)(h * kc + cf * 16 + kq * 4 + r) * p.o_cs + col0 + li] = a0[r];
        }
    }
}

// ------------------------------------------------------------------------------------
// RMVPE head: bidirectional GRU recurrence (input projections come from the implicit GEMM)
// gi: [B][2*3H][ld] (forward gates rows 0..3H, backward rows 3H..6H; biases b_ih included)
// whhT: [2][H][3H] (transposed so lanes read consecutive rows), bhh: [2][3H]
// out: [B][2H][ld] (forward h rows 0..H, backward rows H..2H)
// ------------------------------------------------------------------------------------
static __global__ __launch_bounds__(1024) void gru_kernel(const float *gi, int gi_cs, long long gi_bs, const float *whhT, const float *bhh,
                                                   float *out, int o_cs, long long o_bs, int H, int Tm)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hs = smem;          // [H]
    float *gh = smem + H;      // [3H]
    const int dir = blockIdx.x, b = blockIdx.y, r = threadIdx.x;
    const float *gib = gi + (long long)b * gi_bs + (long long)dir * 3 * H * gi_cs;
    const float *wt = whhT + (long long)dir * H * 3 * H;
    const float *bh = bhh + dir * 3 * H;
    float *ob = out + (long long)b * o_bs + (long long)dir * H * o_cs;
    if (r < H) hs[r] = 0.f;
    __syncthreads();
    for (int step = 0; step < Tm; step++) {
        const int t = dir == 0 ? step : Tm - 1 - step;
        if (r < 3 * H) {
            float a = bh[r];
            for (int j = 0; j < H; j++) a += wt[(long long)j * 3 * H + r] * hs[j];
            gh[r] = a;
        }
        __syncthreads();
        if (r < H) {
            float ir = gib[(long long)r * gi_cs + t], iz = gib[(long long)(H + r) * gi_cs + t], in_ = gib[(long long)(2 * H + r) * gi_cs + t];
            float rg = 1.0f / (1.0f + expf(-(ir + gh[r])));
            float zg = 1.0f / (1.0f + expf(-(iz + gh[H + r])));
            float ng = tanhf(in_ + rg * gh[2 * H + r]);
            float hn = (1.f - zg) * ng + zg * hs[r];
            hs[r] = hn;
            ob[(long long)r * o_cs + t] = hn;
        }
        __syncthreads();
    }
}

// Multi-CU recurrence for few streams (B <= 8): 8 workgroups per direction, each owning 32 hidden units = 96 gate rows
// whose 96 KB slice of W_hh stays in LDS for all steps.  After every step the 8 slices exchange their 32 new h values through
// 8-byte {epoch, value} granules (cdna_hip_programming.md guideline 16, form R2: the data is the flag; relaxed agent-scope
// stores / loads, no fences, placement independent).  Two granule slots alternate by step parity; the granule buffer is zeroed
// by a memset node before every launch of a captured graph, and once per plan for eager launches (GruMultiP::epoch).  Every spin is bounded: on timeout the stream's status word is raised instead of hanging.
struct GruMultiP {
    const float *gi; int gi_cs; long long gi_bs;
    const float *whh;          // [2][3H][H] row-major
    const float *bhh;          // [2][3H]
    float *out; int o_cs; long long o_bs;
    unsigned long long *gran;  // [B][2 dirs][2 slots][H]
    int *status;               // per stream, stride status_stride ints
    int status_stride;
    int Tm;
    unsigned epoch;            // tags of this launch are epoch + 1 ... epoch + Tm (round 6: the host advances it by Tm per launch, so stale granules of earlier
                               // chunks never match and the buffer needs no memset in front of every launch -- a 5 us fill kernel on the f0 chain)
};
// Round 3: the 96 x 256 slice of W_hh lives in REGISTERS (thread (row r, quarter q) keeps its 64 weights for all steps: the matvec
// reads only h from LDS, 16 broadcast b128 reads per thread, instead of streaming 96 KB of weights through LDS every step), the
// slice's input gates gi[Tm][96] are copied to LDS once (they were three global loads per step on the critical path), and a step
// has two workgroup barriers instead of three.  384 threads; 98 -> ~45 us for 2 x 32 steps.
static __global__ __launch_bounds__(384) void gru_multi_kernel(GruMultiP p)
{
    constexpr int H = 256, G = 8, U = H / G, ROWS = 3 * U, NT = 384;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hs = smem;                   // [H]
    float *gh = hs + H;                 // [ROWS]
    float *gis = gh + ROWS;             // [Tm][ROWS]: input gates of this slice (b_ih included), row = gate * U + unit
    const int g = blockIdx.x, dir = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const int r = tid >> 2, q = tid & 3;                 // row of the slice, quarter of the hidden vector
    const int gate = r / U, u = r - gate * U;
    const float *wsrc = p.whh + (long long)dir * 3 * H * H + (long long)(gate * H + g * U + u) * H + q * 64;
    f32x4 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = *reinterpret_cast<const f32x4 *>(wsrc + i * 4);
    const float bias = p.bhh[dir * 3 * H + gate * H + g * U + u];
    const float *gib = p.gi + (long long)b * p.gi_bs + (long long)dir * 3 * H * p.gi_cs;
    // (round 6: in batches of eight requests -- one memory round trip per batch instead of one per element of the strided copy loop)
    for (int i0 = tid; i0 < p.Tm * ROWS; i0 += 8 * NT) {
        float v8[8]; int d8[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = i0 + k * NT;
            const bool ok = i < p.Tm * ROWS;
            const int ii = ok ? i : 0;
            const int rr = ii / p.Tm, t = ii - rr * p.Tm, gg = rr / U, uu = rr - gg * U;      // coalesced along time
            v8[k] = gib[(long long)(gg * H + g * U + uu) * p.gi_cs + t];
            d8[k] = ok ? t * ROWS + rr : -1;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) if (d8[k] >= 0) gis[d8[k]] = v8[k];
    }
    if (tid < H) hs[tid] = 0.f;
    float *ob = p.out + (long long)b * p.o_bs + (long long)dir * H * p.o_cs;
    unsigned long long *gr = p.gran + ((long long)(b * 2 + dir) * 2) * H;
    __syncthreads();
    bool dead = false;
    for (int step = 0; step < p.Tm; step++) {
        const int t = dir == 0 ? step : p.Tm - 1 - step;
        if (step > 0) {
            // gather h_{step-1}: granule `tid` of slot (step-1)&1 must carry tag == step
            if (tid < H) {
                const unsigned long long *src = gr + ((step - 1) & 1) * H + tid;
                unsigned long long x = 0;
                unsigned spins = 0;
                while (!dead) {
                    x = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(x >> 32) == p.epoch + (unsigned)step) break;
                    if (++spins > (1u << 22)) { dead = true; atomicOr(&p.status[b * p.status_stride], (int)ST_HANDOFF); }
                    __builtin_amdgcn_s_sleep(1);
                }
                hs[tid] = __uint_as_float((unsigned)x);
            }
            __syncthreads();
        }
        // row r, columns q * 64 ..: the four quarters of a row sit in adjacent lanes
        {
            const float *hq = hs + q * 64;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const f32x4 hv = *reinterpret_cast<const f32x4 *>(hq + i * 4);
                a0 = fmaf(w[i][0], hv[0], a0); a1 = fmaf(w[i][1], hv[1], a1); a2 = fmaf(w[i][2], hv[2], a2); a3 = fmaf(w[i][3], hv[3], a3);
            }
            float a = (a0 + a1) + (a2 + a3);
            a += __shfl_xor(a, 1, 64);
            a += __shfl_xor(a, 2, 64);
            if (q == 0) gh[r] = a + bias;
        }
        __syncthreads();
        if (tid < U) {
            const int unit = g * U + tid;
            const float *gi = gis + t * ROWS;
            const float ir = gi[tid], iz = gi[U + tid], in_ = gi[2 * U + tid];
            const float rg = 1.0f / (1.0f + expf(-(ir + gh[tid])));
            const float zg = 1.0f / (1.0f + expf(-(iz + gh[U + tid])));
            const float ng = tanhf(in_ + rg * gh[2 * U + tid]);
            const float hn = (1.f - zg) * ng + zg * hs[unit];
            ob[(long long)unit * p.o_cs + t] = hn;
            __hip_atomic_store(gr + (step & 1) * H + unit, ((unsigned long long)(p.epoch + (unsigned)(step + 1)) << 32) | (unsigned long long)__float_as_uint(hn),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (no barrier here: hs[unit] of the own slice is rewritten only after this thread's granule has been polled back, gh only
        // after the next step's first barrier)
    }
}

// ------------------------------------------------------------------------------------
// per-stream state + per-call parameters
// ------------------------------------------------------------------------------------
// (CallParams / StreamState: state.hip.h)

// RMVPE decode (rmvpe.rs:118-133, 243-248) + pitch shift (rvc.rs:121-122) + pitch cache update and
// slice (rvc.rs:167-179) + get_f0_post (f0/mod.rs:7-12).  One workgroup per stream.
struct PitchP {
    const float *sal; int sal_cs; long long sal_bs;   // salience [B][360][ld] (channel-major)
    int Tm;
    StreamState *st; const CallParams *cp;
    float *f0;            // [B][Tm] (shifted f0, tap)
    float *pitchf;        // [B][R]
    int *pitch;           // [B][R]
    int shift, cache_start, read_start, R;
    float threshold;
    int update;           // 0: decode only (RvcInfer::pitch, rvc.rs:111-131), 1: also update + slice the cache (infer)
};

static __global__ __launch_bounds__(1024) void pitch_post_kernel(PitchP p)
{
    __shared__ float f0s[1024];
    __shared__ float cache[1024];
    __shared__ int idxs[1024];
    const int b = blockIdx.x, t = threadIdx.x;
    StreamState *st = p.st + b;
    const float up = st->uppower;        // per stream: every stream of a batch is its own caller with its own pitch shift (obs-rvc/src/lib.rs:701-707)
    // Row scan split over bin groups: thread (tt = t % TT, grp = t / TT) scans bins [grp*BPG, ...) of time step tt (loads
    // coalesced along time), then group 0 combines.  Same result as the sequential scan of the zero-padded row (368 wide,
    // "first strictly greater wins", padded[0] = 0): start = first index of the maximum if it is > 0, else 0.
    int TT = 1; while (TT < p.Tm) TT <<= 1;
    TT = TT < 1024 ? TT : 1024;
    const int NG = 1024 / TT, BPG = (360 + NG - 1) / NG, tt = t & (TT - 1), grp = t / TT;
    {
        float best = 0.f, mx = -INFINITY; int start = 0;
        if (tt < p.Tm) {
            const float *col = p.sal + (long long)b * p.sal_bs + tt;
            const int i0 = grp * BPG, i1 = (i0 + BPG < 360) ? i0 + BPG : 360;
#pragma unroll 4
            for (int i = i0; i < i1; i++) { const float v = col[(long long)i * p.sal_cs]; if (v > best) { best = v; start = i + 4; } mx = fmaxf(mx, v); }
        }
        f0s[t] = best; cache[t] = mx; idxs[t] = start;
    }
    __syncthreads();
    float hz_out = 0.f;
    if (grp == 0 && tt < p.Tm) {
        const float *col = p.sal + (long long)b * p.sal_bs + tt;
        int start = 0; float best = 0.f, mx = -INFINITY;
        for (int g = 0; g < NG; g++) {
            const float v = f0s[g * TT + tt];
            if (v > best) { best = v; start = idxs[g * TT + tt]; }
            mx = fmaxf(mx, cache[g * TT + tt]);
        }
        float hz = 0.f;
        if (start + 8 >= 360) { atomicOr(&st->status, (int)ST_PANIC); }
        else {
            float sv[9];
#pragma unroll
            for (int y = 0; y < 9; y++) sv[y] = col[(long long)(start + y) * p.sal_cs];
            float ps = 0.f, ws = 0.f;
#pragma unroll
            for (int y = 0; y < 9; y++) { const float cm = ((float)(start + y) - 4.f) * 20.f + 1997.3794084376191f; ps += sv[y] * cm; ws += sv[y]; }
            float cents = ps / ws;
            if (!(mx > p.threshold)) cents = 0.f;
            hz = 10.0f * powf(2.0f, cents / 1200.0f);
            if (hz == 10.0f) hz = 0.f;
        }
        hz *= up;
        hz_out = hz;
        p.f0[(long long)b * p.Tm + tt] = hz;
    }
    __syncthreads();
    if (grp == 0 && tt < p.Tm) f0s[tt] = hz_out;
    if (!p.update) return;
    __syncthreads();
    cache[t] = st->cache_pitchf[t];
    __syncthreads();
    // copy_within(shift.., 0): cache[i] = cache[i+shift] for i < 1024-shift (tail keeps old values)
    float v = (t + p.shift < 1024) ? cache[t + p.shift] : cache[t];
    // cache[cache_start..] = pitchf[3..len-1]
    if (t >= p.cache_start) v = f0s[3 + (t - p.cache_start)];
    __syncthreads();
    cache[t] = v;
    st->cache_pitchf[t] = v;
    __syncthreads();
    if (t < p.R) {
        float f = cache[p.read_start + t];
        p.pitchf[(long long)b * p.R + t] = f;
        const float mel_min = logf(50.0f / 700.0f + 1.f) * 1127.f, mel_max = logf(500.0f / 700.0f + 1.f) * 1127.f;
        float x = logf(f / 700.0f + 1.f) * 1127.f;
        if (!(x <= 0.f)) x = (x - mel_min) * 254.f / (mel_max - mel_min) + 1.f;
        x = fminf(fmaxf(x, 1.f), 255.f);
        p.pitch[(long long)b * p.R + t] = (int)roundf(x);
    }
}

// ------------------------------------------------------------------------------------
// small glue kernels
// ------------------------------------------------------------------------------------
// phone[c][r] = feats[min((skip_head + r) / 2, T - 1)][c]   (rvc.rs:99-109 + 155; Q2, Q8)
static __global__ void gather_phone_kernel(const float *cv, int cv_cs, long long cv_bs, int C, int T, int skip_head, int R,
                                    float *phone, int ph_cs, long long ph_bs)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= C * R) return;
    int c = i / R, r = i - c * R;
    int s = (skip_head + r) / 2; s = s < T - 1 ? s : T - 1;
    phone[(long long)b * ph_bs + (long long)c * ph_cs + r] = cv[(long long)b * cv_bs + (long long)c * cv_cs + s];
}

// (1, 2T+1, C) output of RvcInfer::extract_feature (rvc.rs:99-109), contiguous
static __global__ void extract_feature_kernel(const float *cv, int cv_cs, int C, int T, float *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int T2 = 2 * T + 1;
    if (i >= T2 * C) return;
    int k = i / C, c = i - k * C;
    int s = k / 2; s = s < T - 1 ? s : T - 1;
    out[i] = cv[(long long)c * cv_cs + s];
}

// TextEncoder front: x = lrelu((lin + emb_pitch[pitch]) * sqrt(H), 0.1), in place on lin [B][H][ld]
static __global__ void embed_pitch_kernel(float *x, int cs, long long bs, const float *emb, const int *pitch, int H, int R, float sq)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= H * R) return;
    int c = i / R, t = i - c * R;
    float *xp = x + (long long)b * bs + (long long)c * cs + t;
    float a = *xp + emb[(long long)pitch[(long long)b * R + t] * H + c];
    a *= sq;
    *xp = a > 0.f ? a : a * 0.1f;
}

// Philox4x32-10, the same counter layout as oracle/rvc_oracle.c (ora_philox_normal)
// (philox4x32_10 / u01 / philox_normal4: state.hip.h)

// z_p = m + exp(logs) * eps * 0.66666 ; stats [B][2I][ld] -> z [B][I][ld]; eps index = c*T + t
static __global__ void prior_sample_kernel(const float *stats, int s_cs, long long s_bs, float *z, int z_cs, long long z_bs, int I, int T,
                                    const StreamState *st, const CallParams *cp)
{
    int blk = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    int total = I * T;
    if (blk * 4 >= total) return;
    float n4[4];
    philox_normal4(cp->seed, st[b].stream_id, st[b].chunk, 0u, (uint32_t)blk, n4);
    for (int j = 0; j < 4; j++) {
        int i = blk * 4 + j;
        if (i >= total) break;
        int c = i / T, t = i - c * T;
        float m = stats[(long long)b * s_bs + (long long)c * s_cs + t], lg = stats[(long long)b * s_bs + (long long)(I + c) * s_cs + t];
        z[(long long)b * z_bs + (long long)c * z_cs + t] = m + expf(lg) * n4[j] * 0.66666f;
    }
}

// channel flip (Flip flow): y[c] = x[C-1-c]
static __global__ void flip_channels_kernel(const float *x, int x_cs, long long x_bs, float *y, int y_cs, long long y_bs, int C, int T)
{
    // (source and destination have their own strides: with composed WaveNets the latent is a row range of a wider tensor -- a shared stride put every
    //  stream but the first in the wrong place: found in round 5 by the first multi-stream test of an odd flow count)
    int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= C * T) return;
    int c = i / T, t = i - c * T;
    y[(long long)b * y_bs + (long long)c * y_cs + t] = x[(long long)b * x_bs + (long long)(C - 1 - c) * x_cs + t];
}

// WaveNet gate: acts[c] = tanh(a[c]) * sigmoid(a[H + c])   (conditioning already folded into the conv bias)
static __global__ void gate_kernel(const float *a, int a_cs, long long a_bs, float *y, int y_cs, long long y_bs, int H, int T)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= H * T) return;
    int c = i / T, t = i - c * T;
    float ta = a[(long long)b * a_bs + (long long)c * a_cs + t], sa = a[(long long)b * a_bs + (long long)(H + c) * a_cs + t];
    y[(long long)b * y_bs + (long long)c * y_cs + t] = tanhf(ta) * (1.0f / (1.0f + expf(-sa)));
}

// timeline probe (RVC_STAMPS=1): device wall clock (constant 100 MHz) at a point of a stream's kernel chain
static __global__ void stamp_kernel(unsigned long long *slot) { *slot = wall_clock64(); }

// average of up to three ResBlock outputs: y = ((a + b) + c) * inv
static __global__ void mean3_kernel(const float *a, const float *b2, const float *c, int i_cs, long long i_bs, float *y, int y_cs, long long y_bs, int C, int T, float inv)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= C * T) return;
    int ch = i / T, t = i - ch * T;
    const long long o = (long long)b * i_bs + (long long)ch * i_cs + t;
    float v = a[o];
    if (b2) v += b2[o];
    if (c) v += c[o];
    y[(long long)b * y_bs + (long long)ch * y_cs + t] = v * inv;
}

// AvgPool2d(2,2): x [B][C][H(+2)][ld] -> y [B][C][H/2(+2)][ld2]
static __global__ void avgpool2_kernel(const float *x, int x_ld, int x_cs, long long x_bs, float *y, int y_ld, int y_cs, long long y_bs, int C, int H2, int W2)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= C * H2 * W2) return;
    int c = i / (H2 * W2), r = i - c * H2 * W2, h = r / W2, w = r - h * W2;
    const float *s = x + (long long)b * x_bs + (long long)c * x_cs + (long long)(2 * h) * x_ld + 2 * w;
    y[(long long)b * y_bs + (long long)c * y_cs + (long long)h * y_ld + w] = (s[0] + s[1] + s[x_ld] + s[x_ld + 1]) * 0.25f;
}

// (3, Tm, n_mels) conv output image -> GRU input [B][3*n_mels][ld]: feat[c*n_mels + m][t] = img[c][t][m]
static __global__ void gru_input_kernel(const float *img, int i_ld, int i_cs, long long i_bs, float *feat, int f_cs, long long f_bs, int Tm, int n_mels)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= 3 * n_mels * Tm) return;
    int row = i / Tm, t = i - row * Tm, c = row / n_mels, m = row - c * n_mels;
    feat[(long long)b * f_bs + (long long)row * f_cs + t] = img[(long long)b * i_bs + (long long)c * i_cs + (long long)t * i_ld + m];
}

// ------------------------------------------------------------------------------------
// NSF harmonic source (SineGen, harmonic_num = 0, + Linear(1,1) + tanh).  One workgroup of
// 1024 threads per stream; the sample-rate phase cumsum is a block prefix scan.
// ------------------------------------------------------------------------------------
struct SrcP {
    const float *pitchf;   // [B][T]
    float *src;            // [B][1][ld] interior pointer
    long long src_bs;
    int T, upp; float sr;
    float lin_w, lin_b;
    const StreamState *st; const CallParams *cp;
};

static __global__ __launch_bounds__(1024) void nsf_source_kernel(SrcP p)
{
    __shared__ float rad[512], cum[512];
    __shared__ float part[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int T = p.T, upp = p.upp;
    const long long N = (long long)T * upp;
    const float *f0 = p.pitchf + (long long)b * T;
    if (tid == 0) {
        float c = 0.f;
        for (int t = 0; t < T; t++) { float r = fmodf(f0[t] / p.sr, 1.0f); rad[t] = r; c += r; cum[t] = c * (float)upp; }
    }
    __syncthreads();
    // each thread owns a contiguous segment whose length is a multiple of 4 (one Philox block = 4 samples)
    long long seg = ((N + 1023) / 1024 + 3) / 4 * 4;
    long long i0 = (long long)tid * seg, i1 = i0 + seg < N ? i0 + seg : N;
    auto interp = [&](long long i) -> float {
        float pos = (N > 1) ? (float)i * (float)(T - 1) / (float)(N - 1) : 0.f;
        int j0 = (int)floorf(pos); if (j0 > T - 1) j0 = T - 1; int j1 = j0 + 1 < T ? j0 + 1 : T - 1;
        float w = pos - (float)j0;
        float v = cum[j0] * (1.0f - w) + cum[j1] * w;
        return fmodf(v, 1.0f);
    };
    float local = 0.f;
    if (i0 < N) {
        float prev = i0 > 0 ? interp(i0 - 1) : 0.f;
        for (long long i = i0; i < i1; i++) {
            float cur = interp(i);
            float shift = (i > 0 && (cur - prev) < 0.f) ? -1.0f : 0.f;
            local += rad[(int)(i / upp)] + shift;
            prev = cur;
        }
    }
    part[tid] = local;
    __syncthreads();
    // inclusive scan over the 1024 partial sums (Hillis-Steele)
    for (int o = 1; o < 1024; o <<= 1) {
        float v = tid >= o ? part[tid - o] : 0.f;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    if (i0 < N) {
        float phase = tid > 0 ? part[tid - 1] : 0.f;
        float prev = i0 > 0 ? interp(i0 - 1) : 0.f;
        float *out = p.src + (long long)b * p.src_bs;
        const uint32_t seed = p.cp->seed, sid = p.st[b].stream_id, chunk = p.st[b].chunk;
        float nz[4];
        for (long long i = i0; i < i1; i++) {
            if (((i - i0) & 3) == 0) philox_normal4(seed, sid, chunk, 1u, (uint32_t)(i >> 2), nz);
            float cur = interp(i);
            float shift = (i > 0 && (cur - prev) < 0.f) ? -1.0f : 0.f;
            int t = (int)(i / upp);
            phase += rad[t] + shift;
            prev = cur;
            float sine = sinf(phase * 6.28318530717958647692f) * 0.1f;
            float uv = f0[t] > 0.f ? 1.f : 0.f;
            float namp = uv * 0.003f + (1.f - uv) * 0.1f / 3.f;
            float sw = sine * uv + namp * nz[(i - i0) & 3];
            out[i] = tanhf(p.lin_w * sw + p.lin_b);
        }
    }
}

// rvc_infer_batch_g: the states of one geometry bucket, gathered into a contiguous block (dir = 0) / scattered back (dir = 1); one workgroup per stream
static __global__ void state_gather_kernel(StreamState *all, StreamState *bucket, const int *idx, int dir)
{
    const int j = blockIdx.x, s = idx[j];
    const uint32_t *src = reinterpret_cast<const uint32_t *>(dir ? bucket + j : all + s);
    uint32_t *dst = reinterpret_cast<uint32_t *>(dir ? all + s : bucket + j);
    for (int i = threadIdx.x; i < (int)(sizeof(StreamState) / 4); i += blockDim.x) dst[i] = src[i];
}

// bump the per-stream chunk counters after a call
// end of a chunk: the streams' chunk counters and the streams' status words, written straight into host-mapped memory (the host reads them after the call's one
// synchronisation: no copy kernel behind the chunk)
static __global__ void advance_chunk_kernel(StreamState *st, int B, int *host_status)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { st[b].chunk += 1; if (host_status) host_status[b] = st[b].status; }
}

// recover_retrieval (engine.hip): the chunk is issued again from the retrieval on, with the counter it had
static __global__ void rewind_chunk_kernel(StreamState *st, int B)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) st[b].chunk -= 1;
}

// ------------------------------------------------------------------------------------
// flat-L2 retrieval (rvc.rs:159 is a TODO; definition in SURVEY.md Appendix A.4).
// Stage 1: every thread owns one index vector (transposed index [dim][n] -> coalesced) and
// accumulates exact sequential-fmaf distances to all queries of its stream; each workgroup
// keeps its own top-4 per query.  Stage 2 merges the per-workgroup candidates
// (ascending (distance, index)), forms w = (1/d)^2 and blends.
// ------------------------------------------------------------------------------------
#define KNN_K 4
#define KNN_MAXQ 16
struct KnnP {
    const float *indexT;     // [dim][n] (or nullptr: the scan walks the row-major index, v_stride = dim, d_stride = 1)
    const float *index;      // [n][dim]
    long long v_stride, d_stride;   // element (vector i, dimension d) of the scanned copy = base[i * v_stride + d * d_stride]
    int n, dim;
    const float *q;          // unique queries, stream stride q_bs
    long long q_bs, cand_bs;
    int nq;
    float *cand_d; int *cand_i;   // [B][nq][nblk][K]
    int nblk;
    const int *overflow;     // when set: run only for streams whose candidate set overflowed (exhaustive fallback)
};

static __global__ __launch_bounds__(256) void knn_scan_kernel(KnnP p)
{
    if (p.overflow && p.overflow[blockIdx.y] == 0) return;
    __shared__ float bd[KNN_MAXQ][4][KNN_K];
    __shared__ int bi[KNN_MAXQ][4][KNN_K];
    const int b = blockIdx.y;
    // the queries are wave-uniform: read through the scalar cache (s_load) so they cost no LDS/VALU bandwidth
    const float *__restrict__ smem = p.q + (long long)b * p.q_bs;
    const int i = blockIdx.x * 256 + threadIdx.x;
    float acc[KNN_MAXQ];
#pragma unroll
    for (int j = 0; j < KNN_MAXQ; j++) acc[j] = 0.f;
    if (i < p.n) {
        // HBM-streaming loop: 8 independent coalesced loads in flight per thread, then the FMAs in ascending-d order
        // (the distance stays a sequential fmaf chain over d, bit-identical to the reference definition)
        // (transposed copy: coalesced across the threads; without one -- a single stream, where this scan only runs for degenerate
        // data -- every thread streams its own row of the row-major index: same arithmetic, same order, bit-identical distances)
        const float *col = (p.indexT ? p.indexT : p.index) + (long long)i * p.v_stride;
        int d = 0;
        for (; d + 8 <= p.dim; d += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = __builtin_nontemporal_load(col + (long long)(d + u) * p.d_stride);
#pragma unroll
            for (int u = 0; u < 8; u++) {
#pragma unroll
                for (int j = 0; j < KNN_MAXQ; j++) if (j < p.nq) { float df = smem[j * p.dim + d + u] - v[u]; acc[j] = fmaf(df, df, acc[j]); }
            }
        }
        for (; d < p.dim; d++) {
            float v = col[(long long)d * p.d_stride];
#pragma unroll
            for (int j = 0; j < KNN_MAXQ; j++) if (j < p.nq) { float df = smem[j * p.dim + d] - v; acc[j] = fmaf(df, df, acc[j]); }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per query: wave-level top-4 by repeated argmin over (distance, index)
    for (int j = 0; j < p.nq; j++) {
        float d = i < p.n ? acc[j] : INFINITY; int id = i < p.n ? i : 0x7fffffff;
        for (int k = 0; k < KNN_K; k++) {
            float md = d; int mi = id;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                float od = __shfl_xor(md, o, 64); int oi = __shfl_xor(mi, o, 64);
                if (od < md || (od == md && oi < mi)) { md = od; mi = oi; }
            }
            if (lane == 0) { bd[j][wave][k] = md; bi[j][wave][k] = mi; }
            if (id == mi) { d = INFINITY; id = 0x7fffffff; }
        }
    }
    __syncthreads();
    // merge the 4 waves' lists: thread j (< nq) does a tiny selection
    if (threadIdx.x < p.nq) {
        const int j = threadIdx.x;
        int pos[4] = {0, 0, 0, 0};
        for (int k = 0; k < KNN_K; k++) {
            float md = INFINITY; int mi = 0x7fffffff, mw = 0;
            for (int w = 0; w < 4; w++) if (pos[w] < KNN_K) {
                float od = bd[j][w][pos[w]]; int oi = bi[j][w][pos[w]];
                if (od < md || (od == md && oi < mi)) { md = od; mi = oi; mw = w; }
            }
            pos[mw]++;
            long long o = (long long)b * p.cand_bs + ((long long)j * p.nblk + blockIdx.x) * KNN_K + k;
            p.cand_d[o] = md; p.cand_i[o] = mi;
        }
    }
}

struct KnnBlendP {
    const float *cand_d; const int *cand_i; int nblk, nq;
    const float *index; int dim;
    const float *q;        // unique queries [B][nq][dim]
    int skip_head, T, R, first_raw;   // sliced frame r uses unique query min((skip_head+r)/2, T-1) - first_raw
    float rate;
    float *phone; int ph_cs; long long ph_bs;
    int *out_idx; float *out_dist;   // [B][R][K]
    const int *overflow;
};

// one workgroup per (unique query, stream): merge candidates, then blend every sliced frame that maps to it
static __global__ __launch_bounds__(256) void knn_merge_blend_kernel(KnnBlendP p)
{
    __shared__ float sd[KNN_K]; __shared__ int si[KNN_K];
    __shared__ float wd[4][KNN_K]; __shared__ int wi[4][KNN_K];
    const int j = blockIdx.x, b = blockIdx.y;
    if (p.overflow && p.overflow[b] == 0) return;
    const long long base = ((long long)b * p.nq + j) * p.nblk * KNN_K;
    const int total = p.nblk * KNN_K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each thread keeps a sorted local top-4 of its strided candidates
    float ld[KNN_K]; int li[KNN_K];
    for (int k = 0; k < KNN_K; k++) { ld[k] = INFINITY; li[k] = 0x7fffffff; }
    for (int c = threadIdx.x; c < total; c += 256) {
        float d = p.cand_d[base + c]; int id = p.cand_i[base + c];
        if (d < ld[KNN_K - 1] || (d == ld[KNN_K - 1] && id < li[KNN_K - 1])) {
            int q = KNN_K - 1;
            while (q > 0 && (d < ld[q - 1] || (d == ld[q - 1] && id < li[q - 1]))) { ld[q] = ld[q - 1]; li[q] = li[q - 1]; q--; }
            ld[q] = d; li[q] = id;
        }
    }
    int pos = 0;
    for (int k = 0; k < KNN_K; k++) {
        float md = pos < KNN_K ? ld[pos] : INFINITY; int mi = pos < KNN_K ? li[pos] : 0x7fffffff;
        float d0 = md; int i0 = mi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float od = __shfl_xor(md, o, 64); int oi = __shfl_xor(mi, o, 64);
            if (od < md || (od == md && oi < mi)) { md = od; mi = oi; }
        }
        if (d0 == md && i0 == mi && mi != 0x7fffffff) pos++;
        if (lane == 0) { wd[wave][k] = md; wi[wave][k] = mi; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int ps[4] = {0, 0, 0, 0};
        for (int k = 0; k < KNN_K; k++) {
            float md = INFINITY; int mi = 0x7fffffff, mw = 0;
            for (int w = 0; w < 4; w++) if (ps[w] < KNN_K) {
                float od = wd[w][ps[w]]; int oi = wi[w][ps[w]];
                if (od < md || (od == md && oi < mi)) { md = od; mi = oi; mw = w; }
            }
            ps[mw]++;
            sd[k] = md; si[k] = mi;
        }
    }
    __syncthreads();
    float w[KNN_K], ws = 0.f;
    for (int k = 0; k < KNN_K; k++) { float inv = 1.0f / sd[k]; w[k] = inv * inv; ws += w[k]; }
    const float *qv = p.q + ((long long)b * p.nq + j) * p.dim;
    for (int r = 0; r < p.R; r++) {
        int s = (p.skip_head + r) / 2; s = s < p.T - 1 ? s : p.T - 1;
        if (s - p.first_raw != j) continue;
        if (threadIdx.x < KNN_K) {
            p.out_idx[((long long)b * p.R + r) * KNN_K + threadIdx.x] = si[threadIdx.x] == 0x7fffffff ? -1 : si[threadIdx.x];
            p.out_dist[((long long)b * p.R + r) * KNN_K + threadIdx.x] = sd[threadIdx.x];
        }
        for (int c = threadIdx.x; c < p.dim; c += 256) {
            float acc = 0.f;
            for (int k = 0; k < KNN_K; k++) if (si[k] >= 0 && si[k] != 0x7fffffff) acc = fmaf(w[k] / ws, p.index[(long long)si[k] * p.dim + c], acc);   // no valid hit for a non-finite query
            p.phone[(long long)b * p.ph_bs + (long long)c * p.ph_cs + r] = fmaf(p.rate, acc, (1.0f - p.rate) * qv[c]);      // (explicit fmaf: the three blend kernels agree bit for bit)
        }
    }
}

// ---- HBM-roofline retrieval: approximate distances on the matrix cores, exact re-rank of a provably sufficient candidate set ----
// Stage A for one stream / few streams lives in knn_scan_select_kernel (below); with many streams it is one implicit GEMM (retrieval.hip).
// compare-exchange of two (distance, index) pairs, ascending, ties by index (deterministic)
__device__ __forceinline__ void knn_cx(float &d0, int &i0, float &d1, int &i1)
{
    const bool sw = d1 < d0 || (d1 == d0 && i1 < i0);
    const float td = sw ? d1 : d0, ud = sw ? d0 : d1; const int ti = sw ? i1 : i0, ui = sw ? i0 : i1;
    d0 = td; i0 = ti; d1 = ud; i1 = ui;
}
// One stream / few streams: the WHOLE retrieval as one launch (round 4; before: knn_queries + knn_dot + knn_select_blend + two idle
// fallback launches = 116 us of kernels for a 307 MB scan).
//  * scan: a workgroup owns tiles of 16 consecutive index vectors (a contiguous 16*dim*4-byte block of HBM in MFMA-fragment order, read
//    exactly once; tiles blockIdx.x, + gridDim.x, ...) against up to 16 queries; its four waves split the K range of every tile (dot
//    products on v_mfma_f32_16x16x4_f32, each wave streams its 12 KB of the tile with the NEXT tile's fragments requested as the slots
//    free up), the partial 16 x 16 tiles meet in LDS and one wave (in turn) forms approx = |y|^2 - 2 x.y and keeps a running top-4 per
//    query.  Splitting K instead of handing whole tiles to waves makes the unit of work a quarter as long: 6 250 tiles over 768
//    workgroups is 8 or 9 each, where 3 072 waves had 2 or 3 (the last third of the launch ran at 3 % occupancy).  The queries are
//    gathered straight from the ContentVec output; NO approximate distance is written: the workgroup publishes 4 {distance, index}
//    granules per query (agent-scope stores, no cache maintenance), then takes a ticket;
//  * select: the last S = min(queries, workgroups) arrivals stay, wait until every list of their stream is published and each runs
//    stages 1-4 of knn_select_blend_kernel for its query: the global top-4 of the approximations is in the union of the workgroups'
//    lists; a workgroup whose 4th entry is inside the margin may hide a 5th candidate ("flagged"): ALL of its vectors become candidates,
//    so no approximation array and no second pass exist, and nothing can overflow (a degenerate index costs exact distances for the
//    flagged workgroups' vectors, 8 per round).  Same candidate superset, same exact re-rank, same hits as the three-launch form.
struct KnnFusedP {
    const float *indexF, *index, *ynorm; int n, dim;
    const float *cv; int cv_cs; long long cv_bs; int first_raw, nq, q0;
    unsigned long long *lists;         // [B][16][gridDim.x][4] granules: low word = distance bits, high word = index
    unsigned *ticket;                  // [B][2]: arrivals, selectors done; zero between launches
    unsigned spin_limit;               // polls of the arrival counter before a selector gives up (ST_KNN_TIMEOUT)
    int test_lose;                     // test hook RVC_KNN_LOSE_TICKET: workgroup 0 of every stream never takes its ticket (a hand-off that cannot complete)
    int skip_head, T, R; float rate;
    float *phone; int ph_cs; long long ph_bs;
    int *out_idx; float *out_dist;
    int *status; int status_stride;
#ifdef RVC_KNN_STAMPS
    long long *stamps;                 // tests/tools/knn_probe.hip: [gridDim.x][16] wall-clock stamps of thread 0
#endif
};
#ifdef RVC_KNN_STAMPS
#define KNN_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.y == 0) p.stamps[blockIdx.x * 16 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define KNN_STAMP(i) do { } while (0)
#endif
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // a 16-byte global load from a dword-aligned address
// minimum over the 64 lanes without LDS-crossbar shuffles: four DPP row rotations (every lane then holds its 16-lane row's minimum), the four
// rows' values through v_readlane.  No NaN may come in.
__device__ __forceinline__ float wave_min_dpp(float v)
{
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false)));      // row_ror:1
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false)));      // row_ror:2
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));      // row_ror:4
    v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));      // row_ror:8
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
#define KNN_FUSED_MAXG 1024            // workgroups per stream (a selector thread keeps 4 workgroup lists in registers)
#define KNN_FUSED_ROWS 10              // candidate rows staged per round of the exact re-rank (<= 16: the final four are found among lanes 0..15)
// dynamic LDS, in floats: scan = query rows + two buffers of partial tiles; select = query row + candidate ids + staged rows
__host__ __device__ inline size_t knn_fused_lds_floats(int dim, int nqg, int G)
{
    const size_t QS = (size_t)dim + 4, scan = (size_t)nqg * QS + 2 * 4 * 256, sel = QS + (((size_t)3 * G + 3) & ~(size_t)3) + (size_t)KNN_FUSED_ROWS * QS;
    return scan > sel ? scan : sel;
}
static __global__ __launch_bounds__(256) void knn_scan_select_kernel(KnnFusedP p)
{
    constexpr int D = 12;
    extern __shared__ __attribute__((aligned(16))) float s_q[];
    __shared__ float wl_d[4][16][KNN_K]; __shared__ int wl_i[4][16][KNN_K];
    __shared__ __attribute__((aligned(16))) float wd[4][KNN_K];
    __shared__ float sd[KNN_K]; __shared__ int si[KNN_K];
    __shared__ float s_red[4];
    __shared__ int s_role, s_cnt, s_dead;
    __shared__ unsigned s_flag[KNN_FUSED_MAXG / 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, G = gridDim.x;
    const int QS = p.dim + 4, nc = p.dim >> 4;
    const int li = lane & 15, kq = lane >> 4;
    const int nqg = p.nq - p.q0 < 16 ? p.nq - p.q0 : 16;
    const int F0 = wave * nc / 4, F = (wave + 1) * nc / 4 - F0;          // this wave's fragments of every tile
    const long long ntile = ((long long)p.n + 15) >> 4;
    long long t = blockIdx.x;
    KNN_STAMP(0);
    // the first tile's fragments are requested before the queries are gathered
    f32x4 a_st[D];
    {
        const float *ar = p.indexF + (t * nc + F0) * 256 + lane * 4;
#pragma unroll
        for (int s = 0; s < D; s++)
            if (s < F && t < ntile) a_st[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(ar + s * 256));
    }
    // queries: s_q[r][c] = cv[c][first_raw + q0 + r], r < nqg.  A lane takes 4 consecutive queries of one channel -- one dword-aligned
    // 16-byte load where all four exist --, 16 channels per wave instruction; LDS writes conflict-free
    const float *cvb = p.cv + (long long)b * p.cv_bs + p.first_raw;
    {
        const int r0 = (lane & 3) * 4;
        const bool whole = r0 + 3 < nqg;
        for (int c = (tid >> 2); c < p.dim; c += 64) {
            const float *src = cvb + (long long)c * p.cv_cs + p.q0;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (whole) v = *reinterpret_cast<const f32x4u *>(src + r0);
            else {
#pragma unroll
                for (int i = 0; i < 4; i++) if (r0 + i < nqg) v[i] = src[r0 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) if (r0 + i < nqg) s_q[(r0 + i) * QS + c] = v[i];
        }
    }
    float *part = s_q + (size_t)nqg * QS;                                 // [2][4 waves][64 lanes][4]
    __syncthreads();
    KNN_STAMP(1);
    float rd[KNN_K]; int ri[KNN_K];
#pragma unroll
    for (int k = 0; k < KNN_K; k++) { rd[k] = INFINITY; ri[k] = 0x7fffffff; }
    // (query columns past the last query of the group repeat it: their results are never read)
    const float *br = s_q + (li < nqg ? li : nqg - 1) * QS + kq * 4;
    for (int it = 0; t < ntile; t += G, it++) {
        const float *ar = p.indexF + (t * nc + F0) * 256 + lane * 4;
        const bool more = t + G < ntile;
        const float *an = p.indexF + ((t + G) * nc + F0) * 256 + lane * 4;
        const int ew = it & 3;                                            // the wave that ranks this tile
        const long long i0 = t * 16;
        // |y|^2 of the lane's four vectors: requested now, consumed behind the dot products
        float yn[4] = {0.f, 0.f, 0.f, 0.f};
        if (wave == ew) {
#pragma unroll
            for (int r = 0; r < 4; r++) { const long long v = i0 + kq * 4 + r; yn[r] = v < p.n ? p.ynorm[v] : 0.f; }
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < F; c += D) {
#pragma unroll
            for (int s = 0; s < D; s++) {
                if (c + s < F) {
                    const f32x4 bq = *reinterpret_cast<const f32x4 *>(br + (F0 + c + s) * 16);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_st[s][0], bq[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_st[s][1], bq[1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_st[s][2], bq[2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_st[s][3], bq[3], acc1, 0, 0, 0);
                    // the wave's fragment stream continues into the workgroup's next tile: after its last use in this tile, slot s takes fragment s of the next
                    if (c + s + D < F) a_st[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(ar + (c + s + D) * 256));
                    else if (more) a_st[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(an + s * 256));
                }
            }
        }
        float *pb = part + (it & 1) * 1024;
        {
            f32x4 a01;
#pragma unroll
            for (int r = 0; r < 4; r++) a01[r] = acc0[r] + acc1[r];
            *reinterpret_cast<f32x4 *>(pb + wave * 256 + lane * 4) = a01;
        }
        __syncthreads();        // (one barrier per tile: buffer it & 1 is rewritten two tiles later, behind the next barrier, which the ranking wave reaches after reading it)
        if (wave == ew) {
            // D layout: row (index vector) = (lane >> 4) * 4 + r, column (query) = lane & 15; partial tiles summed in wave order
            const f32x4 p0 = *reinterpret_cast<const f32x4 *>(pb + lane * 4), p1 = *reinterpret_cast<const f32x4 *>(pb + 256 + lane * 4);
            const f32x4 p2 = *reinterpret_cast<const f32x4 *>(pb + 512 + lane * 4), p3 = *reinterpret_cast<const f32x4 *>(pb + 768 + lane * 4);
            float vd[4]; int vi[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const long long v = i0 + kq * 4 + r;
                const float a = yn[r] - 2.0f * (((p0[r] + p1[r]) + p2[r]) + p3[r]);
                vd[r] = (v < p.n && a == a) ? a : INFINITY;            // (a NaN never is a candidate; as +inf it cannot upset the sorting networks either)
                vi[r] = v < p.n ? (int)v : 0x7fffffff;
            }
            // the tile's 4 smallest per query column: sort the lane's 4, two bitonic merges with the lanes holding the column's other rows
            knn_cx(vd[0], vi[0], vd[1], vi[1]); knn_cx(vd[2], vi[2], vd[3], vi[3]); knn_cx(vd[0], vi[0], vd[2], vi[2]);
            knn_cx(vd[1], vi[1], vd[3], vi[3]); knn_cx(vd[1], vi[1], vd[2], vi[2]);
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                float od[4]; int oi[4];
#pragma unroll
                for (int r = 0; r < 4; r++) { od[r] = __shfl_xor(vd[3 - r], o, 64); oi[r] = __shfl_xor(vi[3 - r], o, 64); }
#pragma unroll
                for (int r = 0; r < 4; r++) { const bool tk = od[r] < vd[r] || (od[r] == vd[r] && oi[r] < vi[r]); vd[r] = tk ? od[r] : vd[r]; vi[r] = tk ? oi[r] : vi[r]; }
                knn_cx(vd[0], vi[0], vd[2], vi[2]); knn_cx(vd[1], vi[1], vd[3], vi[3]); knn_cx(vd[0], vi[0], vd[1], vi[1]); knn_cx(vd[2], vi[2], vd[3], vi[3]);
            }
            // ... merged into the wave's running list (both sorted: min(a[k], b[3 - k]) keeps the 4 smallest, then the bitonic clean-up)
#pragma unroll
            for (int r = 0; r < 4; r++) { const bool tk = vd[3 - r] < rd[r] || (vd[3 - r] == rd[r] && vi[3 - r] < ri[r]); rd[r] = tk ? vd[3 - r] : rd[r]; ri[r] = tk ? vi[3 - r] : ri[r]; }
            knn_cx(rd[0], ri[0], rd[2], ri[2]); knn_cx(rd[1], ri[1], rd[3], ri[3]); knn_cx(rd[0], ri[0], rd[1], ri[1]); knn_cx(rd[2], ri[2], rd[3], ri[3]);
        }
    }
    if (kq == 0) {
#pragma unroll
        for (int k = 0; k < KNN_K; k++) { wl_d[wave][li][k] = rd[k]; wl_i[wave][li][k] = ri[k]; }
    }
    __syncthreads();
    KNN_STAMP(2);
    // the workgroup's list per query: merge of its four waves' lists, published as 4 granules; then the ticket (same wave: program order)
    if (tid < 16) {
        int pos[4] = {0, 0, 0, 0};
        unsigned long long *out = p.lists + (((long long)b * 16 + tid) * G + blockIdx.x) * KNN_K;
#pragma unroll
        for (int k = 0; k < KNN_K; k++) {
            float md = INFINITY; int mi = 0x7fffffff, mw = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int pw = pos[w] < KNN_K ? pos[w] : KNN_K - 1;
                const float od = wl_d[w][tid][pw]; const int oi = wl_i[w][tid][pw];
                if (pos[w] < KNN_K && (od < md || (od == md && oi < mi))) { md = od; mi = oi; mw = w; }
            }
#pragma unroll
            for (int w = 0; w < 4; w++) pos[w] += (w == mw && mi != 0x7fffffff) ? 1 : 0;
            __hip_atomic_store(out + k, ((unsigned long long)(unsigned)mi << 32) | (unsigned long long)__float_as_uint(md), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (wave == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the granules have left (write-through) before the ticket is taken
        // release: the granule stores above are ordered before the ticket; the selectors' acquire load below pairs with it
        if (tid == 0) s_role = (p.test_lose && blockIdx.x == 0) ? 0 : (int)__hip_atomic_fetch_add(p.ticket + b * 2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    KNN_STAMP(3);
    const int S = nqg < G ? nqg : G;
    if (s_role >= 0 && s_role < G - S) return;
    // A role outside [0, G): the counters were left behind by a launch that timed out (they are NOT re-armed on that path, so every later launch on
    // them fails fast here instead of assigning selector roles from stale counts).  The host re-arms them (engine.hip recover_retrieval / plan rebuild).
    const bool stale = s_role < 0 || s_role >= G;
    const int sel = stale ? 0 : s_role - (G - S);
    if (tid == 0) {
        unsigned spins = 0; int dead = stale ? 1 : 0;
        while (!dead && __hip_atomic_load(p.ticket + b * 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
            if (++spins > p.spin_limit) { dead = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (dead) {
            atomicOr(&p.status[b * p.status_stride], (int)ST_KNN_TIMEOUT);
            // poison the arrival counter: whatever is launched on it before the host has re-armed it sees a role beyond G and stops here
            __hip_atomic_store(p.ticket + b * 2, 0x40000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_dead = dead;
    }
    __syncthreads();
    KNN_STAMP(4);
    // select-phase layout of the dynamic LDS: query row | candidate ids (<= 3 per unflagged workgroup) | KNN_FUSED_ROWS staged rows
    float *s_x = s_q;
    int *cand = reinterpret_cast<int *>(s_q + QS);
    float *s_rows = s_q + QS + ((3 * G + 3) & ~3);
    const int nv = p.dim >> 2;
    for (int jq = sel; jq < nqg && !s_dead; jq += S) {
        const int j = p.q0 + jq;
        __syncthreads();
        // the query (exact arithmetic below reads it from LDS) and |x|^2 (only scales the error margin): still in the scan's LDS rows for
        // the selector's first query, gathered again for further ones (the select-phase layout has overwritten them)
        float xn = 0.f;
        if (jq == sel) {
            for (int c = tid; c < p.dim; c += 256) { const float v = s_q[jq * QS + c]; if (jq) s_x[c] = v; xn += v * v; }     // (row jq -> row 0: disjoint)
        } else {
            for (int c = tid; c < p.dim; c += 256) { const float v = cvb[(long long)c * p.cv_cs + j]; s_x[c] = v; xn += v * v; }
        }
        xn = wave_sum(xn);
        if (lane == 0) s_red[wave] = xn;
        if (tid < KNN_FUSED_MAXG / 32) s_flag[tid] = 0u;
        if (tid == 0) s_cnt = 0;
        // this thread's workgroup lists: w = tid + 256 * u
        float ld_[4][KNN_K]; int li_[4][KNN_K];
        const unsigned long long *lq = p.lists + ((long long)b * 16 + jq) * G * KNN_K;
        {
            unsigned long long x[4][KNN_K];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int w = tid + 256 * u;
#pragma unroll
                for (int k = 0; k < KNN_K; k++) {
                    x[u][k] = 0x7fffffff7f800000ull;      // {+inf, no index}
                    if (w < G) x[u][k] = __hip_atomic_load(lq + (long long)w * KNN_K + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
#pragma unroll
                for (int k = 0; k < KNN_K; k++) { ld_[u][k] = __uint_as_float((unsigned)x[u][k]); li_[u][k] = (int)(unsigned)(x[u][k] >> 32); }
            }
        }
#ifdef RVC_KNN_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KNN_STAMP(12);
#endif
        // 1. the 4th smallest approximate distance: per-thread sorted top-4 (static indices only), wave extraction, 4-way merge
        float td[KNN_K];
#pragma unroll
        for (int k = 0; k < KNN_K; k++) td[k] = INFINITY;
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int k = 0; k < KNN_K; k++) {
                const float d = ld_[u][k];
                if (d < td[KNN_K - 1]) {
#pragma unroll
                    for (int q = KNN_K - 1; q >= 1; q--) {
                        const bool left = d < td[q - 1], here = !left && d < td[q];
                        td[q] = left ? td[q - 1] : (here ? d : td[q]);
                    }
                    if (d < td[0]) td[0] = d;
                }
            }
        }
        {
            // the wave's four smallest, with multiplicity: the minimum of the lanes' heads, the lowest lane holding it moves on
            int pos = 0;
            for (int k = 0; k < KNN_K; k++) {
                const float d0 = pos == 0 ? td[0] : pos == 1 ? td[1] : pos == 2 ? td[2] : pos == 3 ? td[3] : INFINITY;
                const float md = wave_min_dpp(d0);
                const unsigned long long holders = __ballot(d0 == md && md < INFINITY);
                if (holders && lane == __ffsll((long long)holders) - 1) pos++;
                if (lane == 0) wd[wave][k] = md;
            }
        }
        KNN_STAMP(13);
        __syncthreads();
        // (every thread merges the four waves' lists itself: 16 broadcast reads instead of a one-thread merge between two barriers)
        float a4;
        {
            float m4[KNN_K];
#pragma unroll
            for (int k = 0; k < KNN_K; k++) m4[k] = INFINITY;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(&wd[w][0]);
#pragma unroll
                for (int k = 0; k < KNN_K; k++) {
                    const float d = wv[k];
                    if (d < m4[KNN_K - 1]) {
#pragma unroll
                        for (int q = KNN_K - 1; q >= 1; q--) {
                            const bool left = d < m4[q - 1], here = !left && d < m4[q];
                            m4[q] = left ? m4[q - 1] : (here ? d : m4[q]);
                        }
                        if (d < m4[0]) m4[0] = d;
                    }
                }
            }
            a4 = m4[KNN_K - 1];
        }
        KNN_STAMP(5);
        // 2. candidates within the error margin of the approximate 4th distance (bound and margin as in knn_select_blend_kernel)
        const float s_xn = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        const float margin = 2e-3f * (fabsf(a4 + s_xn) + s_xn + 1e-3f);
        const float thr = a4 + margin;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int w = tid + 256 * u;
            if (w < G) {
                if (ld_[u][KNN_K - 1] <= thr) atomicOr(&s_flag[w >> 5], 1u << (w & 31));     // may hide a 5th candidate: expanded below
                else {
#pragma unroll
                    for (int k = 0; k < KNN_K - 1; k++) if (ld_[u][k] <= thr) cand[atomicAdd(&s_cnt, 1)] = li_[u][k];
                }
            }
        }
        __syncthreads();
        // 3. exact distances in the reference's order (ascending-d sequential fmaf).  The rows are fetched with coalesced 16-byte loads
        //    and staged in LDS as DIFFERENCES x - y (every thread subtracts what it fetched), so that the one thread per row that walks
        //    the chain issues a single dependent fmaf per dimension; thread r < KNN_FUSED_ROWS keeps a sorted top-4 of what it has seen
        float bd[KNN_K]; int bi[KNN_K];
#pragma unroll
        for (int k = 0; k < KNN_K; k++) { bd[k] = INFINITY; bi[k] = 0x7fffffff; }
        auto round = [&](auto row_of, int nrows) {
            __syncthreads();
            for (int i = tid; i < nrows * nv; i += 256) {
                const int r = i / nv, c4 = i - r * nv;
                const f32x4 y = *reinterpret_cast<const f32x4 *>(p.index + (long long)row_of(r) * p.dim + c4 * 4);
                *reinterpret_cast<f32x4 *>(s_rows + r * QS + c4 * 4) = *reinterpret_cast<const f32x4 *>(s_x + c4 * 4) - y;
            }
            __syncthreads();
            KNN_STAMP(11);
            if (tid < nrows) {
                const float *v = s_rows + tid * QS;
                float acc = 0.f;
#ifdef RVC_KNN_STAMPS
                const long long cyc0 = clock64();
#endif
                // two register sets of 32 differences each: one is consumed while the other is on its way from LDS.  Dimensions that are a
                // multiple of 64 (768, 256) take the loop without per-group guards (the guards compile to a select and five scalar
                // instructions per four dimensions, in the middle of the dependent chain)
                f32x4 ra[8], rb[8];
                if ((p.dim & 63) == 0) {
#pragma unroll
                    for (int u = 0; u < 8; u++) ra[u] = *reinterpret_cast<const f32x4 *>(v + 4 * u);
                    for (int d = 0; d < p.dim; d += 64) {
#pragma unroll
                        for (int u = 0; u < 8; u++) rb[u] = *reinterpret_cast<const f32x4 *>(v + d + 32 + 4 * u);
#pragma unroll
                        for (int u = 0; u < 8; u++) {
#pragma unroll
                            for (int i = 0; i < 4; i++) acc = fmaf(ra[u][i], ra[u][i], acc);
                        }
                        if (d + 64 < p.dim) {
#pragma unroll
                            for (int u = 0; u < 8; u++) ra[u] = *reinterpret_cast<const f32x4 *>(v + d + 64 + 4 * u);
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++) {
#pragma unroll
                            for (int i = 0; i < 4; i++) acc = fmaf(rb[u][i], rb[u][i], acc);
                        }
                    }
                } else {
                    auto fetch = [&](f32x4 (&r)[8], int d) {
#pragma unroll
                        for (int u = 0; u < 8; u++) r[u] = *reinterpret_cast<const f32x4 *>(v + (d + 4 * u < p.dim ? d + 4 * u : 0));
                    };
                    auto chain = [&](const f32x4 (&r)[8], int d) {
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            if (d + 4 * u < p.dim) {
#pragma unroll
                                for (int i = 0; i < 4; i++) acc = fmaf(r[u][i], r[u][i], acc);
                            }
                        }
                    };
                    fetch(ra, 0);
                    for (int d = 0; d < p.dim; d += 64) {
                        fetch(rb, d + 32);
                        chain(ra, d);
                        fetch(ra, d + 64);
                        chain(rb, d + 32);
                    }
                }
#ifdef RVC_KNN_STAMPS
                if (tid == 0 && blockIdx.y == 0) p.stamps[blockIdx.x * 16 + 1] = clock64() - cyc0 + (acc == 1.2345f ? 1 : 0);
#endif
                float cd = acc; int ci = row_of(tid);
                // sorted insert by (distance, index); a NaN distance never enters
#pragma unroll
                for (int k = 0; k < KNN_K; k++) {
                    const bool sw = cd < bd[k] || (cd == bd[k] && ci < bi[k]);
                    const float t0 = sw ? bd[k] : cd; const int t1 = sw ? bi[k] : ci;
                    bd[k] = sw ? cd : bd[k]; bi[k] = sw ? ci : bi[k];
                    cd = t0; ci = t1;
                }
            }
        };
        const int ncand = s_cnt;
        KNN_STAMP(10);
        for (int base = 0; base < ncand; base += KNN_FUSED_ROWS) {
            const int nr = ncand - base < KNN_FUSED_ROWS ? ncand - base : KNN_FUSED_ROWS;
            round([&](int r) { return cand[base + r]; }, nr);
        }
        for (int fw = 0; fw < (G + 31) / 32; fw++) {
            unsigned m = s_flag[fw];
            while (m) {
                const int w = fw * 32 + __builtin_ctz(m); m &= m - 1;
                // every vector the flagged workgroup scanned: tiles w, w + G, ...
                for (long long t0 = (long long)w * 16; t0 < p.n; t0 += (long long)G * 16) {
                    const int nt = p.n - t0 < 16 ? (int)(p.n - t0) : 16;
                    for (int base = 0; base < nt; base += KNN_FUSED_ROWS) {
                        const int nr = nt - base < KNN_FUSED_ROWS ? nt - base : KNN_FUSED_ROWS;
                        round([&](int r) { return (int)t0 + base + r; }, nr);
                    }
                }
            }
        }
        KNN_STAMP(6);
        // the final four: wave 0 holds every list (threads < KNN_FUSED_ROWS); four rounds of a minimum by (distance, index) over those lanes
        if (wave == 0) {
            int pos = 0;
            for (int k = 0; k < KNN_K; k++) {
                float md = pos == 0 ? bd[0] : pos == 1 ? bd[1] : pos == 2 ? bd[2] : pos == 3 ? bd[3] : INFINITY;
                int mi = pos == 0 ? bi[0] : pos == 1 ? bi[1] : pos == 2 ? bi[2] : pos == 3 ? bi[3] : 0x7fffffff;
                int ml = lane;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    const float od = __shfl_xor(md, o, 64); const int oi = __shfl_xor(mi, o, 64), ol = __shfl_xor(ml, o, 64);
                    if (oi != 0x7fffffff && (mi == 0x7fffffff || od < md || (od == md && oi < mi))) { md = od; mi = oi; ml = ol; }
                }
                if (ml == lane && mi != 0x7fffffff) pos++;
                if (lane == 0) { sd[k] = mi == 0x7fffffff ? INFINITY : md; si[k] = mi; }
            }
        }
        __syncthreads();
        KNN_STAMP(8);
        // 4. blend: w = (1/d)^2 normalised, feat = rate * sum w_i y_i + (1 - rate) * feat; computed once per
        //    query, written to every sliced frame that duplicates it (a contiguous range of r: frames (skip_head + r) / 2, clamped to T - 1)
        float wn[KNN_K], ws = 0.f;
#pragma unroll
        for (int k = 0; k < KNN_K; k++) { const float inv = 1.0f / sd[k]; wn[k] = inv * inv; ws += wn[k]; }
#pragma unroll
        for (int k = 0; k < KNN_K; k++) wn[k] = wn[k] / ws;
        const int raw = j + p.first_raw;
        int r_lo = 2 * raw - p.skip_head, r_hi = raw >= p.T - 1 ? p.R : 2 * raw + 2 - p.skip_head;
        r_lo = r_lo < 0 ? 0 : r_lo; r_hi = r_hi > p.R ? p.R : r_hi;
        float *ph = p.phone + (long long)b * p.ph_bs;
        for (int c = tid; c < p.dim; c += 256) {
            float y[KNN_K];
#pragma unroll
            for (int k = 0; k < KNN_K; k++) y[k] = (si[k] >= 0 && si[k] != 0x7fffffff) ? p.index[(long long)si[k] * p.dim + c] : 0.f;   // (no valid hit for a non-finite query)
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < KNN_K; k++) if (si[k] >= 0 && si[k] != 0x7fffffff) acc = fmaf(wn[k], y[k], acc);
            const float val = fmaf(p.rate, acc, (1.0f - p.rate) * s_x[c]);
            for (int r = r_lo; r < r_hi; r++) ph[(long long)c * p.ph_cs + r] = val;
        }
        KNN_STAMP(9);
        if (tid < KNN_K)
            for (int r = r_lo; r < r_hi; r++) {
                p.out_idx[((long long)b * p.R + r) * KNN_K + tid] = si[tid] == 0x7fffffff ? -1 : si[tid];   // -1: no hit (non-finite query)
                p.out_dist[((long long)b * p.R + r) * KNN_K + tid] = sd[tid];
            }
    }
    __syncthreads();
    KNN_STAMP(7);
    // the last selector to leave re-arms the counters for the next launch (never after a time-out: a late workgroup may still take a ticket)
    if (tid == 0 && !s_dead) {
        const unsigned d = __hip_atomic_fetch_add(p.ticket + b * 2 + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)d == S - 1) {
            __hip_atomic_store(p.ticket + b * 2 + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.ticket + b * 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// |y_i|^2 for every index vector (load time); nhn = -|y_i|^2 / 2 is the per-column "residual" of the many-stream distance GEMM
static __global__ void knn_norms_kernel(const float *index, int n, int dim, float *ynorm, float *nhn)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = index + (long long)i * dim;
    float s = 0.f;
    for (int d = 0; d < dim; d++) s = fmaf(r[d], r[d], s);
    ynorm[i] = s;
    nhn[i] = -0.5f * s;
}

// Load-time repack of the index on the device (the matrix arrives in HBM by upload or by the RCCL broadcast and never goes back to
// the host): [n][dim] -> MFMA-fragment order [tile of 16 vectors][chunk of 16 dims][lane][4] for knn_scan_select_kernel (vectors past n zero).
// One thread per float4 of the output; reads are 16-byte pieces of 16 neighbouring rows.
static __global__ __launch_bounds__(256) void knn_pack_index_kernel(const float *index, long long n, int dim, float *indexF, long long total4)
{
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total4) return;
    const int nc = dim / 16;
    const int l = (int)(o & 63);
    const long long tc = o >> 6, tl = tc / nc;
    const int c = (int)(tc - tl * nc);
    const long long v = tl * 16 + (l & 15);
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    if (v < n) val = *reinterpret_cast<const f32x4 *>(index + v * dim + c * 16 + (l >> 4) * 4);
    *reinterpret_cast<f32x4 *>(indexF + o * 4) = val;
}
// [n][dim] -> [dim][n] through a 32 x 33 LDS tile (only plans that need the transposed copy build it: the many-stream distance GEMM
// and the forced exhaustive scan)
static __global__ __launch_bounds__(256) void knn_transpose_kernel(const float *index, long long n, int dim, float *indexT)
{
    __shared__ float tile[32][33];
    const long long v0 = (long long)blockIdx.x * 32; const int d0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) { const long long v = v0 + r; const int d = d0 + tx; tile[r][tx] = (v < n && d < dim) ? index[v * dim + d] : 0.f; }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) { const int d = d0 + r; const long long v = v0 + tx; if (d < dim && v < n) indexT[(long long)d * n + v] = tile[tx][r]; }
}

// Many streams: the queries of all streams as the WEIGHT operand of one implicit GEMM against the transposed index (approx[q][i] =
// |y_i|^2 - 2 x_q . y_i for every stream's queries in ONE pass over the index instead of one pass per 16 queries): [Q][dim] ->
// MFMA-fragment order [tile of 16 queries][chunk of 16 dims][lane][4], rows past Q zero.  grid = (Qpad / 16, dim / 16), 64 threads.
static __global__ __launch_bounds__(64) void knn_pack_queries_kernel(const float *q, int Q, int dim, float *qf)
{
    const int t = blockIdx.x, c = blockIdx.y, l = threadIdx.x, v = t * 16 + (l & 15);
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (v < Q) x = *reinterpret_cast<const f32x4 *>(q + (long long)v * dim + c * 16 + (l >> 4) * 4);
    *reinterpret_cast<f32x4 *>(qf + (((long long)t * gridDim.y + c) * 64 + l) * 4) = x;
}

// Stage B (knn_select_blend_kernel): one workgroup per (unique query, stream).
//  1. exact top-4 of the APPROXIMATE distances -> 4th smallest a4;
//  2. candidate set = { i : approx_i <= a4 + margin }: since |approx - true| <= err < margin/2, every vector of the true top-4
//     (true distance <= true 4th distance <= a4 + err) is in the set;
//  3. exact sequential-fmaf distances (the reference definition) for the candidates, final order by (distance, index);
//  4. w = (1/d)^2 blend of the duplicated frames (as before).
// More than KNN_CAND candidates (degenerate data, e.g. thousands of duplicate vectors): overflow[b] is raised and the
// exhaustive exact scan (knn_scan_kernel + knn_merge_blend_kernel) recomputes this stream.
#define KNN_CAND 512
struct KnnSelP {
    const float *approx; long long approx_bs; int n, dim, nq;
    const float *index; const float *q; long long q_bs;
    int skip_head, T, R, first_raw; float rate;
    float *phone; int ph_cs; long long ph_bs;
    int *out_idx; float *out_dist; int *overflow;
};
static __global__ __launch_bounds__(1024) void knn_select_blend_kernel(KnnSelP p)
{
    __shared__ float wd[16][KNN_K]; __shared__ int wi[16][KNN_K];
    __shared__ float sd[KNN_K]; __shared__ int si[KNN_K];
    __shared__ int cand_i[KNN_CAND]; __shared__ float cand_d[KNN_CAND];
    __shared__ int cnt; __shared__ float s_xn;
    extern __shared__ __attribute__((aligned(16))) float s_rows[];     // [32][dim + 4] candidate rows + the query
    const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *a = p.approx + (long long)b * p.approx_bs + (long long)j * p.n;
    const float *qv = p.q + (long long)b * p.q_bs + (long long)j * p.dim;
    if (tid == 0) cnt = 0;
    // |x|^2 (only to scale the error margin)
    float xn = 0.f;
    for (int d = tid; d < p.dim; d += 1024) xn += qv[d] * qv[d];
    xn = wave_sum(xn);
    if (lane == 0) wd[wave][0] = xn;
    __syncthreads();
    if (tid == 0) { float t = 0.f; for (int w = 0; w < 16; w++) t += wd[w][0]; s_xn = t; }
    __syncthreads();
    // 1. per-thread sorted top-4 of the approximate distances
    float ld[KNN_K]; int li_[KNN_K];
#pragma unroll
    for (int k = 0; k < KNN_K; k++) { ld[k] = INFINITY; li_[k] = 0x7fffffff; }
    // sorted insert with static indices only (a `while (q > 0 && d < ld[q - 1])` walk indexes the arrays dynamically, which puts them
    // in scratch memory): slot q takes its left neighbour if d belongs further left, d itself if it belongs here
    auto keep = [&](float d, int i) {
        if (d < ld[KNN_K - 1]) {
#pragma unroll
            for (int q = KNN_K - 1; q >= 1; q--) {
                const bool left = d < ld[q - 1], here = !left && d < ld[q];
                ld[q] = left ? ld[q - 1] : (here ? d : ld[q]);
                li_[q] = left ? li_[q - 1] : (here ? i : li_[q]);
            }
            if (d < ld[0]) { ld[0] = d; li_[0] = i; }
        }
    };
    // the scan of the n approximate distances: 16-byte loads, four of them in flight per thread (one load per iteration and a
    // data-dependent branch behind it made this pass ~100 dependent round trips: 80 us for 100 k vectors, more than the scan that
    // produced the distances).  Only the VALUE of the 4th smallest is used below, so the visiting order does not matter.
    const int n4 = ((p.n & 3) == 0 && (reinterpret_cast<size_t>(a) & 15) == 0) ? p.n >> 2 : 0;
    for (int i4 = tid; i4 < n4; i4 += 4 * 1024) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int g = i4 + u * 1024; v[u] = *reinterpret_cast<const f32x4 *>(a + 4 * (g < n4 ? g : i4)); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = i4 + u * 1024;
            if (g < n4) {
#pragma unroll
                for (int e = 0; e < 4; e++) keep(v[u][e], 4 * g + e);
            }
        }
    }
    for (int i = 4 * n4 + tid; i < p.n; i += 1024) keep(a[i], i);
    int pos = 0;
    for (int k = 0; k < KNN_K; k++) {
        float md = pos < KNN_K ? ld[pos] : INFINITY; int mi = pos < KNN_K ? li_[pos] : 0x7fffffff;
        const float d0 = md; const int i0 = mi;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float od = __shfl_xor(md, o, 64); int oi = __shfl_xor(mi, o, 64);
            if (od < md || (od == md && oi < mi)) { md = od; mi = oi; }
        }
        if (d0 == md && i0 == mi && mi != 0x7fffffff) pos++;
        if (lane == 0) { wd[wave][k] = md; wi[wave][k] = mi; }
    }
    __syncthreads();
    if (tid == 0) {
        int ps[16];
        for (int w = 0; w < 16; w++) ps[w] = 0;
        float last = INFINITY;
        for (int k = 0; k < KNN_K; k++) {
            float md = INFINITY; int mi = 0x7fffffff, mw = 0;
            for (int w = 0; w < 16; w++) if (ps[w] < KNN_K) {
                float od = wd[w][ps[w]]; int oi = wi[w][ps[w]];
                if (od < md || (od == md && oi < mi)) { md = od; mi = oi; mw = w; }
            }
            ps[mw]++;
            last = md;
        }
        sd[0] = last;      // approximate 4th-smallest
    }
    __syncthreads();
    // 2. candidates within the error margin of the approximate 4th distance.  fp32 error of approx is bounded by
    //    ~dim*2^-24*(|y|^2 + 2|x||y|) <= 1e-4*(|x|^2 + |y|^2) for dim <= 1024; the margin is 20x that.
    const float a4 = sd[0];
    const float margin = 2e-3f * (fabsf(a4 + s_xn) + s_xn + 1e-3f);
    const float thr = a4 + margin;
    __syncthreads();
    // The candidates are among the per-thread top-4 lists unless some thread holds MORE than four values within the margin (its list is
    // then truncated: its 4th entry is still <= thr).  Common case: collect from the lists, no second pass over the n distances.
    if (ld[KNN_K - 1] <= thr) atomicOr(&cnt, 0x40000000);       // (a truncated list sends the stream through the full pass below)
    __syncthreads();
    const bool truncated = (cnt & 0x40000000) != 0;
    __syncthreads();
    if (tid == 0) cnt = 0;
    __syncthreads();
    if (!truncated) {
#pragma unroll
        for (int k = 0; k < KNN_K; k++)
            if (ld[k] <= thr) { int c = atomicAdd(&cnt, 1); if (c < KNN_CAND) cand_i[c] = li_[k]; }
    } else {
    for (int i4 = tid; i4 < n4; i4 += 4 * 1024) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int g = i4 + u * 1024; v[u] = *reinterpret_cast<const f32x4 *>(a + 4 * (g < n4 ? g : i4)); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int g = i4 + u * 1024;
            if (g < n4) {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (v[u][e] <= thr) { int c = atomicAdd(&cnt, 1); if (c < KNN_CAND) cand_i[c] = 4 * g + e; }
            }
        }
    }
    for (int i = 4 * n4 + tid; i < p.n; i += 1024) {
        if (a[i] <= thr) { int c = atomicAdd(&cnt, 1); if (c < KNN_CAND) cand_i[c] = i; }
    }
    }
    __syncthreads();
    const int ncand = cnt;
    if (ncand > KNN_CAND) { if (tid == 0) p.overflow[b] = 1; return; }
    // 3. exact distances in the reference's order (ascending-d sequential fmaf), one candidate per thread; the rows are first
    //    staged in LDS with coalesced 16-byte loads (32 candidates per round) so the dependent chain never waits on HBM
    {
        const int RS = p.dim + 4, nv = p.dim >> 2;
        float *s_qv = s_rows + 32 * RS;
        for (int i = tid; i < nv; i += 1024) *reinterpret_cast<f32x4 *>(s_qv + i * 4) = *reinterpret_cast<const f32x4 *>(qv + i * 4);
        for (int base = 0; base < ncand; base += 32) {
            __syncthreads();
            for (int i = tid; i < 32 * nv; i += 1024) {
                const int r = i / nv, c4 = i - r * nv;
                if (base + r < ncand)
                    *reinterpret_cast<f32x4 *>(s_rows + r * RS + c4 * 4) = *reinterpret_cast<const f32x4 *>(p.index + (long long)cand_i[base + r] * p.dim + c4 * 4);
            }
            __syncthreads();
            if (tid < 32 && base + tid < ncand) {
                const float *v = s_rows + tid * RS;
                float acc = 0.f;
                for (int d = 0; d < p.dim; d++) { const float df = s_qv[d] - v[d]; acc = fmaf(df, df, acc); }
                cand_d[base + tid] = acc;
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < KNN_K; k++) {
            float md = INFINITY; int mi = 0x7fffffff, mc = -1;
            for (int c = 0; c < ncand; c++) {
                const float od = cand_d[c]; const int oi = cand_i[c];
                if (oi >= 0 && (od < md || (od == md && oi < mi))) { md = od; mi = oi; mc = c; }
            }
            sd[k] = md; si[k] = mi;
            if (mc >= 0) cand_i[mc] = -1;
        }
    }
    __syncthreads();
    // 4. blend: w = (1/d)^2 normalised, feat = rate * sum w_i y_i + (1 - rate) * feat
    float w[KNN_K], ws = 0.f;
#pragma unroll
    for (int k = 0; k < KNN_K; k++) { float inv = 1.0f / sd[k]; w[k] = inv * inv; ws += w[k]; }
    for (int r = 0; r < p.R; r++) {
        int s = (p.skip_head + r) / 2; s = s < p.T - 1 ? s : p.T - 1;
        if (s - p.first_raw != j) continue;
        if (tid < KNN_K) {
            p.out_idx[((long long)b * p.R + r) * KNN_K + tid] = si[tid] == 0x7fffffff ? -1 : si[tid];   // -1: no hit (non-finite query)
            p.out_dist[((long long)b * p.R + r) * KNN_K + tid] = sd[tid];
        }
        for (int c = tid; c < p.dim; c += 1024) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < KNN_K; k++) if (si[k] >= 0 && si[k] != 0x7fffffff) acc = fmaf(w[k] / ws, p.index[(long long)si[k] * p.dim + c], acc);   // no valid hit for a non-finite query
            p.phone[(long long)b * p.ph_bs + (long long)c * p.ph_cs + r] = fmaf(p.rate, acc, (1.0f - p.rate) * qv[c]);      // (explicit fmaf: the three blend kernels agree bit for bit)
        }
    }
}

// unique query rows for retrieval: q[j][c] = cv[c][first_raw + j]
static __global__ void knn_queries_kernel(const float *cv, int cv_cs, long long cv_bs, int C, int first_raw, int nq, float *q)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (i >= nq * C) return;
    int j = i / C, c = i - j * C;
    q[(long long)b * nq * C + i] = cv[(long long)b * cv_bs + (long long)c * cv_cs + first_raw + j];
}

// ------------------------------------------------------------------------------------
// caller-side post-processing (reference: obs-rvc/src/rt_utils.rs:60-132, lib.rs:758-794)
// ------------------------------------------------------------------------------------
// rt_utils.rs:94-103: zero-pad frame/2, square, windowed mean (window frame, step hop), sqrt.  One workgroup per frame.
// (all post-processing kernels take a stream index in blockIdx.y -- blockIdx.x for post_sola_kernel -- and per-stream strides)
static __global__ __launch_bounds__(256) void post_rms_kernel(const float *y, int n, int frame, int hop, float *out, long long y_bs, long long out_bs)
{
    __shared__ float red[16];
    y += blockIdx.y * y_bs; out += blockIdx.y * out_bs;
    const int f = blockIdx.x, pad = frame / 2;
    float s = 0.f;
    for (int j = threadIdx.x; j < frame; j += 256) {
        int q = f * hop + j - pad;
        float v = (q >= 0 && q < n) ? y[q] : 0.f;
        s += v * v;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[f] = sqrtf(s / (float)frame);
}
// rt_utils.rs:105-117 evaluated at one index of the (size)-point output
__device__ __forceinline__ float lerp_align_corners_at(const float *in, int n_in, int size, int i)
{
    const float step = (float)(n_in - 1) / (float)(size - 1);
    const float idx = (float)i * step;
    int fl = (int)floorf(idx), ce = (int)ceilf(idx);
    fl = fl < 0 ? 0 : (fl > n_in - 1 ? n_in - 1 : fl);
    ce = ce < 0 ? 0 : (ce > n_in - 1 ? n_in - 1 : ce);
    const float fr = idx - (float)fl;
    return in[fl] * (1.0f - fr) + in[ce] * fr;
}
// rt_utils.rs:119-132
// mix_power_v: per-stream exponent (or nullptr: mix_power for every stream); an exponent of 0 leaves the stream untouched (powf(x, 0) = 1)
static __global__ void post_mix_kernel(float *out, int n, const float *r1, int n1, const float *r2, int n2, float mix_power, long long out_bs, long long r_bs, const float *mix_power_v)
{
    if (mix_power_v) mix_power = mix_power_v[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out += blockIdx.y * out_bs; r1 += blockIdx.y * r_bs; r2 += blockIdx.y * r_bs;
    const float a = lerp_align_corners_at(r1, n1, n + 1, i);
    const float b = fmaxf(lerp_align_corners_at(r2, n2, n + 1, i), 1e-3f);
    out[i] = out[i] * powf(a / b, mix_power);
}
// rt_utils.rs:60-90 + lib.rs:768-794 in one workgroup: normalised cross-correlation over search+1 lags (last maximum wins),
// sin^2 crossfade with the previous tail, new tail saved, first `frame` samples returned.
// normalised cross-correlation of get_sola_offset (rt_utils.rs:60-77), one wave per lag: cor[l] = <out[l..], sola> / sqrt(<out[l..], out[l..]> + 1e-8)
// with f64 accumulation (the reference's FFT convolution carries f32 rounding noise of the same order as an f32 direct sum)
static __global__ __launch_bounds__(256) void post_sola_corr_kernel(const float *output, const float *sola, int sola_len, int search, float *cor,
                                                             long long out_bs, long long sola_bs, long long cor_bs)
{
    const int lane = threadIdx.x & 63, l = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (l > search) return;
    output += blockIdx.y * out_bs; sola += blockIdx.y * sola_bs; cor += blockIdx.y * cor_bs;
    double nom = 0.0, den = 0.0;
    for (int j = lane; j < sola_len; j += 64) { const double v = (double)output[l + j]; nom += v * (double)sola[j]; den += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nom += __shfl_xor(nom, o, 64); den += __shfl_xor(den, o, 64); }
    if (lane == 0) cor[l] = (float)nom / sqrtf((float)den + 1e-8f);
}

// arg-max with the reference's tie rule (the LAST maximum wins, rt_utils.rs:79-88), sin^2 crossfade with the previous tail, tail save and
// frame extraction (lib.rs:768-794)
static __global__ __launch_bounds__(1024) void post_sola_kernel(float *output, float *sola, int sola_len, int search, int frame, float *frame_out, int *offset_out,
                                                         const float *cor_g, long long out_bs, long long sola_bs, long long frame_bs, long long cor_bs)
{
    __shared__ float cor[1024];
    __shared__ int s_off;
    const int t = threadIdx.x;
    output += blockIdx.x * out_bs; sola += blockIdx.x * sola_bs; frame_out += blockIdx.x * frame_bs; cor_g += blockIdx.x * cor_bs; offset_out += blockIdx.x;
    for (int l = t; l <= search; l += 1024) cor[l] = cor_g[l];
    __syncthreads();
    if (t == 0) {
        int best = 0; float bv = cor[0];
        for (int l = 1; l <= search; l++) if (!(bv > cor[l])) { best = l; bv = cor[l]; }
        s_off = best; *offset_out = best;
    }
    __syncthreads();
    float *o = output + s_off;
    for (int i = t; i < sola_len; i += 1024) {
        const float x = sola_len > 1 ? (float)i / (float)(sola_len - 1) : 0.f;
        const float sn = sinf(x * 0.5f * 3.14159265358979323846f);
        const float fi = sn * sn, fo = 1.0f - fi;
        o[i] = o[i] * fi + sola[i] * fo;
    }
    __syncthreads();
    for (int i = t; i < sola_len; i += 1024) sola[i] = o[frame + i];
    for (int i = t; i < frame; i += 1024) frame_out[i] = o[i];
}

}  // namespace rvc
