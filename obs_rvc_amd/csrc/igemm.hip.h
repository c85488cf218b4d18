// igemm.hip.h -- the implicit-GEMM kernel family (templates only: every instantiation lives in igemm2_inst.hip / igemm_tiled_inst.hip,
// so that the library builds as parallel translation units).  Part of the hand-written gfx950 (CDNA4) kernels for the RVC per-chunk hot path.
//
// Layout convention: every activation is channel-major [B][C][ld] with the time (or H*W)
// axis contiguous and a zero halo on both sides of every row, so convolution taps never
// need bounds checks (the halo is zeroed once at allocation and never written).
//
// The dense work (every Conv1d / ConvTranspose1d / Conv2d / ConvTranspose2d / Linear of
// ContentVec, RMVPE and the NSF-HiFiGAN synthesizer) runs through ONE implicit-GEMM
// kernel on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32, 157 TF peak):
//   D[m][n] = sum_k W[m][k] * X[koff[k] + noff(n)]
// where koff[] is a per-layer table of input offsets (channel stride, tap, dilation) and
// noff(n) is the per-lane offset of output position n.  Transposed convolutions are run
// as `stride` polyphase sub-convolutions ("phases"), grouped convolutions as one phase
// per group.  64-wide wavefronts: one wave owns a (16*MF) x (16*NF) output tile.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rvc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_GELU = 3, ACT_TANH = 4, ACT_SIGMOID = 5 };

// GELU(erf) in ONE place (every kernel's epilogue and the fused ContentVec stem round the same way).
#ifdef RVC_FAST_GELU
// (round-6 experiment, never the default: erf as the 13 / 9-term rational polynomial of Eigen / XLA -- 15 vector instructions against libm's ~34 with both of its
//  branches taken inside a wave -- at 4e-7 absolute error instead of 3e-8)
__device__ __forceinline__ float erf_dev(float x)
{
    x = fminf(fmaxf(x, -4.f), 4.f);
    const float x2 = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, x2, 2.77068142495902e-08f); p = fmaf(p, x2, -2.10102402082508e-06f); p = fmaf(p, x2, -5.69250639462346e-05f);
    p = fmaf(p, x2, -7.34990630326855e-04f); p = fmaf(p, x2, -2.95459980854025e-03f); p = fmaf(p, x2, -1.60960333262415e-02f);
    float q = -1.45660718464996e-05f;
    q = fmaf(q, x2, -2.13374055278905e-04f); q = fmaf(q, x2, -1.68282697438203e-03f); q = fmaf(q, x2, -7.37332916720468e-03f); q = fmaf(q, x2, -1.42647390514189e-02f);
    return x * p * __builtin_amdgcn_rcpf(q);
}
#else
__device__ __forceinline__ float erf_dev(float x) { return erff(x); }
#endif
__device__ __forceinline__ float gelu_dev(float v) { return 0.5f * v * (1.0f + erf_dev(v * 0.70710678118654752440f)); }

__device__ __forceinline__ float apply_act(float v, int act, float slope)
{
    switch (act) {
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_LRELU: return v > 0.f ? v : v * slope;
    case ACT_GELU: return gelu_dev(v);
    case ACT_TANH: return tanhf(v);
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
    }
}

struct PhaseD {
    long long w_off;   // element offset of this phase's [M][Kp] weight panel
    int x_off;         // element offset added to the input base
    int y_c0;          // first output channel of this phase (grouped convs)
    int y_h0;          // output row of nh = 0 (polyphase 2-D transposed convs)
    int y_pos;         // output column of nw = 0 (polyphase transposed convs); checked against [0, OW)
    int bias_off;      // offset into bias
    int koff_off;      // offset into the koff table
    int nchunks;       // K/16 of this phase (fused launches of convs with different kernel sizes; filled by queue_igemm)
    // igemm2 only (register-direct kernel, launches that fuse two DIFFERENT convolutions of one input: RMVPE's 3x3 + shortcut):
    int act_p1;        // this phase's epilogue activation + 1 (0: the launch's IgemmP::act)
    int pad_;
    long long y_off;   // element offset of this phase's output tensor from IgemmP::y
    // conv_tile_kernel only (conv_tile.hip.h): the staged input tile of this phase
    int t_tab;         // offset of the phase's LDS-offset table in IgemmP::ttab (ints; padded to whole 256-int pieces)
    int t_cin;         // input rows staged
    int t_rs;          // LDS row stride (floats) = BN + reach of the taps, padded
    int t_dmin;        // column of the leftmost tap relative to the output column (<= 0)
};

struct IgemmP {
    const float *x, *w, *bias, *res;
    float *y, *part;
    const int *koff;
    const PhaseD *ph;
    PhaseD ph0;              // copy of ph[0]: single-phase layers skip the dependent table load
    int M, N, K;             // K already padded to a multiple of 16
    int NW;                  // n -> (nh, nw) = (n / NW, n % NW)
    int x_hs, x_ws;          // input offset of position n  = nh*x_hs + nw*x_ws
    int y_hm, y_ws;          // output coordinates of position n: row = nh*y_hm + y_h0, col = nw*y_ws + y_pos
    int OW;                  // valid iff 0 <= col < OW
    long long x_bs, y_bs, res_bs;
    int y_cs, res_cs;        // channel strides of the output / residual tensors
    int y_rs, res_rs;        // row strides (0 for 1-D tensors)
    int nphase, ksplit, chunks_per_split;
    int act; float slope; float scale; int accumulate;
    int pre_act; float pre_slope;   // fused input LeakyReLU: x -> max(x, x*pre_slope); pre_slope = 1 disables it
    int ntn, ntm;
    int koff_bias;           // bytes: the koff table holds (offset - min offset) * 4, the base pointer is moved back by this
    int glu;                 // WaveNet gate fused into the epilogue: rows are GLU-packed (see glu_store), output has M/2 channels
    int res_nogroup;         // residual channel = m (a tensor shared by all phases) instead of m + y_c0
    unsigned long long *probe;   // tuning build only (-DRVC_KPROBE): per-wave phase timestamps
    int m_fast;              // XCD-aware tile order: >0 = ntm rounded up to 8, m fastest (workgroup b runs on XCD b%8, so all
                             // n-tiles of one weight-row block share one XCD's L2); 0 = n fastest (activation-heavy layers)
    int nbatch;              // igemm2: streams in the launch (grid z = batch * nphase + phase)
    int fold_n;              // > 0: the streams of the launch are folded into the N axis: position n = stream (n / fold_n), local position
                             // (n % fold_n); N = streams * fold_n and the launch has one batch (tiles may straddle streams, nothing is padded per stream)
    int bf3;                 // exploratory (igemm_bf3_kernel): w points at the split-bf16 panels of this layer, the products run as three bf16 MFMAs
    int lin_cs4;             // igemm2 LIN layers (1x1 conv on a 1-D tensor): input channel stride in BYTES, k-th operand row = k * lin_cs4
    // LayerNorm folded into its neighbours (one stream, ContentVec's post-LN layers):
    //  * consumer of a not-yet-normalised tensor y (igemm2 LNB instantiations): the weights carry the LayerNorm scale, the bias its
    //    shift; the kernel sums y and y^2 per column from the operand stream it reads anyway and finishes
    //    out = rstd[n] * (acc - mean[n] * ln_wsum[m]) + bias[m];  the tm == 0 workgroups also publish (mean, rstd) per column
    //  * a later layer whose RESIDUAL is LayerNorm(y): res = (y - mean[n]) * rstd[n] * ln_g[row] + ln_bt[row], from the published stats
    const float *ln_wsum;    // [M] sum_k W'[m][k]
    float *ln_stats_out;     // [N][2]
    const float *ln_stats_in, *ln_g, *ln_bt;
    float ln_eps, ln_inv_rows;
    // conv_tile_kernel (conv_tile.hip.h): work-item table (phase | m-tile << 8 | n-tile << 16, or -1) indexed by blockIdx.x, LDS-offset
    // tables, and the input row geometry (row stride, first / last readable column of a row: the halo)
    const int *items, *ttab;
    int x_ld, x_lo, x_lim, pad2_;
};

__device__ __forceinline__ void epilogue_store(const IgemmP &p, const PhaseD &ph, int b, int m, int n, float acc)
{
    if (m >= p.M || n >= p.N) return;
    if (p.fold_n) { b = n / p.fold_n; n -= b * p.fold_n; }
    int nh = 0, nw = n;
    if (p.y_hm) { nh = n / p.NW; nw = n - nh * p.NW; }
    const int ow = nw * p.y_ws + ph.y_pos, oh = nh * p.y_hm + ph.y_h0;
    if (ow < 0 || ow >= p.OW) return;
    const int ch = m + ph.y_c0;
    float v = acc;
    if (p.bias) v += p.bias[ph.bias_off + m];
    v = apply_act(v, p.act, p.slope);
    if (p.res) v += p.res[(long long)b * p.res_bs + (long long)(p.res_nogroup ? m : ch) * p.res_cs + (long long)oh * p.res_rs + ow];
    v *= p.scale;
    float *yp = p.y + (long long)b * p.y_bs + (long long)ch * p.y_cs + (long long)oh * p.y_rs + ow;
    if (p.accumulate) v += *yp;
    *yp = v;
}

// Fused WaveNet gate: the weight rows of the in-layer are packed so that every 16-row MFMA fragment holds 8 output channels --
// fragment row kq*4 + r is the tanh row of channel f*8 + kq*2 + (r&1) for r < 2 and the sigmoid row of the same channel for
// r >= 2 -- so one lane owns both halves of a channel in its accumulator registers (r, r + 2).
__device__ __forceinline__ void glu_store(const IgemmP &p, const PhaseD &ph, int b, int m1, int n, float a1, float a2)
{
    if (m1 >= p.M || n >= p.N) return;
    if (p.fold_n) { b = n / p.fold_n; n -= b * p.fold_n; }
    const float ta = a1 + p.bias[ph.bias_off + m1], sa = a2 + p.bias[ph.bias_off + m1 + 2];
    const int ch = (m1 >> 4) * 8 + ((m1 & 15) >> 2) * 2 + (m1 & 1) + ph.y_c0;
    p.y[(long long)b * p.y_bs + (long long)ch * p.y_cs + n] = tanhf(ta) * (1.0f / (1.0f + expf(-sa)));
}

// Latency-chain reduction for short kernels (B = 1): the epilogue's operands (bias, residual, previous output for
// accumulate) depend only on the kernel arguments, so they are loaded at kernel start and consumed at the end.
struct EpiPre { float bias, res, yold; };
__device__ __forceinline__ bool epi_locate(const IgemmP &p, const PhaseD &ph, int m, int n, int &ch, int &oh, int &ow)
{
    if (m >= p.M || n >= p.N) return false;
    int nh = 0, nw = n;
    if (p.y_hm) { nh = n / p.NW; nw = n - nh * p.NW; }
    ow = nw * p.y_ws + ph.y_pos; oh = nh * p.y_hm + ph.y_h0;
    if (ow < 0 || ow >= p.OW) return false;
    ch = m + ph.y_c0;
    return true;
}
__device__ __forceinline__ EpiPre epi_prefetch(const IgemmP &p, const PhaseD &ph, int b, int m, int n)
{
    EpiPre e = {0.f, 0.f, 0.f};
    int ch, oh, ow;
    if (!epi_locate(p, ph, m, n, ch, oh, ow)) return e;
    if (p.bias) e.bias = p.bias[ph.bias_off + m];
    if (p.res) e.res = p.res[(long long)b * p.res_bs + (long long)(p.res_nogroup ? m : ch) * p.res_cs + (long long)oh * p.res_rs + ow];
    if (p.accumulate) e.yold = p.y[(long long)b * p.y_bs + (long long)ch * p.y_cs + (long long)oh * p.y_rs + ow];
    return e;
}
__device__ __forceinline__ void epi_finish(const IgemmP &p, const PhaseD &ph, int b, int m, int n, float acc, const EpiPre &e)
{
    int ch, oh, ow;
    if (!epi_locate(p, ph, m, n, ch, oh, ow)) return;
    float v = apply_act(acc + e.bias, p.act, p.slope);
    v += e.res;
    v *= p.scale;
    v += e.yold;
    p.y[(long long)b * p.y_bs + (long long)ch * p.y_cs + (long long)oh * p.y_rs + ow] = v;
}

// Workgroup barrier that only orders LDS traffic: waits for this wave's LDS operations (lgkmcnt(0)) and joins the barrier, leaving
// global loads in flight (gfx9 s_waitcnt encoding: vmcnt = 63, expcnt = 7, lgkmcnt = 0).
__device__ __forceinline__ void lds_only_barrier()
{
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
}

// igemm_kernel<MF, NF, D, KS>
//   One wave owns a (16*MF) x (16*NF) output tile of 16x16x4 fp32 MFMA fragments.
//   KS == 1: the 4 waves of a workgroup work on 4 consecutive tiles (they share weight rows through L1).
//   KS  > 1: the KS waves of a workgroup split the K chunks of ONE tile and sum their partial accumulators
//            through LDS in a fixed order (deterministic) -- the shape for B=1, where a layer has few tiles
//            but a long K (weight streaming): KS times more loads in flight, no second kernel.
//   The workgroup's slice of the koff table is staged in LDS once; weights and gathered activations are
//   register-prefetched D chunks (of 16 k) ahead; the koff entries of the next chunk are read from LDS
//   one stage early so the LDS latency is off the critical path.
#ifdef RVC_KPROBE
#define RVC_KP(i) do { if (p.probe && (threadIdx.x & 63) == 0) p.probe[((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define RVC_KP(i)
#endif
template <int MF, int NF, int D, int KS, bool PRE>
__global__ __launch_bounds__((KS > 1 ? KS : 4) * 64) void igemm_kernel(IgemmP p)
{
    constexpr int WAVES = KS > 1 ? KS : 4;
    RVC_KP(0);
    constexpr int NACC = (MF * NF == 1) ? 2 : 1;   // a lone fragment alternates two accumulators (MFMA dependency)
    constexpr int TE = MF * NF * 256;              // elements of one tile
    constexpr int PE = (KS > 1) ? ((TE + WAVES * 64 - 1) / (WAVES * 64)) : 1;
    constexpr bool PF = (KS > 1) || (MF * NF <= 4);   // prefetch the epilogue operands (register budget permitting)
    extern __shared__ __attribute__((aligned(16))) int s_koff[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = KS > 1 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    int z = blockIdx.y;
    const int ks = z % p.ksplit; z /= p.ksplit;
    const int phase = z % p.nphase;
    const int b = z / p.nphase;
    const PhaseD ph = p.nphase == 1 ? p.ph0 : p.ph[phase];
    const int nchunks = ph.nchunks;
    const int g0 = ks * p.chunks_per_split;
    int g1 = g0 + p.chunks_per_split;
    g1 = g1 < nchunks ? g1 : nchunks;
    const int gn = g1 - g0;                 // chunks of this workgroup (grid-level split)
    {
        const int4 *src = reinterpret_cast<const int4 *>(p.koff + ph.koff_off + g0 * 16);
        int4 *dst = reinterpret_cast<int4 *>(s_koff);
        for (int i = threadIdx.x; i < gn * 4; i += WAVES * 64) dst[i] = src[i];
    }
    RVC_KP(8);
    int tn, tm;
    if (p.m_fast) { tm = tile % p.m_fast; tn = tile / p.m_fast; }
    else { tn = tile % p.ntn; tm = tile / p.ntn; }
    const bool live = tm < p.ntm && tn < p.ntn;   // padding of the XCD-aware order / grid tail (uniform per workgroup when KS > 1)
    const int li = lane & 15, kq = lane >> 4;

    // epilogue operands, loaded up front
    EpiPre pre_w[(KS > 1 || !PF) ? 1 : MF][(KS > 1 || !PF) ? 1 : NF][4];
    EpiPre pre_r[PE];
    if (PF && live && p.ksplit == 1) {
        if (KS > 1) {
#pragma unroll
            for (int q = 0; q < PE; q++) {
                const int e = threadIdx.x + q * WAVES * 64;
                const int l = e & 63, r = (e >> 6) & 3, f = e >> 8, mf = f / NF, nf = f - mf * NF;
                pre_r[q] = (e < TE) ? epi_prefetch(p, ph, b, tm * 16 * MF + mf * 16 + (l >> 4) * 4 + r, tn * 16 * NF + nf * 16 + (l & 15)) : EpiPre{0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int mf = 0; mf < ((KS > 1 || !PF) ? 1 : MF); mf++)
#pragma unroll
                for (int nf = 0; nf < ((KS > 1 || !PF) ? 1 : NF); nf++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        pre_w[mf][nf][r] = epi_prefetch(p, ph, b, tm * 16 * MF + mf * 16 + kq * 4 + r, tn * 16 * NF + nf * 16 + li);
        }
    }
    RVC_KP(9);
    // (the barrier that publishes the koff slice comes after the weight loads of the first D stages have been issued:
    //  they do not depend on it, so their latency overlaps the table's round trip)
    // this wave's chunk range inside the workgroup's slice
    int c0 = 0, nc = gn;
    if (KS > 1) {
        const int cpw = (gn + KS - 1) / KS;
        c0 = wave * cpw;
        int c1 = c0 + cpw;
        c1 = c1 < gn ? c1 : gn;
        nc = c1 > c0 ? c1 - c0 : 0;
    }

    // gathered-activation addressing: wave-uniform base (SGPR pair) + unsigned 32-bit BYTE offset per lane, so each load
    // costs one v_add_u32 (table entries are byte offsets biased by koff_bias to be non-negative)
    const char *xb = reinterpret_cast<const char *>(p.x + (long long)b * p.x_bs + ph.x_off) - p.koff_bias;
    unsigned xo[NF];
#pragma unroll
    for (int nf = 0; nf < NF; nf++) {
        int n = tn * 16 * NF + nf * 16 + li;
        n = n < p.N ? n : p.N - 1;
        int nh = 0, nw = n;
        if (p.x_hs) { nh = n / p.NW; nw = n - nh * p.NW; }
        xo[nf] = (unsigned)(nh * p.x_hs + nw * p.x_ws) * 4u;
    }
    // weights are pre-packed in MFMA-fragment order [m_tile][chunk][lane][4]: one wave-wide dwordx4 load of a
    // (16 rows x 16 k) fragment is 1 KiB fully contiguous (lane l holds W[mt*16 + (l&15)][c*16 + (l>>4)*4 + 0..3])
    const float *wrow[MF];
    const int mtiles = (p.M + 15) >> 4;
#pragma unroll
    for (int mf = 0; mf < MF; mf++) {
        int mt = tm * MF + mf;
        mt = mt < mtiles ? mt : mtiles - 1;
        wrow[mf] = p.w + ph.w_off + ((long long)mt * nchunks + (g0 + c0)) * 256 + lane * 4;
    }
    const int4 *kol = reinterpret_cast<const int4 *>(s_koff) + c0 * 4 + kq;
    const float pre_slope = p.pre_slope;

    f32x4 acc[NACC][MF][NF];
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int nf = 0; nf < NF; nf++) acc[a][mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

    RVC_KP(10);
    f32x4 a_st[D][MF];
    float b_st[D][NF][4];
#define RVC_LOAD_A(S, C)                                                                               \
    {                                                                                                  \
        _Pragma("unroll") for (int mf = 0; mf < MF; mf++) a_st[S][mf] = *reinterpret_cast<const f32x4 *>(wrow[mf] + (C) * 256); \
    }
#define RVC_LOAD_B(S, C)                                                                               \
    {                                                                                                  \
        const int cc_ = (C);                                                                           \
        const int4 ko_ = ko_nx;                                                                        \
        ko_nx = kol[(cc_ + 1 < nc ? cc_ + 1 : cc_) * 4];                                               \
        _Pragma("unroll") for (int nf = 0; nf < NF; nf++) {                                            \
            b_st[S][nf][0] = *reinterpret_cast<const float *>(xb + (xo[nf] + (unsigned)ko_.x));        \
            b_st[S][nf][1] = *reinterpret_cast<const float *>(xb + (xo[nf] + (unsigned)ko_.y));        \
            b_st[S][nf][2] = *reinterpret_cast<const float *>(xb + (xo[nf] + (unsigned)ko_.z));        \
            b_st[S][nf][3] = *reinterpret_cast<const float *>(xb + (xo[nf] + (unsigned)ko_.w));        \
        }                                                                                              \
    }
#define RVC_COMPUTE_STAGE(S)                                                                          \
    {                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                                  \
            _Pragma("unroll") for (int nf = 0; nf < NF; nf++) {                                        \
                /* fused input LeakyReLU, branch-free so the loads stay in flight (PRE layers only) */ \
                const float bv_ = PRE ? fmaxf(b_st[S][nf][j], b_st[S][nf][j] * pre_slope) : b_st[S][nf][j]; \
                _Pragma("unroll") for (int mf = 0; mf < MF; mf++)                                      \
                    acc[j % NACC][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_st[S][mf][j], bv_, acc[j % NACC][mf][nf], 0, 0, 0); \
            }                                                                                          \
    }
#define RVC_LOAD_STAGE(S, C) { RVC_LOAD_A(S, C) RVC_LOAD_B(S, C) }
    if (live) {
#pragma unroll
        for (int s = 0; s < D; s++)
            if (s < nc) RVC_LOAD_A(s, s)
    }
    RVC_KP(11);
    lds_only_barrier();        // not __syncthreads(): its vmcnt(0) would drain the weight loads just issued
    RVC_KP(1);
    if (!live) return;
    int4 ko_nx = nc > 0 ? kol[0] : make_int4(0, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < D; s++)
        if (s < nc) RVC_LOAD_B(s, s)
    RVC_KP(2);
    int c = 0;
    for (; c + 2 * D <= nc; c += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            RVC_COMPUTE_STAGE(s)
            __builtin_amdgcn_sched_barrier(0);
            RVC_LOAD_STAGE(s, c + s + D)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (; c < nc; c += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            if (c + s < nc) {
                RVC_COMPUTE_STAGE(s)
                if (c + s + D < nc) RVC_LOAD_STAGE(s, c + s + D)
            }
        }
    }
#undef RVC_COMPUTE_STAGE
#undef RVC_LOAD_STAGE
#undef RVC_LOAD_A
#undef RVC_LOAD_B
    if (NACC == 2) {
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int nf = 0; nf < NF; nf++) acc[0][mf][nf] += acc[NACC - 1][mf][nf];
    }
    RVC_KP(3);

    if (KS > 1) {
        // fixed-order reduction of the KS partial tiles through LDS, then every thread finishes its share of the tile
        float *red = reinterpret_cast<float *>(s_koff + gn * 16);     // [KS][TE]
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int nf = 0; nf < NF; nf++)
#pragma unroll
                for (int r = 0; r < 4; r++) red[wave * TE + ((mf * NF + nf) * 4 + r) * 64 + lane] = acc[0][mf][nf][r];
        RVC_KP(4);
        __syncthreads();
        RVC_KP(5);
#pragma unroll
        for (int q = 0; q < PE; q++) {
            const int e = threadIdx.x + q * WAVES * 64;
            if (e >= TE) break;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < KS; w++) v += red[w * TE + e];
            const int l = e & 63, r = (e >> 6) & 3, f = e >> 8, mf = f / NF, nf = f - mf * NF;
            const int m = tm * 16 * MF + mf * 16 + (l >> 4) * 4 + r, n = tn * 16 * NF + nf * 16 + (l & 15);
            if (p.glu) {
                if (r < 2) {
                    float v2 = 0.f;
#pragma unroll
                    for (int w = 0; w < KS; w++) v2 += red[w * TE + e + 128];
                    glu_store(p, ph, b, m, n, v, v2);
                }
                continue;
            }
            if (p.ksplit == 1) epi_finish(p, ph, b, m, n, v, pre_r[q]);
            else if (m < p.M && n < p.N)
                p.part[(((long long)(b * p.nphase + phase) * p.ksplit + ks) * p.M + m) * (long long)p.N + n] = v;
        }
        RVC_KP(6);
        return;
    }
    // D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
    if (p.glu) {
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int nf = 0; nf < NF; nf++)
#pragma unroll
                for (int r = 0; r < 2; r++)
                    glu_store(p, ph, b, tm * 16 * MF + mf * 16 + kq * 4 + r, tn * 16 * NF + nf * 16 + li, acc[0][mf][nf][r], acc[0][mf][nf][r + 2]);
        return;
    }
    if (p.ksplit == 1) {
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int nf = 0; nf < NF; nf++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int m = tm * 16 * MF + mf * 16 + kq * 4 + r, n = tn * 16 * NF + nf * 16 + li;
                    if (PF) epi_finish(p, ph, b, m, n, acc[0][mf][nf][r], pre_w[PF ? mf : 0][PF ? nf : 0][r]);
                    else epilogue_store(p, ph, b, m, n, acc[0][mf][nf][r]);
                }
        RVC_KP(6);
    } else {
        // partial sums: part[((b*nphase + phase)*ksplit + ks)][M][N]
        float *pp = p.part + ((long long)(b * p.nphase + phase) * p.ksplit + ks) * (long long)p.M * p.N;
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int nf = 0; nf < NF; nf++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int m = tm * 16 * MF + mf * 16 + kq * 4 + r, n = tn * 16 * NF + nf * 16 + li;
                    if (m < p.M && n < p.N) pp[(long long)m * p.N + n] = acc[0][mf][nf][r];
                }
    }
}

// Workgroups reach the 8 XCDs round-robin in dispatch order (block b of a launch on XCD (b + c) % 8, tests/tools/xcd_probe.hip), and every XCD
// has its own L2.  With m-fastest tile order on the raw block index the m-tiles of one activation tile sat on ntm DIFFERENT XCDs, and every
// one of them fetched the tile again (round 4 PMC pass per kernel: the 768 x 3072 projection at 64 streams read 471 MB per launch for 118 MB
// of operands, the first stride-2 stem convolution 4.4 GB for 0.94).  xcd_tile_id gives the workgroups that one XCD receives CONSECUTIVE tile
// ids: an XCD then owns whole activation tiles (each fetched once), and only the weights -- the small operand when streams are folded into N --
// are fetched once per XCD.  Scalar arithmetic, once per workgroup.  (The tiled kernels take m_fast = 1 for this order and m_fast = 2 for the raw
// block index: when the m-tile count is a multiple of 8 AND the weights exceed an L2 -- ContentVec's 3072 x 768 projection, 24 m-tiles, 9.4 MB --
// the raw order keeps every weight-row block on ONE XCD and streams the small activation tensor through all eight: 218 MB per launch against 319.)
__device__ __forceinline__ int xcd_tile_id(int x, int gx, int row_start)
{
    const int off = row_start & 7, c = (x + off) & 7;
    int base = 0;
#pragma unroll
    for (int cc = 0; cc < 8; cc++) {
        const int f = (cc - off) & 7;
        const int cnt = f < gx ? (gx - f + 7) >> 3 : 0;
        base += cc < c ? cnt : 0;
    }
    return base + ((x - ((c - off) & 7)) >> 3);
}

// ------------------------------------------------------------------------------------------------------------------------
// igemm2_kernel -- the same tile computation as igemm_kernel with a lean launch prologue / epilogue for the one-stream
// latency chain (measured with tests/tools/kprobe.py: of the 16 us a 0.5 GFLOP ContentVec GEMM took, 2.0 us went from wave
// entry to the first barrier -- scalar divisions for the tile index, a dependent global -> LDS copy of the offset table,
// 64-bit address arithmetic -- and 1.3 us into a branchy per-element epilogue):
//   * tile coordinates come from a 2-D grid (x = fast axis, y = slow axis, z = batch * nphase + phase): no divisions.  With
//     m_fast the fast axis is m and gridDim.x is a multiple of 8, so workgroup (x, y) still runs on XCD x % 8;
//   * the offset table is requested FIRST and written to LDS only after the index arithmetic, the epilogue operand requests
//     and the first D weight loads have been issued; with KS > 1 every wave stages just its own K slice (no workgroup barrier);
//   * LIN layers (1x1 convolution on a 1-D tensor = every Linear of the transformers) need no table at all: the k-th operand
//     row is k * channel stride, so the activation gathers leave together with the weight loads;
//   * epilogue addresses are 32-bit element offsets from per-batch bases, the activation is dispatched once per tile.
// Grid-level split-K (a table too long for LDS) stays on igemm_kernel.
template <int ACT> __device__ __forceinline__ float act_t(float v, float slope)
{
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_LRELU) return v > 0.f ? v : v * slope;
    if (ACT == ACT_GELU) return gelu_dev(v);
    if (ACT == ACT_TANH) return tanhf(v);
    if (ACT == ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}
struct Epi2 { float bias, res, yold; int yo; };
__device__ __forceinline__ Epi2 epi2_prefetch(const IgemmP &p, const PhaseD &ph, const float *resb, const float *yb, int m, int n)
{
    Epi2 e = {0.f, 0.f, 0.f, -1};
    if (m >= p.M || n >= p.N) return e;
    const int n_launch = n;
    int bb = 0;
    if (p.fold_n) { bb = n / p.fold_n; n -= bb * p.fold_n; }
    int nh = 0, nw = n;
    if (p.y_hm) { nh = n / p.NW; nw = n - nh * p.NW; }
    const int ow = nw * p.y_ws + ph.y_pos, oh = nh * p.y_hm + ph.y_h0;
    if (ow < 0 || ow >= p.OW) return e;
    const int ch = m + ph.y_c0;
    e.yo = bb * (int)p.y_bs + ch * p.y_cs + oh * p.y_rs + ow;
    if (p.bias) e.bias = p.bias[ph.bias_off + m];
    if (resb) {
        const int rrow = p.res_nogroup ? m : ch;
        e.res = resb[bb * (int)p.res_bs + rrow * p.res_cs + oh * p.res_rs + ow];
        if (p.ln_stats_in)       // the residual is LayerNorm(stored tensor): normalise on the fly
            e.res = (e.res - p.ln_stats_in[2 * n_launch]) * p.ln_stats_in[2 * n_launch + 1] * p.ln_g[rrow] + p.ln_bt[rrow];
    }
    if (p.accumulate) e.yold = yb[e.yo];
    return e;
}
// the epilogue's arithmetic, in ONE place (every path must round the same way: eager/graph and tile choice are bit-identical)
template <int ACT> __device__ __forceinline__ float epi2_value(float acc, float bias, float res, float yold, float slope, float scale)
{
    float v = act_t<ACT>(acc + bias, slope);
    v += res;
    v *= scale;
    v += yold;
    return v;
}
template <int ACT> __device__ __forceinline__ void epi2_finish(const IgemmP &p, float *yb, float acc, const Epi2 &e)
{
    if (e.yo < 0) return;
    yb[e.yo] = epi2_value<ACT>(acc, e.bias, e.res, e.yold, p.slope, p.scale);
}
// Column part of the output address (everything that depends on n only: stream, row, column, validity), computed once per MFMA
// column instead of once per element -- with folded streams it holds an integer division.
struct ColOut { int yo, ro; };          // yo < 0: column not stored
__device__ __forceinline__ ColOut col_locate(const IgemmP &p, const PhaseD &ph, int n)
{
    ColOut c = {-1, 0};
    if (n >= p.N) return c;
    int bb = 0;
    if (p.fold_n) { bb = n / p.fold_n; n -= bb * p.fold_n; }
    int nh = 0, nw = n;
    if (p.y_hm) { nh = n / p.NW; nw = n - nh * p.NW; }
    const int ow = nw * p.y_ws + ph.y_pos, oh = nh * p.y_hm + ph.y_h0;
    if (ow < 0 || ow >= p.OW) return c;
    c.yo = bb * (int)p.y_bs + oh * p.y_rs + ow;
    c.ro = bb * (int)p.res_bs + oh * p.res_rs + ow;
    return c;
}
__device__ __forceinline__ Epi2 epi2_from_col(const IgemmP &p, const PhaseD &ph, const float *resb, const float *yb, const ColOut &c, int m)
{
    Epi2 e = {0.f, 0.f, 0.f, -1};
    if (m >= p.M || c.yo < 0) return e;
    const int ch = m + ph.y_c0;
    e.yo = c.yo + ch * p.y_cs;
    if (p.bias) e.bias = p.bias[ph.bias_off + m];
    if (resb) e.res = resb[c.ro + (p.res_nogroup ? m : ch) * p.res_cs];
    if (p.accumulate) e.yold = yb[e.yo];
    return e;
}
__device__ __forceinline__ Epi2 epi2_plain(float bias, int yo) { Epi2 e = {bias, 0.f, 0.f, yo}; return e; }
__device__ __forceinline__ Epi2 epi2_none() { Epi2 e = {0.f, 0.f, 0.f, -1}; return e; }
// the same with the bias already in hand (batched epilogues: biases are loaded once per row, before any store)
__device__ __forceinline__ Epi2 epi2_aux(const IgemmP &p, const PhaseD &ph, const float *resb, const float *yb, const ColOut &c, int m, float bias)
{
    Epi2 e = {bias, 0.f, 0.f, -1};
    if (m >= p.M || c.yo < 0) return e;
    const int ch = m + ph.y_c0;
    e.yo = c.yo + ch * p.y_cs;
    if (resb) e.res = resb[c.ro + (p.res_nogroup ? m : ch) * p.res_cs];
    if (p.accumulate) e.yold = yb[e.yo];
    return e;
}
// fused WaveNet gate on a located column (see glu_store)
__device__ __forceinline__ void glu_from_col(const IgemmP &p, const PhaseD &ph, float *yb, const ColOut &c, int m1, float a1, float a2)
{
    if (m1 >= p.M || c.yo < 0) return;
    const float ta = a1 + p.bias[ph.bias_off + m1], sa = a2 + p.bias[ph.bias_off + m1 + 2];
    const int ch = (m1 >> 4) * 8 + ((m1 & 15) >> 2) * 2 + (m1 & 1) + ph.y_c0;
    yb[c.yo + ch * p.y_cs] = tanhf(ta) * (1.0f / (1.0f + expf(-sa)));
}

__device__ __forceinline__ void glu_from_col_b(const IgemmP &p, const PhaseD &ph, float *yb, const ColOut &c, int m1, float a1, float a2, float b1, float b2)
{
    if (m1 >= p.M || c.yo < 0) return;
    const float ta = a1 + b1, sa = a2 + b2;
    const int ch = (m1 >> 4) * 8 + ((m1 & 15) >> 2) * 2 + (m1 & 1) + ph.y_c0;
    yb[c.yo + ch * p.y_cs] = tanhf(ta) * (1.0f / (1.0f + expf(-sa)));
}

#define RVC_ACT_DISPATCH(STMT) RVC_ACT_DISPATCH_SEL(p.act, STMT)
#define RVC_ACT_DISPATCH_SEL(SEL, STMT)                                          \
    switch (SEL) {                                                             \
    case ACT_RELU: { constexpr int A_ = ACT_RELU; STMT } break;                  \
    case ACT_LRELU: { constexpr int A_ = ACT_LRELU; STMT } break;                \
    case ACT_GELU: { constexpr int A_ = ACT_GELU; STMT } break;                  \
    case ACT_TANH: { constexpr int A_ = ACT_TANH; STMT } break;                  \
    case ACT_SIGMOID: { constexpr int A_ = ACT_SIGMOID; STMT } break;            \
    default: { constexpr int A_ = ACT_NONE; STMT } break;                        \
    }

// Minimum waves per SIMD the register allocation has to leave room for.  LNB: 264 registers otherwise -- one wave per SIMD.  The two others are the
// instantiations that round 5's two-path K loop pushed over an occupancy step (132 registers for the lone fragment with four K shares: three waves
// per SIMD instead of four, and RMVPE's 128-workgroup launches on the 32-CU partition took a second round: 81 -> 116 us per chunk; 172 for the 2 x 4
// tile with the fused input activation).
template <int MF, int NF, int KS, bool PRE, bool LIN, bool LNB> struct Ig2Occ {
    static constexpr int W = LNB ? 2 : ((MF * NF == 1 && KS == 4 && !PRE && !LIN) ? 4 : ((MF * NF == 8 && KS == 1 && PRE) ? 3 : 1));
};
template <int MF, int NF, int D, int KS, bool PRE, bool LIN, bool LNB = false>
__global__ __launch_bounds__((KS > 1 ? KS : 4) * 64) __attribute__((amdgpu_waves_per_eu(Ig2Occ<MF, NF, KS, PRE, LIN, LNB>::W)))
void igemm2_kernel(IgemmP p)
{
    static_assert(!LNB || (LIN && !PRE && KS > 1), "LayerNorm-consumer instantiations: table-free 1x1 layers with the in-workgroup K split");
    constexpr int WAVES = KS > 1 ? KS : 4;
    constexpr int NACC = (MF * NF == 1) ? 2 : 1;   // a lone fragment alternates two accumulators (MFMA dependency)
    constexpr int TE = MF * NF * 256;
    constexpr int PE = (KS > 1) ? ((TE + WAVES * 64 - 1) / (WAVES * 64)) : 1;
    constexpr bool PF = (KS > 1) || (MF * NF <= 4);
    constexpr int KR = 2;                         // offset-table entries (int4) a thread can hold between request and LDS write
    extern __shared__ __attribute__((aligned(16))) int s_koff[];
    RVC_KP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int fast = KS > 1 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave, slow = (int)blockIdx.y;
    const int tm = p.m_fast ? fast : slow, tn = p.m_fast ? slow : fast;
    int phase = 0, b = 0;
    {
        const int z = (int)blockIdx.z;
        if (p.nphase == 1) b = z;
        else if (p.nbatch == 1) phase = z;
        else { b = z / p.nphase; phase = z - b * p.nphase; }
    }
    PhaseD ph = p.ph0;
    if (phase) ph = p.ph[phase];
    const int nchunks = ph.nchunks;
    // this wave's chunk range (KS > 1: the waves of the workgroup split K)
    int c0 = 0, nc = nchunks;
    if (KS > 1) {
        const int cpw = (nchunks + KS - 1) / KS;
        c0 = wave * cpw;
        int c1 = c0 + cpw;
        c1 = c1 < nchunks ? c1 : nchunks;
        nc = c1 > c0 ? c1 - c0 : 0;
    }
    // 1. request the offset-table slice (KS > 1: this wave's; KS == 1: the workgroup's) -- consumed after everything else is in flight
    int4 kr[KR];
    const int kt_n = LIN ? 0 : (KS > 1 ? nc * 4 : nchunks * 4);          // int4 entries to stage
    const int kt_i = KS > 1 ? lane : (int)threadIdx.x;
    constexpr int KT_STRIDE = KS > 1 ? 64 : WAVES * 64;
    const int4 *ksrc = reinterpret_cast<const int4 *>(p.koff + ph.koff_off) + (KS > 1 ? c0 * 4 : 0);
    int4 *kdst = reinterpret_cast<int4 *>(s_koff) + (KS > 1 ? c0 * 4 : 0);
    if (!LIN) {
#pragma unroll
        for (int r = 0; r < KR; r++) { const int i = kt_i + r * KT_STRIDE; if (i < kt_n) kr[r] = ksrc[i]; }
    }
    RVC_KP(8);
    const bool live = tm < p.ntm && tn < p.ntn;
    const int li = lane & 15, kq = lane >> 4;
    const float *resb = p.res ? p.res + (long long)b * p.res_bs : nullptr;
    float *yb = p.y + (long long)b * p.y_bs + ph.y_off;
    const int act_sel = ph.act_p1 ? ph.act_p1 - 1 : p.act;

    RVC_KP(9);
    // gathered-activation addressing: wave-uniform base + unsigned 32-bit BYTE offset per lane
    const char *xb = reinterpret_cast<const char *>(p.x + (long long)b * p.x_bs + ph.x_off) - (LIN ? 0 : p.koff_bias);
    unsigned xo[NF];
#pragma unroll
    for (int nf = 0; nf < NF; nf++) {
        int n = tn * 16 * NF + nf * 16 + li;
        n = n < p.N ? n : p.N - 1;
        int bb = 0;
        if (p.fold_n) { bb = n / p.fold_n; n -= bb * p.fold_n; }
        int nh = 0, nw = n;
        if (p.x_hs) { nh = n / p.NW; nw = n - nh * p.NW; }
        xo[nf] = (unsigned)(bb * (int)p.x_bs + nh * p.x_hs + nw * p.x_ws) * 4u;
        if (LIN) xo[nf] += (unsigned)((c0 * 16 + kq * 4) * p.lin_cs4);
    }
    // weights: MFMA-fragment order [m_tile][chunk][lane][4]
    const float *wrow[MF];
    const int mtiles = (p.M + 15) >> 4;
#pragma unroll
    for (int mf = 0; mf < MF; mf++) {
        int mt = tm * MF + mf;
        mt = mt < mtiles ? mt : mtiles - 1;
        wrow[mf] = p.w + ph.w_off + ((long long)mt * nchunks + c0) * 256 + lane * 4;
    }
    const int4 *kol = reinterpret_cast<const int4 *>(s_koff) + c0 * 4 + kq;
    const float pre_slope = p.pre_slope;
    const unsigned lin1 = (unsigned)p.lin_cs4, lin16 = 16u * (unsigned)p.lin_cs4;

    f32x4 acc[NACC][MF][NF];
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int nf = 0; nf < NF; nf++) acc[a][mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    RVC_KP(10);

    f32x4 a_st[D][MF];
    float b_st[D][NF][4];
    // LNB: column sums of the operand values this lane feeds to the matrix core, taken relative to the column's FIRST element (every
    // lane and wave of a column loads the same one): the one-pass variance E[d^2] - E[d]^2 then cancels against a shift of the order
    // of the spread, not of the mean (a column with |mean| >> std would otherwise lose its variance to rounding)
    float ln_s[LNB ? NF : 1], ln_ss[LNB ? NF : 1], ln_c[LNB ? NF : 1];
#pragma unroll
    for (int nf = 0; nf < (LNB ? NF : 1); nf++) {
        ln_s[nf] = 0.f; ln_ss[nf] = 0.f;
        ln_c[nf] = LNB ? *reinterpret_cast<const float *>(xb + (xo[LNB ? nf : 0] - (unsigned)((c0 * 16 + kq * 4) * p.lin_cs4))) : 0.f;
    }
    int4 ko_nx = make_int4(0, 0, 0, 0);
#define RVC_LOAD_A(S, C)                                                                               \
    {                                                                                                  \
        _Pragma("unroll") for (int mf = 0; mf < MF; mf++) a_st[S][mf] = *reinterpret_cast<const f32x4 *>(wrow[mf] + (C) * 256); \
    }
#define RVC_LOAD_B(S, C)                                                                               \
    {                                                                                                  \
        const int cc_ = (C);                                                                           \
        int4 ko_;                                                                                      \
        if (LIN) { const unsigned kb_ = (unsigned)cc_ * lin16; ko_ = make_int4((int)kb_, (int)(kb_ + lin1), (int)(kb_ + 2u * lin1), (int)(kb_ + 3u * lin1)); } \
        else { ko_ = ko_nx; ko_nx = kol[(cc_ + 1 < nc ? cc_ + 1 : cc_) * 4]; }                         \
        _Pragma("unroll") for (int nf = 0; nf < NF; nf++) {                                            \
            b_st[S][nf][0] = *reinterpret_cast<const float *>(xb + (xo[nf] + (unsigned)ko_.x));        \
            b_st[S][nf][1] = *reinterpret_cast<const float *>(xb + (xo[nf] + (unsigned)ko_.y));        \
            b_st[S][nf][2] = *reinterpret_cast<const float *>(xb + (xo[nf] + (unsigned)ko_.z));        \
            b_st[S][nf][3] = *reinterpret_cast<const float *>(xb + (xo[nf] + (unsigned)ko_.w));        \
        }                                                                                              \
    }
#define RVC_COMPUTE_STAGE(S)                                                                          \
    {                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                                  \
            _Pragma("unroll") for (int nf = 0; nf < NF; nf++) {                                        \
                const float bv_ = PRE ? fmaxf(b_st[S][nf][j], b_st[S][nf][j] * pre_slope) : b_st[S][nf][j]; \
                if (LNB) { const float d_ = bv_ - ln_c[LNB ? nf : 0]; ln_s[LNB ? nf : 0] += d_; ln_ss[LNB ? nf : 0] = fmaf(d_, d_, ln_ss[LNB ? nf : 0]); } \
                _Pragma("unroll") for (int mf = 0; mf < MF; mf++)                                      \
                    acc[j % NACC][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_st[S][mf][j], bv_, acc[j % NACC][mf][nf], 0, 0, 0); \
            }                                                                                          \
    }
#define RVC_LOAD_STAGE(S, C) { RVC_LOAD_A(S, C) RVC_LOAD_B(S, C) }
    // The K loop has TWO code paths (round 5).  The compiler's s_waitcnt counts at a loop head are the most conservative of all ways into the loop: with
    // the first D stages requested under `if (s < nc)` there is a path on which only stage 0 was requested, so the head of the steady-state loop (and
    // every stage of the tail loop) waited with vmcnt(0) -- for the loads issued a few MFMAs earlier, once per round of D chunks: the D-deep register
    // ring was drained every round and a one-stream wave (12-18 chunks) spent its life in two or three full memory round trips.  Now a wave with at
    // least D chunks takes a path on which every request of the prologue is unconditional and in stage order, the steady-state loop has that prologue
    // as its only way in (exact counts: stage s waits until (D - 1) stages' loads are outstanding), and the remainder is two straight-line drain
    // rounds; a wave with fewer chunks takes the short conditional path.
    // 3b (both paths). epilogue operands: requested BEHIND the first weight / activation loads (their address arithmetic alone is 0.8 us at a
    //     4-element share per thread; in front of the main loads it delayed every launch by that much)
    Epi2 pre_w[(KS > 1 || !PF) ? 1 : MF][(KS > 1 || !PF) ? 1 : NF][4];
    Epi2 pre_r[PE];
    float pre_ws[LNB ? PE : 1];
#define RVC_EPI_PREFETCH                                                                               \
    if (PF && live && !p.glu) {                                                                        \
        if (KS > 1) {                                                                                  \
            _Pragma("unroll") for (int q = 0; q < PE; q++) {                                           \
                const int e = threadIdx.x + q * WAVES * 64;                                            \
                const int l = e & 63; const int r = (e >> 6) & 3; const int f = e >> 8; const int mf = f / NF; const int nf = f - mf * NF; \
                pre_r[q] = (e < TE) ? epi2_prefetch(p, ph, resb, yb, tm * 16 * MF + mf * 16 + (l >> 4) * 4 + r, tn * 16 * NF + nf * 16 + (l & 15)) : epi2_none(); \
                if (LNB) { const int m_ = tm * 16 * MF + mf * 16 + (l >> 4) * 4 + r; pre_ws[LNB ? q : 0] = (e < TE && m_ < p.M) ? p.ln_wsum[ph.bias_off + m_] : 0.f; } \
            }                                                                                          \
        } else {                                                                                       \
            _Pragma("unroll") for (int nf = 0; nf < ((KS > 1 || !PF) ? 1 : NF); nf++) {                \
                const ColOut col = col_locate(p, ph, tn * 16 * NF + nf * 16 + li);                     \
                _Pragma("unroll") for (int mf = 0; mf < ((KS > 1 || !PF) ? 1 : MF); mf++)              \
                    _Pragma("unroll") for (int r = 0; r < 4; r++)                                      \
                        pre_w[mf][nf][r] = epi2_from_col(p, ph, resb, yb, col, tm * 16 * MF + mf * 16 + kq * 4 + r); \
            }                                                                                          \
        }                                                                                              \
    }
    // 4 (both paths). publish the offset table: registers -> LDS (the rare long tables finish with a plain copy loop)
#define RVC_PUBLISH_TABLE                                                                              \
    if (!LIN) {                                                                                        \
        _Pragma("unroll") for (int r = 0; r < KR; r++) { const int i = kt_i + r * KT_STRIDE; if (i < kt_n) kdst[i] = kr[r]; } \
        for (int i = kt_i + KR * KT_STRIDE; i < kt_n; i += KT_STRIDE) kdst[i] = ksrc[i];               \
        if (KS > 1) __builtin_amdgcn_s_waitcnt(0xC07F);      /* wave-private slice: LDS writes done (lgkmcnt(0)), no barrier */ \
        else lds_only_barrier();                                                                       \
    }
    // steady state of the fused form: every operand register is reloaded right after its last use, so the loads of the next round are interleaved
    // with the MFMAs of this one instead of forming a block during which the matrix pipe drains (measured at 64 streams: the 768 x 3072 projection
    // 75 -> 92 TF/s, the 768 x 768 one 61 -> 78 TF/s; no change at one stream)
#define RVC_FUSED_STAGE(S, C)                                                                          \
    {                                                                                                  \
        const int cc_ = (C);                                                                           \
        int4 ko_;                                                                                      \
        if (LIN) { const unsigned kb_ = (unsigned)cc_ * lin16; ko_ = make_int4((int)kb_, (int)(kb_ + lin1), (int)(kb_ + 2u * lin1), (int)(kb_ + 3u * lin1)); } \
        else { ko_ = ko_nx; ko_nx = kol[(cc_ + 1 < nc ? cc_ + 1 : cc_) * 4]; }                         \
        const unsigned kov_[4] = {(unsigned)ko_.x, (unsigned)ko_.y, (unsigned)ko_.z, (unsigned)ko_.w}; \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                                  \
            _Pragma("unroll") for (int nf = 0; nf < NF; nf++) {                                        \
                const float bv_ = PRE ? fmaxf(b_st[S][nf][j], b_st[S][nf][j] * pre_slope) : b_st[S][nf][j]; \
                if (LNB) { const float d_ = bv_ - ln_c[LNB ? nf : 0]; ln_s[LNB ? nf : 0] += d_; ln_ss[LNB ? nf : 0] = fmaf(d_, d_, ln_ss[LNB ? nf : 0]); } \
                _Pragma("unroll") for (int mf = 0; mf < MF; mf++)                                      \
                    acc[j % NACC][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_st[S][mf][j], bv_, acc[j % NACC][mf][nf], 0, 0, 0); \
                b_st[S][nf][j] = *reinterpret_cast<const float *>(xb + (xo[nf] + kov_[j]));            \
            }                                                                                          \
        /* the weights are reloaded BEHIND their last use (round 5).  Requested at the top of the stage into the same array, the new value lived in a \
           second register set next to a copy of the old one, and the loop-carried copies back (v_mov on registers a load had just been issued \
           into) were placed at the loop head: every round began by waiting for ALL outstanding loads */ \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        _Pragma("unroll") for (int mf = 0; mf < MF; mf++) a_st[S][mf] = *reinterpret_cast<const f32x4 *>(wrow[mf] + cc_ * 256); \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }
    // Not for the lone-fragment tile: with its two alternating accumulators (NACC = 2) the fused form computes garbage as soon as the loop is
    // entered (K >= 24 chunks; `test_every_tile_configuration_computes_the_same_convolution` with -DRVC_FUSE_ALL), with or without
    // scheduling barriers between the stages, while the same source with one accumulator per fragment is exact -- a code-generation
    // problem of that instantiation as far as could be determined.  Its 12-deep prefetch hides the load block anyway.
#ifdef RVC_FUSE_ALL
    constexpr bool kFuse = true;      // investigation build only
#else
    constexpr bool kFuse = NACC == 1;
#endif
    if (live && nc >= D) {
        // ---- the long path: at least D chunks for this wave (wave-uniform; uniform per workgroup when KS == 1, where all waves share nchunks)
        // 2. first D stages, unconditional and in stage order: weights (and, without a table, the activations) leave first
#pragma unroll
        for (int s = 0; s < D; s++) { RVC_LOAD_A(s, s) if (LIN) RVC_LOAD_B(s, s) }
        RVC_EPI_PREFETCH
        RVC_KP(11);
        RVC_PUBLISH_TABLE
        RVC_KP(1);
        if (!LIN) {
            ko_nx = kol[0];
#pragma unroll
            for (int s = 0; s < D; s++) RVC_LOAD_B(s, s)
        }
        RVC_KP(2);
        int c = 0;
        if (kFuse) {
            for (; c + 2 * D <= nc; c += D) {
#pragma unroll
                for (int s = 0; s < D; s++) RVC_FUSED_STAGE(s, c + s + D)
            }
        } else {
            for (; c + 2 * D <= nc; c += D) {
#pragma unroll
                for (int s = 0; s < D; s++) {
                    RVC_COMPUTE_STAGE(s)
                    __builtin_amdgcn_sched_barrier(0);
                    RVC_LOAD_STAGE(s, c + s + D)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // drain: D <= nc - c < 2 D.  Round one computes the D loaded stages and requests the last nc - c - D chunks; round two computes those.
        const int rem = nc - c - D;
#pragma unroll
        for (int s = 0; s < D; s++) {
            RVC_COMPUTE_STAGE(s)
            if (s < rem) RVC_LOAD_STAGE(s, c + s + D)
        }
#pragma unroll
        for (int s = 0; s < D; s++) {
            if (s < rem) RVC_COMPUTE_STAGE(s)
        }
    } else {
        // ---- the short path: fewer than D chunks (or a dead tile: nothing but the barrier the workgroup shares)
        if (live) {
#pragma unroll
            for (int s = 0; s < D; s++)
                if (s < nc) { RVC_LOAD_A(s, s) if (LIN) RVC_LOAD_B(s, s) }
        }
        RVC_EPI_PREFETCH
        RVC_KP(11);
        RVC_PUBLISH_TABLE
        RVC_KP(1);
        if (!live) { if (KS > 1) { /* whole workgroup is dead: uniform */ } return; }
        if (!LIN) {
            ko_nx = nc > 0 ? kol[0] : make_int4(0, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < D; s++)
                if (s < nc) RVC_LOAD_B(s, s)
        }
        RVC_KP(2);
#pragma unroll
        for (int s = 0; s < D; s++) {
            if (s < nc) RVC_COMPUTE_STAGE(s)
        }
    }
#undef RVC_FUSED_STAGE
#undef RVC_EPI_PREFETCH
#undef RVC_PUBLISH_TABLE
#undef RVC_COMPUTE_STAGE
#undef RVC_LOAD_STAGE
#undef RVC_LOAD_A
#undef RVC_LOAD_B
    if (NACC == 2) {
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int nf = 0; nf < NF; nf++) acc[0][mf][nf] += acc[NACC - 1][mf][nf];
    }
    RVC_KP(3);

    if (KS > 1) {
        // fixed-order reduction of the KS partial tiles through LDS, then every thread finishes its share of the tile
        float *red = reinterpret_cast<float *>(s_koff + (LIN ? 0 : nchunks * 16));     // [KS][TE]
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int nf = 0; nf < NF; nf++)
#pragma unroll
                for (int r = 0; r < 4; r++) red[wave * TE + ((mf * NF + nf) * 4 + r) * 64 + lane] = acc[0][mf][nf][r];
        float *lst = red + KS * TE;                 // LNB: [KS][NF * 16][2] column sums of this wave's K slice
        if (LNB) {
#pragma unroll
            for (int nf = 0; nf < NF; nf++) {
                float s_ = ln_s[LNB ? nf : 0], q_ = ln_ss[LNB ? nf : 0];
                s_ += __shfl_xor(s_, 16, 64); q_ += __shfl_xor(q_, 16, 64);
                s_ += __shfl_xor(s_, 32, 64); q_ += __shfl_xor(q_, 32, 64);
                if (kq == 0) { lst[(wave * NF * 16 + nf * 16 + li) * 2] = s_; lst[(wave * NF * 16 + nf * 16 + li) * 2 + 1] = q_; }
            }
        }
        RVC_KP(4);
        __syncthreads();
        RVC_KP(5);
        if (p.glu) {
#pragma unroll
            for (int q = 0; q < PE; q++) {
                const int e = threadIdx.x + q * WAVES * 64;
                if (e >= TE) break;
                const int l = e & 63, r = (e >> 6) & 3, f = e >> 8, mf = f / NF, nf = f - mf * NF;
                if (r >= 2) continue;
                float v = 0.f, v2 = 0.f;
#pragma unroll
                for (int w = 0; w < KS; w++) { v += red[w * TE + e]; v2 += red[w * TE + e + 128]; }
                glu_store(p, ph, b, tm * 16 * MF + mf * 16 + (l >> 4) * 4 + r, tn * 16 * NF + nf * 16 + (l & 15), v, v2);
            }
            return;
        }
        // LNB: statistics of this lane's NF columns, summed over the waves' K slices in a fixed order
        float ln_mean[LNB ? NF : 1], ln_rstd[LNB ? NF : 1];
        if (LNB) {
#pragma unroll
            for (int nf = 0; nf < NF; nf++) {
                const int cl = nf * 16 + li;
                float s_ = 0.f, q_ = 0.f;
#pragma unroll
                for (int w = 0; w < KS; w++) { s_ += lst[(w * NF * 16 + cl) * 2]; q_ += lst[(w * NF * 16 + cl) * 2 + 1]; }
                const float msh = s_ * p.ln_inv_rows;                       // mean of (y - first element)
                const float var = fmaxf(q_ * p.ln_inv_rows - msh * msh, 0.f);
                const float mean = ln_c[LNB ? nf : 0] + msh;
                const float rstd = 1.0f / sqrtf(var + p.ln_eps);
                ln_mean[LNB ? nf : 0] = mean; ln_rstd[LNB ? nf : 0] = rstd;
                const int n_ = tn * 16 * NF + cl;
                if (p.ln_stats_out && tm == 0 && threadIdx.x < 16 && n_ < p.N) { p.ln_stats_out[2 * n_] = mean; p.ln_stats_out[2 * n_ + 1] = rstd; }
            }
        }
        float vsum[PE];
#pragma unroll
        for (int q = 0; q < PE; q++) {
            const int e = threadIdx.x + q * WAVES * 64;
            float v = 0.f;
            if (e < TE) {
#pragma unroll
                for (int w = 0; w < KS; w++) v += red[w * TE + e];
            }
            if (LNB && e < TE) {          // the folded LayerNorm: see IgemmP::ln_wsum (an element's column is (nf, lane & 15))
                const int nf = (e >> 8) % NF;
                v = ln_rstd[LNB ? nf : 0] * (v - ln_mean[LNB ? nf : 0] * pre_ws[LNB ? q : 0]);
            }
            vsum[q] = v;
        }
        RVC_ACT_DISPATCH_SEL(act_sel, 
            _Pragma("unroll") for (int q = 0; q < PE; q++) epi2_finish<A_>(p, yb, vsum[q], pre_r[q]);
        )
        RVC_KP(6);
        return;
    }
    // D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
    if (p.glu) {
        float gb[MF][4];                 // the gate's biases: rows kq * 4 + {0, 1} (tanh half) and + {2, 3} (sigmoid half)
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = tm * 16 * MF + mf * 16 + kq * 4 + r;
                gb[mf][r] = m < p.M ? p.bias[ph.bias_off + m] : 0.f;
            }
#pragma unroll
        for (int nf = 0; nf < NF; nf++) {
            const ColOut col = col_locate(p, ph, tn * 16 * NF + nf * 16 + li);
#pragma unroll
            for (int mf = 0; mf < MF; mf++)
#pragma unroll
                for (int r = 0; r < 2; r++)
                    glu_from_col_b(p, ph, yb, col, tm * 16 * MF + mf * 16 + kq * 4 + r, acc[0][mf][nf][r], acc[0][mf][nf][r + 2], gb[mf][r], gb[mf][r + 2]);
        }
        return;
    }
    if (PF) {
        RVC_ACT_DISPATCH_SEL(act_sel, 
            _Pragma("unroll") for (int mf = 0; mf < MF; mf++)
                _Pragma("unroll") for (int nf = 0; nf < NF; nf++)
                    _Pragma("unroll") for (int r = 0; r < 4; r++)
                        epi2_finish<A_>(p, yb, acc[0][mf][nf][r], pre_w[PF ? mf : 0][PF ? nf : 0][r]);
        )
    } else {
        // operands in store-free batches (a load cannot be hoisted above an earlier, possibly aliasing store: element-by-element
        // "load, store" costs one memory round trip per element -- see igemm32_kernel)
        ColOut cols[NF];
#pragma unroll
        for (int nf = 0; nf < NF; nf++) cols[nf] = col_locate(p, ph, tn * 16 * NF + nf * 16 + li);
        const bool has_aux = resb != nullptr || p.accumulate;
        if (!p.accumulate && (tm + 1) * 16 * MF <= p.M) {
            // common case (every row of the tile exists, plain store): small straight-line code, one predicate per 16-column block,
            // the residual loaded for a whole 16-row fragment before its stores (see igemm32_kernel's epilogue)
            const float slope = p.slope, scale = p.scale;
            const long long cs = p.y_cs, rcs = p.res_cs;
            RVC_ACT_DISPATCH_SEL(act_sel, 
                _Pragma("unroll") for (int mf = 0; mf < MF; mf++) {
                    const int m0 = tm * 16 * MF + mf * 16 + kq * 4;
                    float bias_r[4];
                    _Pragma("unroll") for (int r = 0; r < 4; r++) bias_r[r] = p.bias ? p.bias[ph.bias_off + m0 + r] : 0.f;
                    float rr[NF][4];
                    _Pragma("unroll") for (int nf = 0; nf < NF; nf++)
                        _Pragma("unroll") for (int r = 0; r < 4; r++) rr[nf][r] = 0.f;
                    if (resb) {
                        _Pragma("unroll") for (int nf = 0; nf < NF; nf++)
                            if (cols[nf].yo >= 0) {
                                const float *rp = resb + cols[nf].ro + (long long)(p.res_nogroup ? m0 : m0 + ph.y_c0) * rcs;
                                _Pragma("unroll") for (int r = 0; r < 4; r++) rr[nf][r] = rp[r * rcs];
                            }
                    }
                    _Pragma("unroll") for (int nf = 0; nf < NF; nf++)
                        if (cols[nf].yo >= 0) {
                            float *yc = yb + cols[nf].yo + (long long)(m0 + ph.y_c0) * cs;
                            _Pragma("unroll") for (int r = 0; r < 4; r++)
                                yc[r * cs] = epi2_value<A_>(acc[0][mf][nf][r], bias_r[r], rr[nf][r], 0.f, slope, scale);
                        }
                }
            )
            RVC_KP(6);
            return;
        }
        RVC_ACT_DISPATCH_SEL(act_sel, 
            _Pragma("unroll") for (int mf = 0; mf < MF; mf++) {
                float bias_r[4];
                _Pragma("unroll") for (int r = 0; r < 4; r++) {
                    const int m = tm * 16 * MF + mf * 16 + kq * 4 + r;
                    bias_r[r] = (p.bias && m < p.M) ? p.bias[ph.bias_off + m] : 0.f;
                }
                _Pragma("unroll") for (int nf = 0; nf < NF; nf++) {
                    if (!has_aux) {
                        _Pragma("unroll") for (int r = 0; r < 4; r++) {
                            const int m = tm * 16 * MF + mf * 16 + kq * 4 + r;
                            const Epi2 e1 = epi2_plain(bias_r[r], (m < p.M && cols[nf].yo >= 0) ? cols[nf].yo + (m + ph.y_c0) * p.y_cs : -1);
                            epi2_finish<A_>(p, yb, acc[0][mf][nf][r], e1);
                        }
                    } else {
                        Epi2 e_[4];
                        _Pragma("unroll") for (int r = 0; r < 4; r++)
                            e_[r] = epi2_aux(p, ph, resb, yb, cols[nf], tm * 16 * MF + mf * 16 + kq * 4 + r, bias_r[r]);
                        _Pragma("unroll") for (int r = 0; r < 4; r++) epi2_finish<A_>(p, yb, acc[0][mf][nf][r], e_[r]);
                    }
                }
            }
        )
    }
    RVC_KP(6);
}

typedef float f32x16w __attribute__((ext_vector_type(16)));
// ------------------------------------------------------------------------------------------------------------------------
// igemm2w_kernel -- register-direct implicit GEMM on v_mfma_f32_32x32x2_f32 for the table-free 1x1 layers (every Linear of the
// transformers) at a FEW streams (round 5: the regime between the one-stream latency kernels and the many-stream LDS-staged tiles).
// One wave owns a (32 MT) x (32 NT) output tile; the KS waves of a workgroup split K and meet in a fixed-order LDS reduction
// (deterministic).  No LDS staging: with the streams folded into N a layer has too few tiles for a workgroup tile to share its
// activation operand across enough rows, and an in-workgroup K split of a staged tile multiplies the staging per MFMA; here the
// K split costs nothing but the final reduction.  What the 32x32x2 form buys over igemm2_kernel's 16x16x4 fragments: one activation
// gather (a dword per lane: 32 consecutive columns of two k rows) feeds a 64-clock MFMA instead of a 32-clock one, and a weight
// float4 feeds NT * 4 of them -- 20 vector-memory instructions per 32 MFMAs of 64 clocks (2 x 2 blocks) against 10 per 16 of 32
// clocks: half the instructions per matrix-pipe clock, and an fp32 MFMA hides none of them (tests/tools/mfma_overlap_probe.hip).
// Operand order inside a 16-deep chunk: MFMA (u, j) takes k = (2u + ks) * 4 + j, ks = lane >> 5 -- the weights come straight from the
// 16-row fragment packing (lane (row r, k-slot ks) reads the float4 of fragment r >> 4, quad 2u + ks), as in igemm32_kernel.
// LNB (round 6; 32 x 32 tile, K split only): the layer consumes a NOT yet normalised tensor -- igemm2_kernel's folded LayerNorm (IgemmP::ln_wsum) on this
// kernel, for the one-stream QKV / first FFN projections (isolated 12.4 / 13.4 -> 9.5 / 10.0 us against igemm2_kernel's LNB tiles).  A lane adds up its
// operand values relative to the column's first element (2 of the 4 k rows of every MFMA are this lane's, the partner lane holds the others), the K shares
// meet in LDS next to the partial tiles in wave order, and the correction rstd * (acc - mean * wsum[m]) is applied in front of the epilogue.
template <int MT, int NT, int KS, bool LNB = false>
__global__ __launch_bounds__(KS * 64) void igemm2w_kernel(IgemmP p)
{
    static_assert(!LNB || (MT == 1 && NT == 1 && KS > 1), "LayerNorm-consumer instantiations: 32 x 32 wave tile with the in-workgroup K split");
    constexpr int D = MT * NT >= 4 ? 2 : 3;             // chunks in flight per wave
    constexpr int TE = MT * NT * 1024;                  // elements of the wave tile
    constexpr int PE = KS > 1 ? (TE + KS * 64 - 1) / (KS * 64) : 1;     // elements a thread finishes after the reduction
    extern __shared__ __attribute__((aligned(16))) float s_red[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int fast = (int)blockIdx.x, slow = (int)blockIdx.y;
    const int tm = p.m_fast ? fast : slow, tn = p.m_fast ? slow : fast;
    if (tm >= p.ntm || tn >= p.ntn) return;             // padding of the XCD-aware order (uniform per workgroup)
    const PhaseD &ph = p.ph0;
    const int nchunks = ph.nchunks;
    int c0 = 0, nc = nchunks;
    if (KS > 1) {
        const int cpw = (nchunks + KS - 1) / KS;
        c0 = wave * cpw;
        int c1 = c0 + cpw;
        c1 = c1 < nchunks ? c1 : nchunks;
        nc = c1 > c0 ? c1 - c0 : 0;
    }
    const int c32 = lane & 31, ks = lane >> 5;
    const float *resb = p.res;
    float *yb = p.y + ph.y_off;
    // activations: wave-uniform base + 32-bit byte offset per lane (column of this lane, k rows c0 * 16 + ks * 4 ...)
    const char *xb = reinterpret_cast<const char *>(p.x + ph.x_off);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xb), 0, 0x7ffff000, 0x00020000);      // raw buffer, 32-bit data format
    const unsigned lin1 = (unsigned)p.lin_cs4;
    // (one byte offset per gather of a chunk, fixed for the whole launch: the chunk's position goes into the wave-uniform base pointer, so a gather
    //  costs no vector ALU instruction)
    unsigned xo[NT][8];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        int n = (tn * NT + nt) * 32 + c32;
        n = n < p.N ? n : p.N - 1;
        int bb = 0;
        if (p.fold_n) { bb = n / p.fold_n; n -= bb * p.fold_n; }
        const unsigned o = (unsigned)(bb * (int)p.x_bs + n * p.x_ws) * 4u + (unsigned)(c0 * 16 + ks * 4) * lin1;
#pragma unroll
        for (int q = 0; q < 8; q++) xo[nt][q] = o + (unsigned)((q >> 2) * 8 + (q & 3)) * lin1;
    }
    // weights: the same addressing (all vector-memory instructions of the loop are buffer loads: one kind of load, so the compiler's
    // s_waitcnt counts stay exact -- with global loads for the weights next to buffer loads for the activations it waited for the NEWEST
    // stage's loads before the oldest stage's MFMAs)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w + ph.w_off), 0, 0x7ffff000, 0x00020000);
    unsigned wo[MT];
    const int mtiles = (p.M + 15) >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int t16 = (tm * MT + mt) * 2 + (c32 >> 4);
        t16 = t16 < mtiles ? t16 : mtiles - 1;
        wo[mt] = (unsigned)(((long long)t16 * nchunks + c0) * 256 + (ks * 16 + (c32 & 15)) * 4) * 4u;
    }
    f32x16w acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;
    // LNB: sums of (value - first element of the column) and of its square over this lane's k rows (a buffer load like every other load of the kernel)
    float ln_s = 0.f, ln_ss = 0.f, ln_c = 0.f;
    float ln_ws[LNB ? PE : 1];                      // wsum of the rows this thread finishes: requested now, consumed behind the reduction
    if (LNB) {
        ln_c = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)(xo[0][0] - (unsigned)(c0 * 16 + ks * 4) * lin1), 0, 0));
        const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.ln_wsum + ph.bias_off), 0, 0x7ffff000, 0x00020000);
#pragma unroll
        for (int q = 0; q < PE; q++) {
            const int e = (int)threadIdx.x + q * KS * 64;
            int m_ = tm * 32 + ((e >> 6) & 3) + 8 * (((e >> 6) & 15) >> 2) + 4 * ((e & 63) >> 5);
            m_ = m_ < p.M ? m_ : p.M - 1;
            ln_ws[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, m_ * 4, 0, 0));
        }
    }
    f32x4 a_st[D][MT][2];
    float b_st[D][NT][8];
    auto load = [&](f32x4 (&a)[MT][2], float (&b)[NT][8], const int c) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int u = 0; u < 2; u++) a[mt][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wo[mt], c * 1024 + u * 512, 0));
        // buffer loads: per-lane byte offset in a VGPR that never changes, the chunk's offset in an SGPR -- no vector ALU work per gather
        const int so = (int)((unsigned)c * 16u * lin1);
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int q = 0; q < 8; q++) b[nt][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, (int)xo[nt][q], so, 0));
    };
    auto compute = [&](const f32x4 (&a)[MT][2], const float (&b)[NT][8]) {
        if (LNB) {
#pragma unroll
            for (int q = 0; q < 8; q++) { const float d_ = b[0][q] - ln_c; ln_s += d_; ln_ss = fmaf(d_, d_, ln_ss); }
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][u][j], b[nt][u * 4 + j], acc[mt][nt], 0, 0, 0);
    };
    // (the first D stages leave unconditionally and in stage order: the compiler's s_waitcnt counts at the loop head are the more conservative of the
    //  ways into the loop, and with a conditional prologue -- a path on which the second stage was never requested -- every iteration waited for the
    //  NEWEST stage's loads before computing the oldest)
#pragma unroll
    for (int s = 0; s < D; s++) {
        load(a_st[s], b_st[s], s < nc ? s : (nc > 0 ? nc - 1 : 0));      // unconditional (a stage past the wave's range re-reads its last chunk and is never used)
        __builtin_amdgcn_sched_barrier(0);
    }
    int c = 0;
    for (; c + 2 * D <= nc; c += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            compute(a_st[s], b_st[s]);
            __builtin_amdgcn_sched_barrier(0);
            load(a_st[s], b_st[s], c + s + D);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (; c < nc; c += D) {
#pragma unroll
        for (int s = 0; s < D; s++) {
            if (c + s < nc) {
                compute(a_st[s], b_st[s]);
                if (c + s + D < nc) load(a_st[s], b_st[s], c + s + D);
            }
        }
    }
    // C / D layout of the 32x32 block: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (KS > 1) {
        // fixed-order reduction of the KS partial tiles through LDS; every thread then finishes its share of the tile (consecutive
        // threads = consecutive columns: coalesced stores)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) s_red[wave * TE + ((mt * NT + nt) * 16 + r) * 64 + lane] = acc[mt][nt][r];
        float *lst = s_red + KS * TE;               // LNB: [KS][32 columns][2] sums of this wave's K share
        if (LNB) {
            const float s_ = ln_s + __shfl_xor(ln_s, 32, 64), q_ = ln_ss + __shfl_xor(ln_ss, 32, 64);        // the partner lane holds the other two k rows of every MFMA
            if (ks == 0) { lst[(wave * 32 + c32) * 2] = s_; lst[(wave * 32 + c32) * 2 + 1] = q_; }
        }
        __syncthreads();
        // LNB: statistics of this thread's column -- every element a thread finishes lies in column threadIdx.x & 31 = this lane's operand column --, the
        // K shares summed in wave order; the tm == 0 workgroups publish (mean, rstd) for the later layer whose residual is LayerNorm(y)
        float ln_mean = 0.f, ln_rstd = 0.f;
        if (LNB) {
            float s_ = 0.f, q_ = 0.f;
#pragma unroll
            for (int w = 0; w < KS; w++) { s_ += lst[(w * 32 + c32) * 2]; q_ += lst[(w * 32 + c32) * 2 + 1]; }
            const float msh = s_ * p.ln_inv_rows;
            const float var = fmaxf(q_ * p.ln_inv_rows - msh * msh, 0.f);
            ln_mean = ln_c + msh; ln_rstd = 1.0f / sqrtf(var + p.ln_eps);
            const int n_ = tn * 32 + c32;
            if (p.ln_stats_out && tm == 0 && threadIdx.x < 32 && n_ < p.N) { p.ln_stats_out[2 * n_] = ln_mean; p.ln_stats_out[2 * n_ + 1] = ln_rstd; }
        }
        // (batches of eight: a whole-share batch of 32 elements took the 64 x 64 tile with two waves to 288 registers)
        constexpr int EB = PE < 8 ? PE : 8;
        RVC_ACT_DISPATCH(
            for (int q0 = 0; q0 < PE; q0 += EB) {
                float v[EB];
                Epi2 ep[EB];
                _Pragma("unroll") for (int qq = 0; qq < EB; qq++) {
                    const int e = (int)threadIdx.x + (q0 + qq) * KS * 64;
                    const bool in = q0 + qq < PE && e < TE;
                    float sum = 0.f;
                    if (in) {
                        _Pragma("unroll") for (int w = 0; w < KS; w++) sum += s_red[w * TE + e];
                    }
                    const int l = e & 63; const int r = (e >> 6) & 15; const int f = e >> 10; const int mt = f / NT; const int nt = f - mt * NT;
                    if (LNB && in) sum = ln_rstd * (sum - ln_mean * ln_ws[LNB ? q0 + qq : 0]);          // the folded LayerNorm: out = rstd[n] * (acc - mean[n] * wsum[m]) (+ the folded bias in the epilogue)
                    v[qq] = sum;
                    ep[qq] = in ? epi2_prefetch(p, ph, resb, yb, (tm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), (tn * NT + nt) * 32 + (l & 31)) : epi2_none();
                }
                _Pragma("unroll") for (int qq = 0; qq < EB; qq++) epi2_finish<A_>(p, yb, v[qq], ep[qq]);
            }
        )
        return;
    }
    ColOut cols[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) cols[nt] = col_locate(p, ph, (tn * NT + nt) * 32 + c32);
    RVC_ACT_DISPATCH(
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++) {
            float bias_r[16];
            _Pragma("unroll") for (int r = 0; r < 16; r++) {
                const int m = (tm * MT + mt) * 32 + ks * 4 + (r & 3) + 8 * (r >> 2);
                bias_r[r] = (p.bias && m < p.M) ? p.bias[ph.bias_off + m] : 0.f;
            }
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {
                _Pragma("unroll") for (int h = 0; h < 16; h += 8) {
                    Epi2 e_[8];
                    _Pragma("unroll") for (int r = 0; r < 8; r++)
                        e_[r] = epi2_aux(p, ph, resb, yb, cols[nt], (tm * MT + mt) * 32 + ks * 4 + ((h + r) & 3) + 8 * ((h + r) >> 2), bias_r[h + r]);
                    _Pragma("unroll") for (int r = 0; r < 8; r++) epi2_finish<A_>(p, yb, acc[mt][nt][h + r], e_[r]);
                }
            }
        }
    )
}

// Throughput-mode implicit GEMM (many streams batched: N = B*T is large).  Classic CDNA anatomy: a 256-thread
// workgroup owns a (WM*MF*16) x (WN*NF*16) tile; per K step of 16 the gathered activation tile [16][BN] is staged
// global -> registers -> LDS (double-buffered, one barrier per step, fused input LeakyReLU applied once per element)
// and read back as MFMA B fragments by all waves; weight fragments stream global -> registers in fragment order.
// Each activation element is fetched once per workgroup instead of once per wave.
template <int WM, int WN, int MF, int NF, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void igemm_lds_kernel(IgemmP p)
{
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
    constexpr int RS = BN + 4;                 // LDS row stride: the 4 k-rows read by one MFMA operand fall on disjoint bank groups
    constexpr int KR = 256 / BN > 0 ? 256 / BN : 1;   // k rows staged per pass
    constexpr int EPT = 16 / KR;               // staged elements per thread per K step
    static_assert(BN <= 256 && 256 % BN == 0, "BN must divide 256");
    extern __shared__ __attribute__((aligned(16))) int s_mem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // m fastest when the streams are folded into N, over XCD-local tile ids (xcd_tile_id): the m-tiles of ONE activation tile run on one XCD
    const int tid_x = p.m_fast == 1 ? xcd_tile_id((int)blockIdx.x, (int)gridDim.x, (int)(blockIdx.y * gridDim.x)) : (int)blockIdx.x;
    const int tn = p.m_fast ? tid_x / p.ntm : tid_x % p.ntn, tm = p.m_fast ? tid_x % p.ntm : tid_x / p.ntn;
    int z = blockIdx.y;
    const int phase = z % p.nphase;
    const int b = z / p.nphase;
    const PhaseD ph = p.nphase == 1 ? p.ph0 : p.ph[phase];
    const int nchunks = ph.nchunks;
    {
        const int4 *src = reinterpret_cast<const int4 *>(p.koff + ph.koff_off);
        int4 *dst = reinterpret_cast<int4 *>(s_mem);
        for (int i = threadIdx.x; i < nchunks * 4; i += 256) dst[i] = src[i];
    }
    float *bt = reinterpret_cast<float *>(s_mem + nchunks * 16);     // [2][16][RS]
    const int li = lane & 15, kq = lane >> 4;
    const char *xb = reinterpret_cast<const char *>(p.x + (long long)b * p.x_bs + ph.x_off) - p.koff_bias;
    // staging role: column n_s of the tile, k rows kr0, kr0 + KR, ...
    const int n_s = threadIdx.x % BN, kr0 = threadIdx.x / BN;
    unsigned xo_s;
    {
        int n = tn * BN + n_s;
        n = n < p.N ? n : p.N - 1;
        int bb = 0;
        if (p.fold_n) { bb = n / p.fold_n; n -= bb * p.fold_n; }
        int nh = 0, nw = n;
        if (p.x_hs) { nh = n / p.NW; nw = n - nh * p.NW; }
        xo_s = (unsigned)(bb * (int)p.x_bs + nh * p.x_hs + nw * p.x_ws) * 4u;
    }
    const float *wrow[MF];
    const int mtiles = (p.M + 15) >> 4;
#pragma unroll
    for (int mf = 0; mf < MF; mf++) {
        int mt = (tm * WM + wm) * MF + mf;
        mt = mt < mtiles ? mt : mtiles - 1;
        wrow[mf] = p.w + ph.w_off + (long long)mt * nchunks * 256 + lane * 4;
    }
    f32x4 acc[MF][NF];
#pragma unroll
    for (int mf = 0; mf < MF; mf++)
#pragma unroll
        for (int nf = 0; nf < NF; nf++) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float pre_slope = p.pre_slope;
    __syncthreads();
    const int *kof = s_mem;
    float sb[EPT];
    f32x4 a_cur[MF], a_nxt[MF];
#pragma unroll
    for (int i = 0; i < EPT; i++) sb[i] = *reinterpret_cast<const float *>(xb + (xo_s + (unsigned)kof[kr0 + i * KR]));
#pragma unroll
    for (int mf = 0; mf < MF; mf++) a_cur[mf] = *reinterpret_cast<const f32x4 *>(wrow[mf]);
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const float v = sb[i];
        bt[(kr0 + i * KR) * RS + n_s] = PRE ? fmaxf(v, v * pre_slope) : v;
    }
    __syncthreads();
    const float *br = bt + wn * NF * 16 + li + kq * 4 * RS;
    for (int c = 0; c < nchunks; c++) {
        const int cn = c + 1 < nchunks ? c + 1 : c;
        const float *bcur = br + (c & 1) * 16 * RS;
        float *bnxt = bt + ((c + 1) & 1) * 16 * RS;
        // prefetch the next K step (global -> registers)
#pragma unroll
        for (int i = 0; i < EPT; i++) sb[i] = *reinterpret_cast<const float *>(xb + (xo_s + (unsigned)kof[cn * 16 + kr0 + i * KR]));
#pragma unroll
        for (int mf = 0; mf < MF; mf++) a_nxt[mf] = *reinterpret_cast<const f32x4 *>(wrow[mf] + cn * 256);
        float bv[NF][4];
#pragma unroll
        for (int nf = 0; nf < NF; nf++)
#pragma unroll
            for (int j = 0; j < 4; j++) bv[nf][j] = bcur[j * RS + nf * 16];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int mf = 0; mf < MF; mf++)
#pragma unroll
                for (int nf = 0; nf < NF; nf++)
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[mf][j], bv[nf][j], acc[mf][nf], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            const float v = sb[i];
            bnxt[(kr0 + i * KR) * RS + n_s] = PRE ? fmaxf(v, v * pre_slope) : v;
        }
#pragma unroll
        for (int mf = 0; mf < MF; mf++) a_cur[mf] = a_nxt[mf];
        __syncthreads();
    }
    const float *resb = p.res ? p.res + (long long)b * p.res_bs : nullptr;
    float *yb = p.y + (long long)b * p.y_bs;
    ColOut cols[NF];
#pragma unroll
    for (int nf = 0; nf < NF; nf++) cols[nf] = col_locate(p, ph, tn * BN + (wn * NF + nf) * 16 + li);
    // operands in store-free batches (see igemm32_kernel's epilogue): the four biases of a 16-row fragment, then per 16x16 block
    if (p.glu) {
#pragma unroll
        for (int mf = 0; mf < MF; mf++) {
            float gb[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = ((tm * WM + wm) * MF + mf) * 16 + kq * 4 + r;
                gb[r] = m < p.M ? p.bias[ph.bias_off + m] : 0.f;
            }
#pragma unroll
            for (int nf = 0; nf < NF; nf++)
#pragma unroll
                for (int r = 0; r < 2; r++)
                    glu_from_col_b(p, ph, yb, cols[nf], ((tm * WM + wm) * MF + mf) * 16 + kq * 4 + r, acc[mf][nf][r], acc[mf][nf][r + 2], gb[r], gb[r + 2]);
        }
        return;
    }
    const bool has_aux = resb != nullptr || p.accumulate;
    if (!p.accumulate && (tm * WM + wm + 1) * MF * 16 <= p.M) {
        // common case: small straight-line code (see igemm32_kernel's epilogue)
        const float slope = p.slope, scale = p.scale;
        const long long cs = p.y_cs, rcs = p.res_cs;
        RVC_ACT_DISPATCH(
            _Pragma("unroll") for (int mf = 0; mf < MF; mf++) {
                const int m0 = ((tm * WM + wm) * MF + mf) * 16 + kq * 4;
                float bias_r[4];
                _Pragma("unroll") for (int r = 0; r < 4; r++) bias_r[r] = p.bias ? p.bias[ph.bias_off + m0 + r] : 0.f;
                float rr[NF][4];
                _Pragma("unroll") for (int nf = 0; nf < NF; nf++)
                    _Pragma("unroll") for (int r = 0; r < 4; r++) rr[nf][r] = 0.f;
                if (resb) {
                    _Pragma("unroll") for (int nf = 0; nf < NF; nf++)
                        if (cols[nf].yo >= 0) {
                            const float *rp = resb + cols[nf].ro + (long long)(p.res_nogroup ? m0 : m0 + ph.y_c0) * rcs;
                            _Pragma("unroll") for (int r = 0; r < 4; r++) rr[nf][r] = rp[r * rcs];
                        }
                }
                _Pragma("unroll") for (int nf = 0; nf < NF; nf++)
                    if (cols[nf].yo >= 0) {
                        float *yc = yb + cols[nf].yo + (long long)(m0 + ph.y_c0) * cs;
                        _Pragma("unroll") for (int r = 0; r < 4; r++)
                            yc[r * cs] = epi2_value<A_>(acc[mf][nf][r], bias_r[r], rr[nf][r], 0.f, slope, scale);
                    }
            }
        )
        return;
    }
    RVC_ACT_DISPATCH(
        _Pragma("unroll") for (int mf = 0; mf < MF; mf++) {
            float bias_r[4];
            _Pragma("unroll") for (int r = 0; r < 4; r++) {
                const int m = ((tm * WM + wm) * MF + mf) * 16 + kq * 4 + r;
                bias_r[r] = (p.bias && m < p.M) ? p.bias[ph.bias_off + m] : 0.f;
            }
            _Pragma("unroll") for (int nf = 0; nf < NF; nf++) {
                if (!has_aux) {
                    _Pragma("unroll") for (int r = 0; r < 4; r++) {
                        const int m = ((tm * WM + wm) * MF + mf) * 16 + kq * 4 + r;
                        const Epi2 e1 = epi2_plain(bias_r[r], (m < p.M && cols[nf].yo >= 0) ? cols[nf].yo + (m + ph.y_c0) * p.y_cs : -1);
                        epi2_finish<A_>(p, yb, acc[mf][nf][r], e1);
                    }
                } else {
                    Epi2 e_[4];
                    _Pragma("unroll") for (int r = 0; r < 4; r++)
                        e_[r] = epi2_aux(p, ph, resb, yb, cols[nf], ((tm * WM + wm) * MF + mf) * 16 + kq * 4 + r, bias_r[r]);
                    _Pragma("unroll") for (int r = 0; r < 4; r++) epi2_finish<A_>(p, yb, acc[mf][nf][r], e_[r]);
                }
            }
        }
    )
}

// ------------------------------------------------------------------------------------------------------------------------
// igemm32_kernel -- throughput-mode implicit GEMM on v_mfma_f32_32x32x2_f32 (many streams: N = streams * positions is large).
// A 256-thread workgroup owns a (WM*MT*32) x (WN*NT*32) tile; every wave a (MT*32) x (NT*32) block of 32x32 accumulators
// (16 registers each).  Per K step of 16:
//   * the gathered activation tile [16][BN] goes global -> registers -> LDS once per workgroup (double-buffered, ONE barrier per
//     step, fused input LeakyReLU applied once per element) and is read back as MFMA B operands by all four waves;
//   * the weights stream global -> registers straight from the 16-row fragment packing the latency kernels use: for a 32-row
//     MFMA the lane (row r, k-slot s) reads the float4 of fragment r>>4, quad q = 2u+s (u = 0, 1), so the eight MFMAs of a step
//     take k = (2u+s)*4 + j -- A and B agree on that order, no repacking, no shuffles;
//   * 32x32x2 halves the MFMA instruction count of the 16x16x4 form and has no dependent-issue gap (64-cycle issue = 64-cycle
//     accumulator latency), so one wave per SIMD already saturates the pipe while the next step's loads are in flight.
// Epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): a lane owns 16 rows of ONE column, so
// the column part of the address (stream, row, validity) is computed once per 32-column block.
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Wide register tiles were measured in round 5 and are not instantiated: MT x NT = 4 x 2 (a wave owns 128 x 64 outputs, 222 registers, two waves per
// SIMD) runs the 64-stream layers at the speed of the 2 x 2 tile (+-1 %; +5 % on the one layer whose grid it rounds better), 4 x 4 (256 accumulator
// registers) spills -- DESIGN.md section 7 round 5.  The occupancy rule and the loop below stay general.  The K loop is unrolled by two so that the weight
// registers of consecutive steps alternate instead of being copied.
template <int MT, int NT> struct G32Occ { static constexpr int W = MT * NT >= 16 ? 1 : (MT * NT >= 8 ? 2 : 3); };
template <int WM, int WN, int MT, int NT, bool PRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(G32Occ<MT, NT>::W, G32Occ<MT, NT>::W))) void igemm32_kernel(IgemmP p)
{
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    // The staged activation tile is kept COLUMN-major, [BN][16 k + 4 pad]: the four k-values (2u + ks) * 4 + j, j = 0..3, that a lane feeds to four
    // consecutive MFMAs are one ds_read_b128, and a staging thread owns EPT consecutive k rows of one column (ds_write_b128).  On this chip an fp32
    // MFMA hides none of its SIMD's other instructions (tests/tools/mfma_overlap_probe.hip: +16 clocks per ds_read_b32, +4 per VALU, +32 per 16-byte
    // vector load, at any occupancy), so the row-major tile's 16 ds_read_b32 + 8 ds_write_b32 per K step were a sixth of the step.
    constexpr int RSK = 20;                        // column stride in floats: 16-byte aligned, 16 lanes x 16 bytes on disjoint banks
    constexpr int KR = 256 / BN > 0 ? 256 / BN : 1;    // staging threads per column
    constexpr int EPT = 16 / KR;                   // consecutive k rows per staging thread per K step (4, 8 or 16)
    static_assert(BN <= 256 && 256 % BN == 0, "BN must divide 256");
    extern __shared__ __attribute__((aligned(16))) int s_mem[];
    RVC_KP(0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tid_x = p.m_fast == 1 ? xcd_tile_id((int)blockIdx.x, (int)gridDim.x, (int)(blockIdx.y * gridDim.x)) : (int)blockIdx.x;
    const int tn = p.m_fast ? tid_x / p.ntm : tid_x % p.ntn, tm = p.m_fast ? tid_x % p.ntm : tid_x / p.ntn;
    int z = blockIdx.y;
    const int phase = z % p.nphase;
    const int b = z / p.nphase;
    const PhaseD ph = p.nphase == 1 ? p.ph0 : p.ph[phase];
    const int nchunks = ph.nchunks;
    {
        const int4 *src = reinterpret_cast<const int4 *>(p.koff + ph.koff_off);
        int4 *dst = reinterpret_cast<int4 *>(s_mem);
        for (int i = threadIdx.x; i < nchunks * 4; i += 256) dst[i] = src[i];
    }
    float *bt = reinterpret_cast<float *>(s_mem + nchunks * 16);     // [2][BN][RSK]
    const int c32 = lane & 31, ks = lane >> 5;                       // MFMA column (B) / row (A) and k-slot of this lane
    const char *xb = reinterpret_cast<const char *>(p.x + (long long)b * p.x_bs + ph.x_off) - p.koff_bias;
    // staging role: column n_s of the tile, k rows kr0 .. kr0 + EPT - 1
    const int n_s = threadIdx.x % BN, kr0 = (threadIdx.x / BN) * EPT;
    unsigned xo_s;
    {
        int n = tn * BN + n_s;
        n = n < p.N ? n : p.N - 1;
        int bb = 0;
        if (p.fold_n) { bb = n / p.fold_n; n -= bb * p.fold_n; }
        int nh = 0, nw = n;
        if (p.x_hs) { nh = n / p.NW; nw = n - nh * p.NW; }
        xo_s = (unsigned)(bb * (int)p.x_bs + nh * p.x_hs + nw * p.x_ws) * 4u;
    }
    // weights: 16-row fragment packing [m_tile16][chunk][lane16x4][4]; this lane's float4 of quad q sits at ((q * 16 + r16) * 4)
    const float *wrow[MT];
    const int mtiles = (p.M + 15) >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int t16 = ((tm * WM + wm) * MT + mt) * 2 + (c32 >> 4);
        t16 = t16 < mtiles ? t16 : mtiles - 1;
        wrow[mt] = p.w + ph.w_off + (long long)t16 * nchunks * 256 + (ks * 16 + (c32 & 15)) * 4;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;
    const float pre_slope = p.pre_slope;
    __syncthreads();
    RVC_KP(1);
    const int *kof = s_mem;
    float sb[EPT];
    f32x4 a_ev[MT][2], a_od[MT][2];          // weights of the even / odd K steps (the loop below is unrolled by two: no copies)
#pragma unroll
    for (int i = 0; i < EPT; i++) sb[i] = *reinterpret_cast<const float *>(xb + (xo_s + (unsigned)kof[kr0 + i]));
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int u = 0; u < 2; u++) a_ev[mt][u] = *reinterpret_cast<const f32x4 *>(wrow[mt] + u * 128);
#pragma unroll
    for (int i = 0; i < EPT; i += 4) {
        f32x4 v4;
#pragma unroll
        for (int q = 0; q < 4; q++) v4[q] = PRE ? fmaxf(sb[i + q], sb[i + q] * pre_slope) : sb[i + q];
        *reinterpret_cast<f32x4 *>(bt + n_s * RSK + kr0 + i) = v4;
    }
    __syncthreads();
    RVC_KP(2);
    // B operand of MFMA (u, j) for column block nt: row k = (2u + ks) * 4 + j of the staged tile
    const float *br = bt + (wn * NT * 32 + c32) * RSK + ks * 4;
    // one K step: request the next step's operands (global -> registers), run this step's MFMAs from a_c and the staged tile, stage the next tile
    auto kstep = [&](const int c, f32x4 (&a_c)[MT][2], f32x4 (&a_n)[MT][2]) {
        const int cn = c + 1 < nchunks ? c + 1 : c;
        const float *bcur = br + (c & 1) * BN * RSK;
        float *bnxt = bt + ((c + 1) & 1) * BN * RSK;
#pragma unroll
        for (int i = 0; i < EPT; i++) sb[i] = *reinterpret_cast<const float *>(xb + (xo_s + (unsigned)kof[cn * 16 + kr0 + i]));
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int u = 0; u < 2; u++) a_n[mt][u] = *reinterpret_cast<const f32x4 *>(wrow[mt] + cn * 256 + u * 128);
        __builtin_amdgcn_sched_barrier(0);      // (the next step's requests stay in FRONT of this step's MFMAs: left to itself the scheduler sinks them behind most of them)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            f32x4 bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) bv[nt] = *reinterpret_cast<const f32x4 *>(bcur + nt * 32 * RSK + u * 8);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[mt][u][j], bv[nt][j], acc[mt][nt], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < EPT; i += 4) {
            f32x4 v4;
#pragma unroll
            for (int q = 0; q < 4; q++) v4[q] = PRE ? fmaxf(sb[i + q], sb[i + q] * pre_slope) : sb[i + q];
            *reinterpret_cast<f32x4 *>(bnxt + n_s * RSK + kr0 + i) = v4;
        }
        __syncthreads();
    };
    {
        int c = 0;
        for (; c + 2 <= nchunks; c += 2) { kstep(c, a_ev, a_od); kstep(c + 1, a_od, a_ev); }
        if (c < nchunks) kstep(c, a_ev, a_od);
    }
    RVC_KP(3);
    const float *resb = p.res ? p.res + (long long)b * p.res_bs : nullptr;
    float *yb = p.y + (long long)b * p.y_bs;
    ColOut cols[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) cols[nt] = col_locate(p, ph, tn * BN + (wn * NT + nt) * 32 + c32);
    const int row0 = (tm * WM + wm) * MT * 32 + ks * 4;          // + mt * 32 + (reg & 3) + 8 * (reg >> 2)
    // Epilogue operands are loaded in batches that contain no store: a load may not be moved above an earlier store (the pointers
    // may alias), so "load bias, store, load bias, store, ..." is one memory round trip per element -- 64 of them per lane, measured
    // 68 us of a 300 us wave lifetime on the ContentVec convolutions.  Batched: one round trip for the 16 biases of a 32-row block,
    // one per 32x32 block for the residual / accumulate operands (none for most layers).  The batches are kept this small on
    // purpose: a whole-tile batch took the kernel from 164 to 268 registers (3 -> 1 waves per SIMD, 1.5x slower overall).
    const bool has_aux = resb != nullptr || p.accumulate;
    const bool full_m = row0 - ks * 4 + MT * 32 <= p.M;          // every row of this wave's tile exists (wave-uniform)
    if (full_m) {
        // The common case, kept small on purpose: straight-line code per element is what the 64-element unrolled epilogue costs
        // in instruction-cache footprint (the general version below is ~10x larger; with every workgroup of the chip walking
        // through it at a different point the epilogue took 43 us per wave, most of it instruction fetch).  One predicate per
        // 32-column block, the block's residual operands (if any) in one batch, 16 stores at scalar row offsets from one
        // per-lane base address.
        const float slope = p.slope, scale = p.scale;
        const long long cs = p.y_cs, rcs = p.res_cs;
        RVC_ACT_DISPATCH(
            _Pragma("unroll") for (int mt = 0; mt < MT; mt++) {
                const int m0 = row0 + mt * 32;
                float bias_r[16];
                _Pragma("unroll") for (int r = 0; r < 16; r++)
                    bias_r[r] = p.bias ? p.bias[ph.bias_off + m0 + (r & 3) + 8 * (r >> 2)] : 0.f;
                _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {
                    if (cols[nt].yo >= 0) {
                        float rr[16];
                        _Pragma("unroll") for (int r = 0; r < 16; r++) rr[r] = 0.f;
                        if (resb) {
                            const float *rp = resb + cols[nt].ro + (long long)(p.res_nogroup ? m0 : m0 + ph.y_c0) * rcs;
                            _Pragma("unroll") for (int r = 0; r < 16; r++) rr[r] = rp[((r & 3) + 8 * (r >> 2)) * rcs];
                        }
                        float *yc = yb + cols[nt].yo + (long long)(m0 + ph.y_c0) * cs;
                        float yo_[16];          // (round 6: accumulating launches take this path too -- the previous output as one more store-free batch)
                        _Pragma("unroll") for (int r = 0; r < 16; r++) yo_[r] = 0.f;
                        if (p.accumulate) { _Pragma("unroll") for (int r = 0; r < 16; r++) yo_[r] = yc[((r & 3) + 8 * (r >> 2)) * cs]; }
                        _Pragma("unroll") for (int r = 0; r < 16; r++)
                            yc[((r & 3) + 8 * (r >> 2)) * cs] = epi2_value<A_>(acc[mt][nt][r], bias_r[r], rr[r], yo_[r], slope, scale);
                    }
                }
            }
        )
        RVC_KP(6);
        return;
    }
    RVC_ACT_DISPATCH(
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++) {
            float bias_r[16];
            _Pragma("unroll") for (int r = 0; r < 16; r++) {
                const int m = row0 + mt * 32 + (r & 3) + 8 * (r >> 2);
                bias_r[r] = (p.bias && m < p.M) ? p.bias[ph.bias_off + m] : 0.f;
            }
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {
                _Pragma("unroll") for (int h = 0; h < 16; h += 8) {
                    Epi2 e_[8];
                    _Pragma("unroll") for (int r = 0; r < 8; r++)
                        e_[r] = epi2_aux(p, ph, resb, yb, cols[nt], row0 + mt * 32 + ((h + r) & 3) + 8 * ((h + r) >> 2), bias_r[h + r]);
                    _Pragma("unroll") for (int r = 0; r < 8; r++) epi2_finish<A_>(p, yb, acc[mt][nt][h + r], e_[r]);
                }
            }
        }
    )
    RVC_KP(6);
}

// ------------------------------------------------------------------------------------------------------------------------
// igemm_bf3_kernel -- EXPLORATORY, off by default (rvc_set_gemm_precision), never part of the headline figure: the table-free 1x1 layers of ContentVec
// at many streams with every fp32 product replaced by three bf16 products, a * b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi (a_hi = bf16(a), a_lo = bf16(a - a_hi):
// 16 of the 24 mantissa bits of each operand; the dropped a_lo b_lo term is 2^-16 relative), accumulated in fp32.  Why: gfx950 has no xf32, its fp32 MFMA
// runs at 1/16 of the bf16 rate (v_mfma_f32_32x32x2_f32: 4 096 flops in 64 clocks; v_mfma_f32_32x32x16_bf16: 32 768 in 32), so three bf16 MFMAs per
// 16-deep chunk cost 96 clocks where the fp32 form costs 512.  The kernel is igemm32_kernel's anatomy (128 x 128 workgroup tile, 2 x 2 waves of 64 x 64,
// activation tile staged through LDS once per workgroup, one barrier per K step of 16):
//   * weights: split at plan time (bf3_pack_kernel) into panels [32-row block][chunk][hi | lo][lane][8 bf16] -- a lane's 16 bytes are its row's eight
//     k values (k = (lane >> 5) * 8 + i) of one half, a wave instruction reads 1 KB contiguous;
//   * activations: a staging thread gathers the eight k rows of one k group for its column (8 dword loads), splits them (v_cvt_pk_bf16_f32) and writes
//     two 16-byte pieces; LDS holds [hi | lo][column][k group 0 | k group 1 | pad] at 48 bytes per column (16 lanes x 16 bytes on disjoint banks);
//   * A and B use the SAME (lane >> 5, i) -> k assignment, which is all the instruction's sum over k needs.
// C / D layout = the 32x32x2 form's (dtype-independent on gfx950).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void bf3_split(const float (&v)[8], bf16x8 &hi, bf16x8 &lo)
{
#pragma unroll
    for (int i = 0; i < 8; i++) { hi[i] = (__bf16)v[i]; lo[i] = (__bf16)(v[i] - (float)hi[i]); }
}
static __global__ void bf3_pack_kernel(const float *wfrag, int M, int nchunks, bf16x8 *out, long long total)
{
    // out element (blk32, chunk, half, lane): eight bf16 of row blk32 * 32 + (lane & 31), k = chunk * 16 + (lane >> 5) * 8 + i, from the fp32
    // fragment packing [m_tile16][chunk][lane16x4][4]
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int lane = (int)(e & 63), half = (int)((e >> 6) & 1);
    const long long bc = e >> 7;
    const int chunk = (int)(bc % nchunks), blk = (int)(bc / nchunks);
    const int m = blk * 32 + (lane & 31), mt16 = (M + 15) >> 4;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int kk = (lane >> 5) * 8 + i;
        v[i] = (m >> 4) < mt16 ? wfrag[((((long long)(m >> 4) * nchunks + chunk) * 64) + (kk >> 2) * 16 + (m & 15)) * 4 + (kk & 3)] : 0.f;
    }
    bf16x8 hi, lo;
    bf3_split(v, hi, lo);
    out[e] = half ? lo : hi;
}
template <int WM, int WN, int MT, int NT, bool LIN, bool PRE>
__global__ __launch_bounds__(256) void igemm_bf3_kernel(IgemmP p)
{
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    static_assert(BN == 128, "two staging threads per column: one per k group");
    constexpr int CSB = 48;                         // bytes per column and half
    constexpr int HALF = BN * CSB, BUF = 2 * HALF;  // one half, one buffer (hi + lo)
    extern __shared__ __attribute__((aligned(16))) char s_raw[];     // [offset table: nchunks * 16 ints, unless LIN][2][hi | lo][BN][48]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tid_x = p.m_fast == 1 ? xcd_tile_id((int)blockIdx.x, (int)gridDim.x, (int)(blockIdx.y * gridDim.x)) : (int)blockIdx.x;
    const int tn = p.m_fast ? tid_x / p.ntm : tid_x % p.ntn, tm = p.m_fast ? tid_x % p.ntm : tid_x / p.ntn;
    const int phase = (int)blockIdx.y;              // (streams are folded into N: one batch)
    const PhaseD ph = p.nphase == 1 ? p.ph0 : p.ph[phase];
    const int nchunks = ph.nchunks;
    const int *kof = reinterpret_cast<const int *>(s_raw);
    char *s_bt = s_raw + (LIN ? 0 : nchunks * 64);
    if (!LIN) {
        const int4 *src = reinterpret_cast<const int4 *>(p.koff + ph.koff_off);
        int4 *dst = reinterpret_cast<int4 *>(s_raw);
        for (int i = threadIdx.x; i < nchunks * 4; i += 256) dst[i] = src[i];
        __syncthreads();
    }
    const int c32 = lane & 31, ks = lane >> 5;
    const char *xb = reinterpret_cast<const char *>(p.x + ph.x_off) - (LIN ? 0 : p.koff_bias);
    const unsigned lin1 = (unsigned)p.lin_cs4;
    const float pre_slope = p.pre_slope;
    const int n_s = threadIdx.x % BN, g_s = threadIdx.x / BN;
    unsigned xo_s;
    {
        int n = tn * BN + n_s;
        n = n < p.N ? n : p.N - 1;
        int bb = 0;
        if (p.fold_n) { bb = n / p.fold_n; n -= bb * p.fold_n; }
        xo_s = (unsigned)(bb * (int)p.x_bs + n * p.x_ws) * 4u + (LIN ? (unsigned)(g_s * 8) * lin1 : 0u);
    }
    const char *wb = reinterpret_cast<const char *>(p.w + ph.w_off);
    const int nblk = (p.M + 31) >> 5;
    long long wo[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int blk = (tm * WM + wm) * MT + mt;
        blk = blk < nblk ? blk : nblk - 1;
        wo[mt] = (long long)blk * nchunks * 2048 + lane * 16;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;
    float sb[8];
    bf16x8 a_ev[MT][2], a_od[MT][2];            // [mt][hi | lo] of the even / odd K steps
    auto gather = [&](const int c) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            sb[i] = *reinterpret_cast<const float *>(xb + (xo_s + (LIN ? (unsigned)(c * 16 + i) * lin1 : (unsigned)kof[c * 16 + g_s * 8 + i])));
    };
    auto stage = [&](char *buf) {
        if (PRE) {
#pragma unroll
            for (int i = 0; i < 8; i++) sb[i] = fmaxf(sb[i], sb[i] * pre_slope);      // fused input LeakyReLU, once per staged element
        }
        bf16x8 hi, lo;
        bf3_split(sb, hi, lo);
        *reinterpret_cast<bf16x8 *>(buf + n_s * CSB + g_s * 16) = hi;
        *reinterpret_cast<bf16x8 *>(buf + HALF + n_s * CSB + g_s * 16) = lo;
    };
    auto wload = [&](bf16x8 (&a)[MT][2], const int c) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int h = 0; h < 2; h++) a[mt][h] = *reinterpret_cast<const bf16x8 *>(wb + wo[mt] + (long long)c * 2048 + h * 1024);
    };
    gather(0);
    wload(a_ev, 0);
    stage(s_bt);
    __syncthreads();
    const char *br = s_bt + (wn * NT * 32 + c32) * CSB + ks * 16;
    auto kstep = [&](const int c, bf16x8 (&a_c)[MT][2], bf16x8 (&a_n)[MT][2]) {
        const int cn = c + 1 < nchunks ? c + 1 : c;
        const char *bcur = br + (c & 1) * BUF;
        gather(cn);
        wload(a_n, cn);
        __builtin_amdgcn_sched_barrier(0);      // (requests in front of the MFMAs, as in igemm32_kernel)
        bf16x8 bh[NT], bl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            bh[nt] = *reinterpret_cast<const bf16x8 *>(bcur + nt * 32 * CSB);
            bl[nt] = *reinterpret_cast<const bf16x8 *>(bcur + HALF + nt * 32 * CSB);
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                // the two small terms first, the leading term last
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c[mt][1], bh[nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c[mt][0], bl[nt], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c[mt][0], bh[nt], acc[mt][nt], 0, 0, 0);
            }
        stage(s_bt + ((c + 1) & 1) * BUF);
        __syncthreads();
    };
    {
        int c = 0;
        for (; c + 2 <= nchunks; c += 2) { kstep(c, a_ev, a_od); kstep(c + 1, a_od, a_ev); }
        if (c < nchunks) kstep(c, a_ev, a_od);
    }
    const float *resb = p.res;
    float *yb = p.y + ph.y_off;
    ColOut cols[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) cols[nt] = col_locate(p, ph, tn * BN + (wn * NT + nt) * 32 + c32);
    const int row0 = (tm * WM + wm) * MT * 32 + ks * 4;
    RVC_ACT_DISPATCH(
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++) {
            float bias_r[16];
            _Pragma("unroll") for (int r = 0; r < 16; r++) {
                const int m = row0 + mt * 32 + (r & 3) + 8 * (r >> 2);
                bias_r[r] = (p.bias && m < p.M) ? p.bias[ph.bias_off + m] : 0.f;
            }
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++) {
                _Pragma("unroll") for (int h = 0; h < 16; h += 8) {
                    Epi2 e_[8];
                    _Pragma("unroll") for (int r = 0; r < 8; r++)
                        e_[r] = epi2_aux(p, ph, resb, yb, cols[nt], row0 + mt * 32 + ((h + r) & 3) + 8 * ((h + r) >> 2), bias_r[h + r]);
                    _Pragma("unroll") for (int r = 0; r < 8; r++) epi2_finish<A_>(p, yb, acc[mt][nt][h + r], e_[r]);
                }
            }
        }
    )
}

}  // namespace rvc
