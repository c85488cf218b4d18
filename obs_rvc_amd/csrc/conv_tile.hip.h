// conv_tile.hip.h -- stride-1 1-D convolutions of ONE stream whose output is long (the HiFiGAN decoder's ResBlock chains: N = 2 520 .. 10 080
// columns, 32 .. 128 channels) as an implicit GEMM whose activation operand is staged ONCE per workgroup.
//
// Why (measured, tests/tools/ta_probe.hip + kprobe.py): the register-direct kernel (igemm2) gathers the B operand with one dword load per
// MFMA k-step and 16-column fragment -- a CU retires such a load every 9-12 clocks, so a 2 x 2 fragment tile spends ~500 clocks of
// vector-memory issue per 16-deep K chunk beside its 512 clocks of MFMA, after a 5 us prologue of table / address / first-gather latency.
// Here the workgroup copies the raw input rows it needs -- [Cin][BN + (KW - 1) * dil] floats -- into LDS with LDS-DMA loads
// (global_load_lds_dword: no registers, everything in flight at once), applies the fused input LeakyReLU once and turns the tile to
// channel-contiguous columns in the same pass.  K is walked tap-major in chunks of 16 channels (the weights are repacked to that order at
// plan time), so the four k-values a lane feeds to an MFMA are ONE ds_read_b128 at a wave-uniform offset from a per-lane base: no offset
// table, no per-element address arithmetic (a first version with [channel][column] rows, an LDS offset table and four ds_read_b32 per
// fragment spent ~40 issue slots per 8 MFMAs and ran at 460 clocks per chunk against 256 of MFMA time -- a wave hides about five
// instructions per MFMA).  The only vector-memory traffic of the main loop is the weight stream (fragment-major, 1 KB per load,
// prefetched DA chunks ahead).  The four waves split M (and N for 32-channel layers): no K split, no reduction, no barrier in the loop.
//
// Work items (phase, m-tile, n-tile) are handed to workgroups through a table built at plan time: the hardware places workgroup b on
// CU b % 256 (tests/tools/place_probe.hip), so the table pairs long items (kernel size 11) with short ones (kernel size 3) on a CU
// instead of leaving the balance to the dispatch order.
#pragma once
#include "igemm.hip.h"

namespace rvc {

#define COMMA_ ,
#ifndef RVC_CT_DBG
#define RVC_CT_DBG 0          // ablation builds (tests/tools/ct_ablate.sh): 1 no weight reloads, 2 no B reads, 8 no MFMAs (timing only)
#endif
template <int ACT, int MF, int NF>
__device__ __forceinline__ void conv_tile_store(const IgemmP &p, const PhaseD &ph, const float *resb, float *yb, const ColOut (&cols)[NF], int m_base, const f32x4 (&acc)[MF][NF])
{
    const float slope = p.slope, scale = p.scale;
    const long long cs = p.y_cs, rcs = p.res_cs;
#pragma unroll
    for (int mf = 0; mf < MF; mf++) {
        const int m0 = m_base + mf * 16;
        float bias_r[4];
#pragma unroll
        for (int r = 0; r < 4; r++) bias_r[r] = (p.bias && m0 + r < p.M) ? p.bias[ph.bias_off + m0 + r] : 0.f;
#pragma unroll
        for (int nf = 0; nf < NF; nf++) {
            if (cols[nf].yo < 0) continue;
            float rr[4] = {0.f, 0.f, 0.f, 0.f}, yo[4] = {0.f, 0.f, 0.f, 0.f};
            float *yc = yb + cols[nf].yo + (long long)(m0 + ph.y_c0) * cs;
            if (resb) {
                const float *rp = resb + cols[nf].ro + (long long)(p.res_nogroup ? m0 : m0 + ph.y_c0) * rcs;
#pragma unroll
                for (int r = 0; r < 4; r++) if (m0 + r < p.M) rr[r] = rp[r * rcs];
            }
            if (p.accumulate) {
#pragma unroll
                for (int r = 0; r < 4; r++) if (m0 + r < p.M) yo[r] = yc[r * cs];
            }
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (m0 + r < p.M) yc[r * cs] = epi2_value<ACT>(acc[mf][nf][r], bias_r[r], rr[r], yo[r], slope, scale);
        }
    }
}

// KS = 2: eight waves, the two waves that share a SIMD take the even / odd K chunks of the same fragments (one wave per SIMD cannot hide its own
// weight loads: their issue blocks behind the CU's single vector-memory path for 20-40 clocks each) and are summed through LDS at the end.
template <int WM, int WN, int MF, int NF, int KS>
__global__ __launch_bounds__(64 * WM * WN * KS) __attribute__((amdgpu_waves_per_eu(2))) void conv_tile_kernel(IgemmP p)
{
    static_assert(KS == 1 || KS == 2, "one or two K shares");
    constexpr int WPS = WM * WN, NT = 64 * WPS * KS;       // waves per K share, threads
    constexpr int BM = WM * MF * 16, BN = WN * NF * 16;
    constexpr int DA = MF == 1 ? 8 : 6;            // weight chunks in flight per wave
    extern __shared__ __attribute__((aligned(16))) int s_mem[];
    const int item = p.items[2 * blockIdx.x];
    if (item < 0) return;
    RVC_KP(0);
    const int phase = item & 0xff, tm = (item >> 8) & 0xff, tn = item >> 16;
    const int b = p.items[2 * blockIdx.x + 1];
    const PhaseD ph = p.nphase == 1 ? p.ph0 : p.ph[phase];
    const int nchunks = ph.nchunks;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int kh = wave / WPS, w4 = wave % WPS;     // K share, wave inside the share
    const int wm = w4 / WN, wn = w4 % WN;
    const int li = lane & 15, kq = lane >> 4;
    const int cin = ph.t_cin, rt = ph.t_rs, kw = ph.t_tab & 0xff, dil = ph.t_tab >> 8;
    const int rl = BN + (kw - 1) * dil;             // input columns the tile reads
    const int cs = cin + 8;                         // channel-contiguous tile: column stride (16 lanes x 16 bytes of a ds_read_b128 on disjoint banks)
    const int G = cin >> 4;
    float *tmp = reinterpret_cast<float *>(s_mem);  // [cin][rt] as it arrives (rt odd)
    const int tmp_n = (cin * rt + 63) & ~63;
    float *fin = tmp + (tmp_n > WPS * MF * NF * 256 ? tmp_n : WPS * MF * NF * 256);     // [rl][cs] activated, channel-contiguous (tmp later holds the K shares' partial sums)
    // 1. input rows -> LDS (LDS-DMA: wave-uniform destination + lane * 4; coalesced along the rows)
    {
        const float *xb = p.x + (long long)b * p.x_bs + ph.x_off;
        const int n0 = tn * BN + ph.t_dmin;
        const int total = cin * rt;
        const float inv_rt = 1.0f / (float)rt;
        for (int q = wave; q * 64 < total; q += WPS * KS) {
            const int e = q * 64 + lane;
            int row = (int)(((float)e + 0.5f) * inv_rt);
            const int col = e - row * rt;
            row = row < cin ? row : cin - 1;
            int gc = n0 + col;
            gc = gc < p.x_lo ? p.x_lo : (gc > p.x_lim ? p.x_lim : gc);
            __builtin_amdgcn_global_load_lds(xb + (long long)row * p.x_ld + gc, (__attribute__((address_space(3))) void *)(tmp + q * 64), 4, 0, 0);
        }
    }
    RVC_KP(1);
    // 2. weights: MFMA-fragment order [m_tile][chunk][lane][4] in THIS kernel's K order (chunk = tap * G + channel group, k = 4 * kq + j the
    //    channel inside the group: repacked at plan time); the first DA chunks leave behind the staging loads
    const float *wrow[MF];
    const int mtiles = (p.M + 15) >> 4;
#pragma unroll
    for (int mf = 0; mf < MF; mf++) {
        int mt = (tm * WM + wm) * MF + mf;
        mt = mt < mtiles ? mt : mtiles - 1;
        wrow[mf] = p.w + ph.w_off + (long long)mt * nchunks * 256;          // wave-uniform; the lane's 16 bytes are added at the load
    }
    const int nloc = (nchunks - kh + KS - 1) / KS;     // this wave's chunks: kh, kh + KS, ...  (>= 1: nchunks >= KS is checked at plan time)
    const unsigned lane16 = (unsigned)lane * 16u;
    const float *wnext[MF];              // wave-uniform: the chunk whose weights are requested next (scalar base + lane offset: no per-load address VALU)
#pragma unroll
    for (int mf = 0; mf < MF; mf++) wnext[mf] = wrow[mf] + (long long)(kh + DA * KS) * 256;
    f32x4 a_st[DA][MF];
#pragma unroll
    for (int s = 0; s < DA; s++) {
        const int cc = kh + (s < nloc ? s : nloc - 1) * KS;
#pragma unroll
        for (int mf = 0; mf < MF; mf++) a_st[s][mf] = *reinterpret_cast<const f32x4 *>(wrow[mf] + cc * 256 + lane * 4);
    }
    f32x4 acc[MF][NF];
#pragma unroll
    for (int mf = 0; mf < MF; mf++)
#pragma unroll
        for (int nf = 0; nf < NF; nf++) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float pre_slope = p.pre_slope;
    RVC_KP(11);
    __syncthreads();                                   // (carries the vmcnt(0) of the LDS-DMA pieces)
    // 3. one pass over the tile: fused input LeakyReLU (slope 1 = none) and the turn to channel-contiguous columns, so that a lane's four
    //    k-values of an MFMA B operand are ONE ds_read_b128 (read: lanes along the channels, odd row stride; write: consecutive floats)
    {
        const int total = cin * rl;
        const float inv_cin = 1.0f / (float)cin;
        for (int e = threadIdx.x; e < total; e += NT) {
            const int col = (int)(((float)e + 0.5f) * inv_cin), ci = e - col * cin;
            const float v = tmp[ci * rt + col];
            fin[col * cs + ci] = fmaxf(v, v * pre_slope);
        }
    }
    __syncthreads();
    RVC_KP(2);
    // 4. main loop over the chunks (tap-major): the B fragment of (tap t, channel group g), column n is fin[(n + t * dil) * cs + g * 16 + 4 kq .. + 3]
    const float *bl = fin + (wn * NF * 16 + li) * cs + kq * 4;
    const int last = nloc - 1;
    // (tap, group) of a chunk index, carried incrementally: requests run 1 (B) and DA (weights) of this wave's chunks ahead of the MFMAs
    int g_b = 0, off_b = 0;                             // chunk whose B fragments are requested next (floats from bl)
    const int wrap_b = dil * cs - 16 * G;               // from the last channel group of a tap to the first of the next
#define RVC_CT_ADV1() { g_b++; off_b += 16; if (g_b == G) { g_b = 0; off_b += wrap_b; } }
#define RVC_CT_ADV() { g_b += KS; off_b += 16 * KS; if (g_b >= G) { g_b -= G; off_b += wrap_b; } }      /* (KS <= G: checked at plan time) */
    if (kh) RVC_CT_ADV1()
    f32x4 bcur[NF];
#pragma unroll
    for (int nf = 0; nf < NF; nf++) bcur[nf] = *reinterpret_cast<const f32x4 *>(bl + off_b + nf * 16 * cs);
    RVC_CT_ADV()
    constexpr int T_ = MF * NF * 4;
#define RVC_CT_STEP(S, CC)                                                                              \
    {                                                                                                  \
        const int cc_ = (CC);                                                                          \
        const int ob_ = cc_ < last ? off_b : 0;                 /* the surplus request of the last chunk stays inside the tile */ \
        f32x4 bnx_[NF];                                                                                \
        _Pragma("unroll") for (int t = 0; t < T_; t++) {                                               \
            const int j = t / (NF * MF), nf = (t / MF) % NF, mf = t % MF;                              \
            if (!(RVC_CT_DBG & 8)) acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_st[S][mf][j], bcur[nf][j], acc[mf][nf], 0, 0, 0); \
            else acc[mf][nf][0] += a_st[S][mf][j] * bcur[nf][j];                                       \
            /* the next chunk's B fragments are requested between the MFMAs */                         \
            _Pragma("unroll") for (int r = t * NF / T_; r < (t + 1) * NF / T_; r++) {                  \
                if (!(RVC_CT_DBG & 2)) bnx_[r] = *reinterpret_cast<const f32x4 *>(bl + ob_ + r * 16 * cs); else bnx_[r] = bcur[r]; \
            }                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                              \
        /* the weights are reloaded BEHIND their last use (round 5): requested between the MFMAs into the array they were still being read from, the \
           new value lived in a second register set and the loop-carried copies back (v_mov on registers a load had just been issued into) sat at \
           the loop head -- every round of DA chunks began by waiting for all weight loads in flight */ \
        if (!(RVC_CT_DBG & 1)) {                                                                       \
            _Pragma("unroll") for (int mf = 0; mf < MF; mf++) a_st[S][mf] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(wnext[mf]) + lane16); \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        RVC_CT_ADV()                                                                                   \
        _Pragma("unroll") for (int mf = 0; mf < MF; mf++) wnext[mf] += KS * 256;      /* (requests past the wave's last chunk read the slack behind the panel) */ \
        _Pragma("unroll") for (int nf = 0; nf < NF; nf++) bcur[nf] = bnx_[nf];                         \
    }
    int c = 0;
    for (; c + DA <= nloc; c += DA) {
#pragma unroll
        for (int s = 0; s < DA; s++) RVC_CT_STEP(s, c + s)
    }
#pragma unroll
    for (int s = 0; s < DA; s++)
        if (c + s < nloc) RVC_CT_STEP(s, c + s)
#undef RVC_CT_STEP
#undef RVC_CT_ADV
#undef RVC_CT_ADV1
    RVC_KP(3);
    if (KS == 2) {
        // the odd-chunk waves hand their partial sums over through LDS (the arrival buffer is free since the turn pass)
        f32x4 *red = reinterpret_cast<f32x4 *>(tmp) + (w4 * MF * NF) * 64 + lane;
        if (kh) {
#pragma unroll
            for (int mf = 0; mf < MF; mf++)
#pragma unroll
                for (int nf = 0; nf < NF; nf++) red[(mf * NF + nf) * 64] = acc[mf][nf];
        }
        __syncthreads();
        if (kh) return;
#pragma unroll
        for (int mf = 0; mf < MF; mf++)
#pragma unroll
            for (int nf = 0; nf < NF; nf++) { const f32x4 o = red[(mf * NF + nf) * 64]; acc[mf][nf][0] += o[0]; acc[mf][nf][1] += o[1]; acc[mf][nf][2] += o[2]; acc[mf][nf][3] += o[3]; }
    }
    // 5. epilogue (the arithmetic of every kernel of the family: epi2_value)
    const float *resb = p.res ? p.res + (long long)b * p.res_bs : nullptr;
    float *yb = p.y + (long long)b * p.y_bs + ph.y_off;
    const int act_sel = ph.act_p1 ? ph.act_p1 - 1 : p.act;
    ColOut cols[NF];
#pragma unroll
    for (int nf = 0; nf < NF; nf++) cols[nf] = col_locate(p, ph, tn * BN + (wn * NF + nf) * 16 + li);
    RVC_ACT_DISPATCH_SEL(act_sel, conv_tile_store<A_ COMMA_ MF COMMA_ NF>(p, ph, resb, yb, cols, ((tm * WM + wm) * MF) * 16 + kq * 4, acc);)
    RVC_KP(6);
}
#undef COMMA_

}  // namespace rvc
