// conv32s.hip.h -- stride-1 1-D convolutions at MANY streams (the HiFiGAN decoder's ResBlock chains from 5 streams up: 32 .. 256 channels, kernel
// sizes 3 / 7 / 11, dilations 1 / 3 / 5) on v_mfma_f32_32x32x2_f32 with the activation operand staged ONCE per workgroup and channel block.
//
// Why (round 5, profiles/r05_layers_64streams.json): igemm32_kernel treats a convolution as a GEMM over K = Cin x taps and re-gathers its [16 k][BN]
// activation tile from global memory for EVERY 16-deep K step -- an input element is fetched once per tap (3 .. 11 times per workgroup), eight or
// sixteen dword gathers, the LeakyReLU, two ds_write_b128 and a barrier per step and thread.  With a 32-row weight panel that is one gather per MFMA,
// and an fp32 MFMA hides none of its SIMD's other instructions (tests/tools/mfma_overlap_probe.hip): the decoder's 32- / 64- / 128-channel layers ran
// at 57-85 / 84-106 / 104-118 TF/s.  Here a workgroup copies the input rows of a 32-channel block it needs -- [32][BN + (KW - 1) * dil] floats -- into LDS
// once (fused input LeakyReLU applied once per element, stored channel-contiguous per column), then walks the taps of that block from LDS: the B operand
// of (tap t, channel group g) is ONE ds_read_b128 per four MFMA k-steps at (column + t * dil) * CS + g * 16 from a per-lane base, no offset table, no
// barrier inside a block, and the main loop's only vector-memory traffic is the weight stream.  K is therefore walked (channel block, tap, channel
// group)-major; the weights are repacked to that order at plan time (plan.hip, queue_conv32s), inside a 16-deep chunk they keep the 16-row fragment
// packing of the whole family (lane (row r, k-slot s) reads the float4 of fragment r >> 4, quad 2u + s: MFMA (u, j) takes channel (2u + s) * 4 + j).
// The streams stay a grid dimension (a tile never straddles two streams: its staged columns are one contiguous range of one row set).
#pragma once
#include "igemm.hip.h"

namespace rvc {

// The epilogue of the family (one body for conv32s_kernel and conv32s_buf_kernel): igemm32_kernel's C / D layout -- col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) --, bias -> activation -> residual -> scale (-> accumulate); full tiles without accumulation take the
// straight-line path (operands in store-free batches, one predicate per column block), everything else the general one.
template <int WM, int WN, int MT, int NT>
__device__ __forceinline__ void c32s_epilogue(const IgemmP &p, const PhaseD &ph, f32x16 (&acc)[MT][NT], const int tm, const int tn, const int wm, const int wn, const int c32,
                                              const int ks, const int b)
{
    constexpr int BN = WN * NT * 32;
    const float *resb = p.res ? p.res + (long long)b * p.res_bs : nullptr;
    float *yb = p.y + (long long)b * p.y_bs;
    ColOut cols[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) cols[nt] = col_locate(p, ph, tn * BN + (wn * NT + nt) * 32 + c32);
    const int row0 = (tm * WM + wm) * MT * 32 + ks * 4;
    const bool full_m = row0 - ks * 4 + MT * 32 <= p.M;
    if (full_m) {
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
}

// (2 x 2 accumulator blocks per wave + the staging registers of the next block pass 170 registers: two waves per SIMD there, three for the 1 x 2 tiles
//  -- the only ones instantiated: conv32s_inst.hip)
template <int MT, int NT> struct C32sOcc { static constexpr int W = MT * NT >= 4 ? 2 : 3; };
template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C32sOcc<MT, NT>::W, C32sOcc<MT, NT>::W))) void conv32s_kernel(IgemmP p)
{
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BN = WN * NT * 32;
    constexpr int CB = 32, GB = CB / 16;            // channels staged per block, 16-deep chunks per (block, tap)
    constexpr int CS = CB + 4;                      // LDS column stride in floats: 16-byte aligned, 16 lanes x 16 bytes on disjoint banks
    static_assert(GB == 2, "the tap body below is written for two chunks");
    extern __shared__ __attribute__((aligned(16))) float s_x[];      // [BN + (KW - 1) * dil][CS]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave / WN, wn = wave % WN;
    const int c32 = lane & 31, ks = lane >> 5;
    // grid = (n-tiles x m-tiles, streams, phases): no division for the common one-m-tile layers (the four runtime divisions of a flat index were ~160 of the
    // prologue's ~900 instructions, and a 32-row layer's wave has only 96-352 MFMAs to set them against); long phases first (the planner sorts them)
    const int tm = p.ntm == 1 ? 0 : (int)blockIdx.x / p.ntn, tn = p.ntm == 1 ? (int)blockIdx.x : (int)blockIdx.x - tm * p.ntn;
    const int phase = (int)blockIdx.z, b = (int)blockIdx.y;
    const PhaseD ph = p.nphase == 1 ? p.ph0 : p.ph[phase];
    const int nchunks = ph.nchunks;
    const int kw = ph.t_tab & 0xff, dil = ph.t_tab >> 8;
    const int nblk = ph.t_cin / CB;
    const int ncol = BN + (kw - 1) * dil;
    // weights: [m_tile16][chunk][lane16x4][4], chunk = (block * KW + tap) * GB + group
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
    f32x4 a_ev[MT][2], a_od[MT][2];          // weights of the even / odd chunk of a tap
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int u = 0; u < 2; u++) a_ev[mt][u] = *reinterpret_cast<const f32x4 *>(wrow[mt] + u * 128);
    const int dbg = p.pad2_;                 // tuning aid (RVC_C32S_DBG): 1 = no staging, 2 = no weight reloads, 4 = no B reads (timing only)
    const float pre_slope = p.pre_slope;
    const float *xb = p.x + (long long)b * p.x_bs + ph.x_off;
    const int n0 = tn * BN + ph.t_dmin;
    const float *bl = s_x + (wn * NT * 32 + c32) * CS + ks * 4;      // B operand base of this lane
    // one chunk: request the next chunk's weights, this chunk's B operands (two ds_read_b128 per 32-column block), 16 MT NT MFMAs
    auto kstep = [&](const int c, const float *bq, f32x4 (&a_c)[MT][2], f32x4 (&a_n)[MT][2]) {
        const int cn = c + 1 < nchunks ? c + 1 : c;
        if (!(dbg & 2)) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int u = 0; u < 2; u++) a_n[mt][u] = *reinterpret_cast<const f32x4 *>(wrow[mt] + (long long)cn * 256 + u * 128);
        } else {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int u = 0; u < 2; u++) a_n[mt][u] = a_c[mt][u];
        }
        f32x4 bv[2][NT];
        if (!(dbg & 4)) {
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) bv[u][nt] = *reinterpret_cast<const f32x4 *>(bq + nt * 32 * CS + u * 8);
        } else {
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) bv[u][nt] = a_c[0][u];
        }
        __builtin_amdgcn_sched_barrier(0);      // (the requests stay in FRONT of the chunk's MFMAs: left to itself the scheduler sinks the weight loads behind most of them)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[mt][u][j], bv[u][nt][j], acc[mt][nt], 0, 0, 0);
    };
    // Staging is software-pipelined: the rows of block blk + 1 are requested (global -> registers) when the taps of block blk begin and written to LDS when
    // they are done, so the only exposed part of a block change is two barriers and the ds_write_b128s.  Item r of a thread: channel quad q, staged column s
    // of a [CB / 4][NCP] item grid (NCP = BN + 64 columns, a multiple of 64: q is wave-uniform, a wave reads 64 consecutive columns of four rows).
    constexpr int NCP = BN + 64, NI = (CB / 4) * NCP / 256;
    static_assert((CB / 4) * NCP % 256 == 0, "item grid must divide over the workgroup");
    f32x4 pf[NI];
    auto request = [&](const int blk) {
        const float *xk = xb + (long long)blk * CB * p.x_ld;
#pragma unroll
        for (int r = 0; r < NI; r++) {
            const int it0 = r * 256 + wave * 64;
            const int q = it0 / NCP;
            int s = it0 % NCP + lane;
            s = s < ncol ? s : ncol - 1;                 // (columns past the tile's reach re-read its last one and are not stored)
            int gc = n0 + s;
            gc = gc < p.x_lo ? p.x_lo : (gc > p.x_lim ? p.x_lim : gc);
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const float *rowp = xk + (long long)(q * 4 + jj) * p.x_ld;       // wave-uniform row base + per-lane column
                pf[r][jj] = rowp[gc];
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int r = 0; r < NI; r++) {
            const int it0 = r * 256 + wave * 64;
            const int q = it0 / NCP, s = it0 % NCP + lane;
            f32x4 v;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) v[jj] = fmaxf(pf[r][jj], pf[r][jj] * pre_slope);
            if (s < ncol) *reinterpret_cast<f32x4 *>(s_x + s * CS + q * 4) = v;
        }
    };
    if (!(dbg & 1)) request(0);
    int c = 0;
    for (int blk = 0; blk < nblk; blk++) {
        if (blk) __syncthreads();                  // every wave has left the previous block's tile
        if (!(dbg & 1)) commit();
        __syncthreads();
        if (blk + 1 < nblk && !(dbg & 1)) request(blk + 1);
        const float *bt = bl;
        for (int t = 0; t < kw; t++) {
            kstep(c, bt, a_ev, a_od);
            kstep(c + 1, bt + 16, a_od, a_ev);
            c += 2;
            bt += dil * CS;
        }
    }
    c32s_epilogue<WM, WN, MT, NT>(p, ph, acc, tm, tn, wm, wn, c32, ks, b);
}

// ------------------------------------------------------------------------------------------------------------------------
// conv32s_buf_kernel -- the same kernel with every vector-memory instruction a buffer load (per-lane byte offset that never changes, the chunk's / row's offset
// in an SGPR: no vector ALU work per load) and the phase descriptor's fields forced scalar (readfirstlane: scalar loop control instead of exec-masked loops).
// Measured per tile, same box, 16 / 32 / 64 streams, us per six launches: the 128 x 64 tile 882 / 1 459 / 2 889 -> 806 / 1 394 / 2 776 on the 128-row layers and
// 1 175 -> 1 077 on the 256-row stage (129 TF/s); the 32 x 256 and 64 x 128 tiles get SLOWER with it (825 -> 897, 775 -> 842), and slower still with scalar loop
// control and global loads (976, 1 725) -- so those two tiles stay on conv32s_kernel above exactly as it was measured, and only the 128 x 64 tile is instantiated
// from this one.  Codegen, not design: re-measure when the compiler changes.
template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C32sOcc<MT, NT>::W, C32sOcc<MT, NT>::W))) void conv32s_buf_kernel(IgemmP p)
{
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BN = WN * NT * 32;
    constexpr int CB = 32, GB = CB / 16;            // channels staged per block, 16-deep chunks per (block, tap)
    constexpr int CS = CB + 4;                      // LDS column stride in floats: 16-byte aligned, 16 lanes x 16 bytes on disjoint banks
    static_assert(GB == 2, "the tap body below is written for two chunks");
    extern __shared__ __attribute__((aligned(16))) float s_x[];      // [BN + (KW - 1) * dil][CS]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave / WN, wn = wave % WN;
    const int c32 = lane & 31, ks = lane >> 5;
    // grid = (n-tiles x m-tiles, streams, phases): no division for the common one-m-tile layers (the four runtime divisions of a flat index were ~160 of the
    // prologue's ~900 instructions, and a 32-row layer's wave has only 96-352 MFMAs to set them against); long phases first (the planner sorts them)
    const int tm = p.ntm == 1 ? 0 : (int)blockIdx.x / p.ntn, tn = p.ntm == 1 ? (int)blockIdx.x : (int)blockIdx.x - tm * p.ntn;
    const int phase = (int)blockIdx.z, b = (int)blockIdx.y;
    const PhaseD ph = p.nphase == 1 ? p.ph0 : p.ph[phase];
    // (the phase descriptor comes through a select between the kernel argument and a global load: the compiler keeps its fields in vector registers and treats
    //  every loop bound derived from them as divergent -- exec-masked loops, and a waterfall loop around every buffer load whose scalar offset depends on them)
    auto uni = [](const int v) { return __builtin_amdgcn_readfirstlane(v); };
    const int nchunks = uni(ph.nchunks);
    const int t_tab = uni(ph.t_tab);
    const int kw = t_tab & 0xff, dil = t_tab >> 8;
    const int nblk = uni(ph.t_cin) / CB;
    const int ncol = BN + (kw - 1) * dil;
    // weights: [m_tile16][chunk][lane16x4][4], chunk = (block * KW + tap) * GB + group
    // (all vector-memory instructions are buffer loads: a per-lane byte offset that never changes, the chunk's / row's offset in an SGPR -- no vector ALU
    //  work per load, and one kind of load keeps the compiler's s_waitcnt counts exact: igemm2w_kernel)
    const long long w_off = ((long long)uni((int)(ph.w_off >> 32)) << 32) | (unsigned)uni((int)ph.w_off);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w + w_off), 0, 0x7ffff000, 0x00020000);
    int wo[MT];
    const int mtiles = (p.M + 15) >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int t16 = ((tm * WM + wm) * MT + mt) * 2 + (c32 >> 4);
        t16 = t16 < mtiles ? t16 : mtiles - 1;
        wo[mt] = (t16 * nchunks * 256 + (ks * 16 + (c32 & 15)) * 4) * 4;
    }
    auto wload = [&](const int mt, const int c, const int u) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wo[mt], c * 1024 + u * 512, 0));
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;
    f32x4 a_ev[MT][2], a_od[MT][2];          // weights of the even / odd chunk of a tap
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int u = 0; u < 2; u++) a_ev[mt][u] = wload(mt, 0, u);
    const float pre_slope = p.pre_slope;
    // input rows: base moved to the first readable column of a row (the halo), so that per-lane offsets are non-negative
    const float *xb = p.x + (long long)b * p.x_bs + uni(ph.x_off);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xb + p.x_lo), 0, 0x7ffff000, 0x00020000);
    const int n0 = tn * BN + uni(ph.t_dmin);
    const float *bl = s_x + (wn * NT * 32 + c32) * CS + ks * 4;      // B operand base of this lane
    // one chunk: request the next chunk's weights, this chunk's B operands (two ds_read_b128 per 32-column block), 16 MT NT MFMAs
    auto kstep = [&](const int c, const float *bq, f32x4 (&a_c)[MT][2], f32x4 (&a_n)[MT][2]) {
        const int cn = c + 1 < nchunks ? c + 1 : c;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int u = 0; u < 2; u++) a_n[mt][u] = wload(mt, cn, u);
        f32x4 bv[2][NT];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) bv[u][nt] = *reinterpret_cast<const f32x4 *>(bq + nt * 32 * CS + u * 8);
        __builtin_amdgcn_sched_barrier(0);      // (the requests stay in FRONT of the chunk's MFMAs: left to itself the scheduler sinks the weight loads behind most of them)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int nt = 0; nt < NT; nt++)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c[mt][u][j], bv[u][nt][j], acc[mt][nt], 0, 0, 0);
    };
    // Staging is software-pipelined: the rows of block blk + 1 are requested (global -> registers) when the taps of block blk begin and written to LDS when
    // they are done, so the only exposed part of a block change is two barriers and the ds_write_b128s.  Item r of a thread: channel quad q, staged column s
    // of a [CB / 4][NCP] item grid (NCP = BN + 64 columns, a multiple of 64: q is wave-uniform, a wave reads 64 consecutive columns of four rows).
    constexpr int NCP = BN + 64, NI = (CB / 4) * NCP / 256;
    static_assert((CB / 4) * NCP % 256 == 0, "item grid must divide over the workgroup");
    f32x4 pf[NI];
    auto request = [&](const int blk) {
#pragma unroll
        for (int r = 0; r < NI; r++) {
            const int it0 = r * 256 + wave * 64;
            const int q = it0 / NCP;
            int s = it0 % NCP + lane;
            s = s < ncol ? s : ncol - 1;                 // (columns past the tile's reach re-read its last one and are not stored)
            int gc = n0 + s;
            gc = gc < p.x_lo ? p.x_lo : (gc > p.x_lim ? p.x_lim : gc);
            const int vo = (gc - p.x_lo) * 4;
#pragma unroll
            for (int jj = 0; jj < 4; jj++)      // wave-uniform row offset (SGPR) + per-lane column
                pf[r][jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, vo, (blk * CB + q * 4 + jj) * p.x_ld * 4, 0));
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int r = 0; r < NI; r++) {
            const int it0 = r * 256 + wave * 64;
            const int q = it0 / NCP, s = it0 % NCP + lane;
            f32x4 v;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) v[jj] = fmaxf(pf[r][jj], pf[r][jj] * pre_slope);
            if (s < ncol) *reinterpret_cast<f32x4 *>(s_x + s * CS + q * 4) = v;
        }
    };
    request(0);
    int c = 0;
    for (int blk = 0; blk < nblk; blk++) {
        if (blk) __syncthreads();                  // every wave has left the previous block's tile
        commit();
        __syncthreads();
        if (blk + 1 < nblk) request(blk + 1);
        const float *bt = bl;
        for (int t = 0; t < kw; t++) {
            kstep(c, bt, a_ev, a_od);
            kstep(c + 1, bt + 16, a_od, a_ev);
            c += 2;
            bt += dil * CS;
        }
    }
    c32s_epilogue<WM, WN, MT, NT>(p, ph, acc, tm, tn, wm, wn, c32, ks, b);
}

}  // namespace rvc
