// igemm32l.hip.h -- igemm32_kernel for the TABLE-FREE 1x1 layers (every Linear of ContentVec at many streams: 48 launches, 11.5 ms of a 64-stream step) with
// every vector-memory instruction a buffer load and scalar loop control (round 5, after conv32s_buf_kernel: finding 6's codegen notes).
// igemm32_kernel addresses operand row k of a 1x1 layer through the layer's offset table like any convolution: a ds_read of the table entry, a 64-bit add
// and a global load per gathered element -- eight per thread and K step next to 32 MFMAs --, and its weight stream pays a 64-bit add per load; its loop bounds
// come from a phase descriptor the compiler keeps in vector registers (exec-masked loops).  Here row k sits at k * channel stride: the per-lane byte offset of
// a gather (stream, column) never changes and the row's offset is an SGPR, the weights the same; nphase == 1, so the descriptor is the kernel argument.
// Anatomy, tile shapes, LDS layout ([2][BN][16 k + 4]), MFMA operand order and epilogue are igemm32_kernel's.
#pragma once
#include "igemm.hip.h"

namespace rvc {

// TAB: layers WITH an offset table (one phase: ContentVec's strided stem convolutions, three-tap decoder layers that stay off the staged kernel).  A staging
// thread's k rows are wave-uniform (kr0), so the table entries of a K step are SCALAR loads (s_load from the table in global memory, no copy in LDS) and go
// straight into the buffer loads' scalar offset: the same zero-VALU K step.  PRE: fused input LeakyReLU at staging, as in igemm32_kernel.
template <int WM, int WN, int MT, int NT, bool TAB = false, bool PRE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(G32Occ<MT, NT>::W, G32Occ<MT, NT>::W))) void igemm32l_kernel(IgemmP p)
{
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int BN = WN * NT * 32;
    constexpr int RSK = 20;
    constexpr int KR = 256 / BN > 0 ? 256 / BN : 1;
    constexpr int EPT = 16 / KR;
    static_assert(BN <= 256 && 256 % BN == 0, "BN must divide 256");
    extern __shared__ __attribute__((aligned(16))) float s_bt[];       // [2][BN][RSK]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave / WN, wn = wave % WN;
    int tn, tm;
    if (p.m_fast == 3) {
        // Panel order (round 6; tall table-free panels with streams folded into N -- ContentVec's 2304- / 3072-row projections): the launch's n-tiles are dealt
        // to the 8 XCDs as contiguous ranges, and INSIDE an XCD the tiles run panel by panel: mp m-tiles (a weight panel that fits the XCD's L2 next to the
        // activation tiles) x all of the XCD's n-tiles, n-tile major, before the next panel.  With m fastest over all m-tiles (m_fast = 1) every n-tile streamed
        // the WHOLE weight matrix (7-9 MB > the 4 MB L2) through the cache: 480 MB of fabric reads per 2304 x 768 launch for 29 MB of operands.  The grid is
        // padded to 8 x (most tiles any XCD owns); surplus workgroups leave at once.  Block x runs on XCD x % 8 (dispatch order, tests/tools/xcd_probe.hip).
        const int c = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
        const int nb = p.ntn >> 3, nr = p.ntn & 7;
        const int nx = nb + (c < nr ? 1 : 0), n0 = c * nb + (c < nr ? c : nr);
        const int mp = p.pad2_, np = p.ntm / mp, mr = p.ntm - np * mp;
        if (i >= nx * p.ntm) return;
        const int full = np * mp * nx;
        if (i < full) { const int g = i / (mp * nx), r = i - g * mp * nx; tn = n0 + r / mp; tm = g * mp + r % mp; }
        else { const int r = i - full; tn = n0 + r / mr; tm = np * mp + r % mr; }
    } else {
        const int tid_x = p.m_fast == 1 ? xcd_tile_id((int)blockIdx.x, (int)gridDim.x, (int)(blockIdx.y * gridDim.x)) : (int)blockIdx.x;
        tn = p.m_fast ? tid_x / p.ntm : tid_x % p.ntn; tm = p.m_fast ? tid_x % p.ntm : tid_x / p.ntn;
    }
    const int b = blockIdx.y;
    const PhaseD &ph = p.ph0;
    const int nchunks = ph.nchunks;
    const int c32 = lane & 31, ks = lane >> 5;
    // activations: staging role = column n_s of the tile, k rows kr0 .. kr0 + EPT - 1 (kr0 is wave-uniform: BN is a multiple of 64)
    // (table entries are non-negative byte offsets from a base moved back by koff_bias)
    const char *xbase = reinterpret_cast<const char *>(p.x + (long long)b * p.x_bs + ph.x_off) - (TAB ? p.koff_bias : 0);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(xbase), 0, 0x7ffff000, 0x00020000);
    const int *kt = p.koff + ph.koff_off;
    const int n_s = threadIdx.x % BN;
    const int kr0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / BN) * EPT);
    int xo_s;
    {
        int n = tn * BN + n_s;
        n = n < p.N ? n : p.N - 1;
        int bb = 0;
        if (p.fold_n) { bb = n / p.fold_n; n -= bb * p.fold_n; }
        xo_s = (bb * (int)p.x_bs + n * p.x_ws) * 4;
    }
    const int lin = p.lin_cs4;
    const float pre_slope = p.pre_slope;
    // weights: 16-row fragment packing [m_tile16][chunk][lane16x4][4]
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w + ph.w_off), 0, 0x7ffff000, 0x00020000);
    int wo[MT];
    const int mtiles = (p.M + 15) >> 4;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int t16 = ((tm * WM + wm) * MT + mt) * 2 + (c32 >> 4);
        t16 = t16 < mtiles ? t16 : mtiles - 1;
        wo[mt] = (t16 * nchunks * 256 + (ks * 16 + (c32 & 15)) * 4) * 4;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][nt][r] = 0.f;
    float sb[EPT];
    f32x4 a_ev[MT][2], a_od[MT][2];
    // TAB: the table entries of a step are loaded (scalar) one step before the gathers that use them
    int so_t[EPT];
    auto table = [&](const int c) {
        if (TAB) {
#pragma unroll
            for (int i = 0; i < EPT; i++) so_t[i] = __builtin_amdgcn_readfirstlane(kt[c * 16 + kr0 + i]);
        }
    };
    auto gather = [&](const int c) {
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            const int so = TAB ? so_t[i] : (c * 16 + kr0 + i) * lin;
            sb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xo_s, so, 0));
        }
    };
    auto wload = [&](f32x4 (&a)[MT][2], const int c) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int u = 0; u < 2; u++) a[mt][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, wo[mt], c * 1024 + u * 512, 0));
    };
    auto stage = [&](float *bnxt) {
#pragma unroll
        for (int i = 0; i < EPT; i += 4) {
            f32x4 v4;
#pragma unroll
            for (int q = 0; q < 4; q++) v4[q] = PRE ? fmaxf(sb[i + q], sb[i + q] * pre_slope) : sb[i + q];
            *reinterpret_cast<f32x4 *>(bnxt + n_s * RSK + kr0 + i) = v4;
        }
    };
    table(0);
    gather(0);
    table(nchunks > 1 ? 1 : 0);
    wload(a_ev, 0);
    stage(s_bt);
    __syncthreads();
    const float *br = s_bt + (wn * NT * 32 + c32) * RSK + ks * 4;
    auto kstep = [&](const int c, f32x4 (&a_c)[MT][2], f32x4 (&a_n)[MT][2]) {
        const int cn = c + 1 < nchunks ? c + 1 : c;
        const float *bcur = br + (c & 1) * BN * RSK;
        float *bnxt = s_bt + ((c + 1) & 1) * BN * RSK;
        gather(cn);
        wload(a_n, cn);
        __builtin_amdgcn_sched_barrier(0);
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
        __builtin_amdgcn_sched_barrier(0);      // (the next tile is written BEHIND the step's MFMAs: hoisted in front of them, its ds_write waits for the gathers the step has just issued)
        table(cn + 1 < nchunks ? cn + 1 : cn);  // (scalar loads share the LDS counter: requested here, they are covered by the wait in front of the barrier)
        stage(bnxt);
        __syncthreads();
    };
    {
        int c = 0;
        for (; c + 2 <= nchunks; c += 2) { kstep(c, a_ev, a_od); kstep(c + 1, a_od, a_ev); }
        if (c < nchunks) kstep(c, a_ev, a_od);
    }
    // epilogue: igemm32_kernel's
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

}  // namespace rvc
