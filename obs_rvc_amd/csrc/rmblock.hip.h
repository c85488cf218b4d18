// rmblock.hip.h -- rm_block_kernel: one ConvBlockRes of RMVPE's SHALLOW levels (<= 64 output channels: encoder levels 0-2, decoder levels 2-0) in ONE
// launch (round 6).  Reference: the rmvpe.onnx run of rvc/src/f0/rmvpe.rs:235-238; block definition SURVEY.md appendix A.2:
//     y1 = ReLU(conv3x3(x) + b1)     (BatchNorm folded)            out = ReLU(conv3x3(y1) + b2) + (1x1 shortcut(x) + bsc | x)
//
// Why: at one stream the f0 branch is a chain of ~125 dependent launches of 5-8 us each on its 32-CU partition (profiles/r06_layers_1streams.json: every
// RMVPE layer runs below 4 TF/s -- the time is the launch, not the arithmetic) and it is the critical path of the chunk's front (timeline: RMVPE + GRU +
// decode end at ~1.09 ms, ContentVec at ~1.04).  A block's two 3x3 convolutions are dependent launches because the second needs a one-pixel halo of the
// first's output.  On the shallow levels the image is large and the channel count small, so a workgroup can own a spatial tile and RECOMPUTE that halo:
// input tile (TH + 4) x (TW + 4) -> y1 tile (TH + 2) x (TW + 2), kept in LDS -> output tile TH x TW.  24 of the 120 RMVPE launches disappear.  The deep
// levels (128+ channels on <= 64 pixels) are weight streaming over all output channels: fusing them needs a grid-wide hand-off per layer, which round 4
// measured at break-even (probe 2); they keep their launches.
//
// Arithmetic: v_mfma_f32_16x16x4_f32.  GEMM view of a convolution: M = Cout, N = pixels of the tile, K = 9 taps x Cin walked tap-major in steps of four
// channels.  A (weights) comes from a per-block panel packed at model load in fragment order [tap][Cin/4][Cout/16][64 lanes] (lane (row li, k-slot kq) holds
// W[16 mt + li][4 c4 + kq][tap]): the K walk is one linear, coalesced stream, prefetched twelve steps ahead through a four-slot register ring.  B (activations)
// is a ds_read_b32 from the staged tile: lane (column li, k-slot kq) reads channel 4 c4 + kq at pixel(li) + tap offset.  The four waves split M first
// (Cout / 16 panels: the weight stream is then read once per workgroup) and N second.
#pragma once
#include "igemm.hip.h"

namespace rvc {

struct RmBlockP {
    const float *x; float *y;
    int Cin, Cin4, Cout;              // Cin4 = (Cin padded to 16) / 4: K steps per tap of the first convolution and of the shortcut
    int H, W, TH, TW, tiles_x, tiles;      // tiles: workgroups that compute (one more, of stream 0, only warms the next block's panels)
    int x_ld, x_cs; long long x_bs;
    int y_ld, y_cs; long long y_bs;
    const float *w1, *w2, *wsc;       // fragment-order panels (wsc == nullptr: identity shortcut, Cin == Cout)
    const float *b1, *b2, *bsc;
    int XS, YS;                       // LDS channel strides of the input tile and of the y1 tile (floats, 16 mod 32)
    float *ypool; int p_ld, p_cs; long long p_bs;      // second output (or nullptr): AvgPool2d(2, 2) of y -- tiles of 8 columns only
    int pool;                         // 1: x is the previous level's output at twice the resolution (H, W are this block's): the staging averages 2 x 2 pixels
    int wlines;                       // 128-byte lines of the [w1 | w2 | wsc] allocation (<= 1536)
    const float *wnext; int wnext_lines;      // the NEXT fused block's panels (or nullptr): requested into the L2 while this block computes
};


// one convolution phase: acc[i] += W . tile for this wave's m panel and n-tiles; `src` = LDS tile with channel stride CS, row width RW; poff[i] = k-slot row + pixel
// offset of this lane's column in n-tile i; steps = K steps per tap (a multiple of 4).  The weight stream runs three batches of four steps ahead through a four-slot
// register ring.  Steady state = whole groups of four batches with NOTHING conditional in them: the compiler derives its s_waitcnt counts from what is outstanding
// on every path into a block, and a loop body with a guard around the requests waits with vmcnt(0) -- one L2 round trip per batch (the first version of this
// kernel: 33-82 us per block instead of 6-9).  The last nb mod 4 batches run in a guarded tail.  The number of n-tiles per wave is a template parameter for the
// same kind of reason: a guard per MFMA (`if (i < ntw)`) put every ds_read directly in front of its MFMA behind an lgkmcnt(0) -- one LDS latency per MFMA.
template <int MT, int NT>
__device__ __forceinline__ void rm_conv_phase(f32x4 (&acc)[NT], const float *wa, const float *src, const int CS, const int RW, const int steps, const int ntaps_side,
                                              const int (&poff)[NT])
{
    const int ksteps = ntaps_side * ntaps_side * steps;          // multiple of 4
    const int nb = ksteps >> 2, ngroups = nb >> 2, rem = nb & 3;
    float a[4][4];
    auto ldA = [&](float (&dst)[4], const int s0) {
#pragma unroll
        for (int u = 0; u < 4; u++) { int s = s0 + u; s = s < ksteps ? s : ksteps - 1; dst[u] = wa[(long long)s * (MT * 64)]; }
    };
    int c4 = 0, tx = 0, ty = 0, tapoff = 0;
    auto batch = [&](const float (&av)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float *sp = src + (c4 + u) * 4 * CS + tapoff;
#pragma unroll
            for (int i = 0; i < NT; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], sp[poff[i]], acc[i], 0, 0, 0);
        }
        c4 += 4;
        if (c4 == steps) { c4 = 0; if (++tx == ntaps_side) { tx = 0; ty++; } tapoff = ty * RW + tx; }
    };
    ldA(a[0], 0); ldA(a[1], 4); ldA(a[2], 8);
    int sb = 0;
    for (int g = 0; g < ngroups; g++) {
#pragma unroll
        for (int bi = 0; bi < 4; bi++) {
            ldA(a[(bi + 3) & 3], sb + 12);                      // the slot consumed one batch ago (a request past the end re-reads the last step)
            batch(a[bi]);
            sb += 4;
        }
    }
    if (rem > 0) batch(a[0]);
    if (rem > 1) batch(a[1]);
    if (rem > 2) batch(a[2]);
}

// NT1 / NT2: n-tiles of 16 pixels per wave in the first / second convolution (a wave whose share of the tile is shorter computes clamped columns and drops them)
template <int MT, int NT1, int NT2>
__global__ __launch_bounds__(256) void rm_block_kernel(RmBlockP p)
{
    static_assert(MT == 1 || MT == 2 || MT == 4, "16 / 32 / 64 output channels");
    constexpr int NWN = 4 / MT;                                  // waves that share an m panel and split the n-tiles
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, li = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wm = wave % MT, wn = wave / MT;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty0 = (tile / p.tiles_x) * p.TH, tx0 = (tile % p.tiles_x) * p.TW;
    const int XW = p.TW + 4, XH = p.TH + 4, YW = p.TW + 2, YH = p.TH + 2;
    const int XS = p.XS, YS = p.YS;
    float *xs = smem;                                            // [Cin4 * 4][XS]  input tile, two-pixel halo, zero outside the image
    float *ys = smem + (size_t)p.Cin4 * 4 * XS;                  // [Cout][YS]      y1 tile, one-pixel halo, zero outside the image
    // --- warm the weight panels: they were last read one chunk (~850 MB of weight traffic) ago, i.e. they are in HBM.  The K walks below keep twelve steps in
    //     flight per wave: enough against an L2 hit, not against HBM (the first version of this kernel spent 3 + 3 memory round trips per block on a 9 KB panel).
    //     Every workgroup of the launch requests every line once, up front, next to the input tile's loads: one round trip for the whole block -- and an L2 hit
    //     when the previous fused block's extra workgroup has requested them already (they only have to reach the L2 of this XCD while that block computes).
    float warm = 0.f;
    if ((int)blockIdx.x >= p.tiles) {
        // the launch's extra workgroup: requests the NEXT block's panels and ends (loads return in order: inside a computing workgroup these HBM reads would sit
        // in front of every later wait of its weight stream)
        const char *wn_ = reinterpret_cast<const char *>(p.wnext);
#pragma unroll
        for (int u = 0; u < 6; u++) { int l = (int)threadIdx.x + u * 256; l = l < p.wnext_lines ? l : p.wnext_lines - 1; warm += *reinterpret_cast<const float *>(wn_ + (size_t)l * 128); }
        asm volatile("" :: "v"(warm));
        return;
    }
    {
        const char *wl = reinterpret_cast<const char *>(p.w1);
#pragma unroll
        for (int u = 0; u < 6; u++) { int l = (int)threadIdx.x + u * 256; l = l < p.wlines ? l : p.wlines - 1; warm += *reinterpret_cast<const float *>(wl + (size_t)l * 128); }
    }
    // --- stage the input tile: a thread owns one tile position r for every G-th channel (two integer divisions per thread, none per element); all loads of a
    //     batch are requested before the first LDS write (one memory round trip per batch of sixteen)
    {
        const float *xg = p.x + (long long)b * p.x_bs;
        const int plane = XH * XW, G = 256 / plane, Ctot = p.Cin4 * 4;          // (plane <= 256: checked by the planner)
        const int g = (int)threadIdx.x / plane, r = (int)threadIdx.x - g * plane;
        const int iy = r / XW, ix = r - iy * XW, gy = ty0 - 2 + iy, gx = tx0 - 2 + ix;
        const bool active = g < G, inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const float *xp = p.pool ? xg + 2 * gy * p.x_ld + 2 * gx : xg + gy * p.x_ld + gx;
        for (int c0 = 0; c0 < Ctot; c0 += 16 * G) {
            float v[16];
            if (p.pool) {          // AvgPool2d(2, 2) of the previous level's output, taken here
                float t[16][4];
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int c = c0 + u * G + g;
                    const bool ok = active && inside && c < p.Cin;
                    const float *q4 = xp + (long long)c * p.x_cs;
                    t[u][0] = ok ? q4[0] : 0.f; t[u][1] = ok ? q4[1] : 0.f; t[u][2] = ok ? q4[p.x_ld] : 0.f; t[u][3] = ok ? q4[p.x_ld + 1] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = (t[u][0] + t[u][1] + t[u][2] + t[u][3]) * 0.25f;          // (avgpool2_kernel's order)
            } else {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int c = c0 + u * G + g;
                v[u] = (active && inside && c < p.Cin) ? xp[(long long)c * p.x_cs] : 0.f;
            }
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int c = c0 + u * G + g;
                if (active && c < Ctot) xs[c * XS + r] = v[u];
            }
        }
    }
    asm volatile("" :: "v"(warm));          // (the warming loads are real: their values are waited for here, together with the input tile's)
    // --- first convolution: y1 on the (TH + 2) x (TW + 2) tile
    const int N1 = YH * YW, nt1 = (N1 + 15) >> 4;
    int poff[NT1], py1[NT1], px1[NT1];
#pragma unroll
    for (int i = 0; i < NT1; i++) {
        const int nt = wn + i * NWN;
        int n = nt * 16 + li; n = n < N1 ? n : N1 - 1;
        py1[i] = n / YW; px1[i] = n - py1[i] * YW;
        poff[i] = kq * XS + py1[i] * XW + px1[i];
    }
    f32x4 acc[NT1];
#pragma unroll
    for (int i = 0; i < NT1; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bias1[4];
#pragma unroll
    for (int r = 0; r < 4; r++) bias1[r] = p.b1[wm * 16 + kq * 4 + r];
    __syncthreads();
    rm_conv_phase<MT, NT1>(acc, p.w1 + wm * 64 + lane, xs, XS, XW, p.Cin4, 3, poff);
    // D layout: col = lane & 15 (pixel), row = (lane >> 4) * 4 + reg (channel)
#pragma unroll
    for (int i = 0; i < NT1; i++) {
        {
            const int n = (wn + i * NWN) * 16 + li;
            if (n < N1) {
                const int gy = ty0 - 1 + py1[i], gx = tx0 - 1 + px1[i];
                const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;      // the second convolution's zero padding applies to y1 itself
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v = acc[i][r] + bias1[r];
                    ys[(wm * 16 + kq * 4 + r) * YS + n] = inside ? (v > 0.f ? v : 0.f) : 0.f;
                }
            }
        }
    }
    // --- second convolution + shortcut on the TH x TW tile
    const int N2 = p.TH * p.TW, nt2 = (N2 + 15) >> 4;
    int qoff[NT2], coff[NT2], oy2[NT2], ox2[NT2];
#pragma unroll
    for (int i = 0; i < NT2; i++) {
        const int nt = wn + i * NWN;
        int n = nt * 16 + li; n = n < N2 ? n : N2 - 1;
        oy2[i] = n / p.TW; ox2[i] = n - oy2[i] * p.TW;
        qoff[i] = kq * YS + oy2[i] * YW + ox2[i];
        coff[i] = kq * XS + (oy2[i] + 2) * XW + ox2[i] + 2;    // the pixel itself in the input tile (k-slot row kq)
    }
    f32x4 acc2[NT2], accs[NT2];
#pragma unroll
    for (int i = 0; i < NT2; i++) { acc2[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; accs[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float bias2[4], biass[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { bias2[r] = p.b2[wm * 16 + kq * 4 + r]; biass[r] = p.wsc ? p.bsc[wm * 16 + kq * 4 + r] : 0.f; }
    if (p.wsc) rm_conv_phase<MT, NT2>(accs, p.wsc + wm * 64 + lane, xs, XS, XW, p.Cin4, 1, coff);       // (reads the input tile only: in front of the barrier)
    __syncthreads();
    rm_conv_phase<MT, NT2>(acc2, p.w2 + wm * 64 + lane, ys, YS, YW, p.Cout >> 2, 3, qoff);
    float *yg = p.y + (long long)b * p.y_bs;
#pragma unroll
    for (int i = 0; i < NT2; i++) {
        {
            const int n = (wn + i * NWN) * 16 + li;
            const int gy = ty0 + oy2[i], gx = tx0 + ox2[i];
            const bool ok = n < N2 && gy < p.H && gx < p.W;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = wm * 16 + kq * 4 + r;
                float t = acc2[i][r] + bias2[r];
                t = t > 0.f ? t : 0.f;
                t += p.wsc ? accs[i][r] + biass[r] : xs[(m - kq) * XS + coff[i]];
                v[r] = t;
                if (ok) yg[(long long)m * p.y_cs + gy * p.y_ld + gx] = t;
            }
            if (p.ypool) {
                // AvgPool2d(2, 2) of this block's output as a second result (the encoder level's last block: the pooling launch behind it disappears).  Tiles of
                // eight columns: an n-tile of 16 pixels is two tile rows, lanes li and li + 8 are vertical neighbours, li and li ^ 1 horizontal ones -- a 2 x 2 block
                // is summed with two shuffles inside the 16-lane group (H, W and the tile origin are even: a block is never cut by the image edge)
                float *pg = p.ypool + (long long)b * p.p_bs;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float t = ok ? v[r] : 0.f;
                    t += __shfl_xor(t, 1, 64);
                    t += __shfl_xor(t, 8, 64);
                    if (ok && (li & 9) == 0) pg[(long long)(wm * 16 + kq * 4 + r) * p.p_cs + (gy >> 1) * p.p_ld + (gx >> 1)] = t * 0.25f;
                }
            }
        }
    }
}

}  // namespace rvc
