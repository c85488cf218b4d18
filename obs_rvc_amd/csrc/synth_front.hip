// synth_front.hip -- persistent kernel for the synthesizer's text encoder, prior sample and flows at one stream (see synth_front.h).
//
// Reference semantics: SynthesizerTrnMs{256,768}NSFsid.infer up to the decoder, restated in
// oracle/rvc_oracle.c synth_forward (TextEncoder, z_p = m + exp(logs) * eps * 0.66666, ResidualCouplingBlock reverse).
//
// Structure.  G = (widest layer's 16-row blocks) x (16-column blocks of T) workgroups of 512 threads stay resident (96 at T = 21,
// one per CU).  A "step" is one layer: Y[M][T] = epi(W[M][K] . X).  Its units are (16-row block of M) x (16-column block of T); unit u
// runs on workgroup u (a workgroup keeps its column block for the whole launch).  The 8 waves of a workgroup have two roles, because
// a wave's vector-memory operations return IN ORDER: a poll queued behind a cold weight load waits for HBM (1.1 us measured):
//   * waves 0-3 ("stagers") only ever touch activations: they sweep the granules of the step's input (agent-scope relaxed atomic
//     loads until every tag matches), build the LDS tile [Cin][24] (LayerNorm applied on the way), and after the MFMAs finish the
//     256 elements of the unit's output tile (thread e = element e) and publish them as 8-byte {tag, value} granules (agent-scope
//     relaxed atomic stores: the data is the flag -- no fences, no cache maintenance, no placement assumption; guideline 16 R2);
//   * waves 4-7 ("MFMA waves") only ever touch parameters: each holds its K-quarter of the unit's weight fragments in registers,
//     requested right after the previous step's last MFMA (a whole epilogue + hand-off + staging ahead of their use), copies the
//     step's bias slice / LayerNorm scale and shift / relative-position tables to LDS while the stagers poll, and issues
//     v_mfma_f32_16x16x4_f32 against the tile.  Partial tiles meet in LDS in a fixed order.
// Two workgroup barriers per step (tile ready / partials ready).  tag = *epoch + step index; the chunk's last kernel advances *epoch
// by 64, so buffers are reused from chunk to chunk without a memset and a captured graph replays correctly.  Every spin is bounded:
// a time-out raises the stream's status word (7).
//
// Fusions the step structure allows (each removes a hand-off): LayerNorm is applied while the consumer builds its tile; attention
// runs inside the output-projection step on the matrix cores (every unit recomputes the 16 query columns it needs); the prior sample
// is the epilogue of the projection to (m, logs); the WaveNet gate is the epilogue of the in-layer (GLU-packed rows, igemm.hip.h
// glu_store); the second FFN convolution is split in two K halves on twice the workgroups, summed by its consumers while staging.
#include "synth_front.h"
#include "state.hip.h"
#include <algorithm>
#include <stdexcept>
#include <string>

namespace rvc {

typedef float sf_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

constexpr int SF_SW = 4, SF_MW = 4;  // stager waves (0..3), MFMA waves (4..7)
constexpr int SF_ST = SF_SW * 64;    // stager threads = elements of a 16 x 16 output tile
constexpr int SF_MAXC = 18;          // weight chunks (16 k each) an MFMA wave holds in registers: 72 / 4 for a K half of the 768 x 3 FFN
constexpr int SF_LW = 24;            // LDS tile row: 16 columns + a halo of 4 on both sides
constexpr int SF_HALO = 4;
constexpr int SF_NB = 16;            // granules a stager thread has in flight per pass
constexpr unsigned SF_SPIN_LIMIT = 1u << 18;

struct SfBuf { u64 *p; unsigned tag; };       // a step's output: granules [rows][NS], valid once their tag equals `tag`
struct SfA { sf_f32x4 a[SF_MAXC]; };

struct SfCtx {
    int g, tid, lane, wave, T, NF, NS;
    int nf, col0;          // this workgroup's 16-column block: unit u = g of every step is (row block g / NF, column block g % NF)
    bool stager;           // waves 0..3
    int mw;                // MFMA wave index 0..3 (waves 4..7)
    int *status;
    bool dead;
    unsigned long long *sub;      // tuning aid: sub-step stamps of workgroup 0
};

__device__ __forceinline__ u64 sf_gload(const u64 *p) { return __hip_atomic_load((const gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sf_gstore(u64 *p, unsigned tag, float v)
{
    __hip_atomic_store((gu64 *)p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one granule, polled until its tag matches (bounded); x = the value of a load issued earlier
__device__ __forceinline__ float sf_wait(SfCtx &c, const u64 *p, unsigned tag, u64 x)
{
    unsigned spins = 0;
    while ((unsigned)(x >> 32) != tag && !c.dead) {
        __builtin_amdgcn_s_sleep(1);
        x = sf_gload(p);
        if (++spins > SF_SPIN_LIMIT) { c.dead = true; *c.status = 7; }
    }
    return __uint_as_float((unsigned)x);
}

// ---------------------------------------------------------------------------------------------------------------- MFMA waves
// K range of MFMA wave `mw` inside the nc chunks of the unit
__device__ __forceinline__ void sf_chunk_range(int nc, int mw, int &c0, int &c1)
{
    const int cpw = (nc + SF_MW - 1) / SF_MW;
    c0 = mw * cpw;
    c1 = c0 + cpw < nc ? c0 + cpw : nc;
    if (c1 < c0) c1 = c0;
}
// request this wave's weight fragments of row block mt, chunks [cb, cb + nc) of the panel (fragment-major [M/16][nchunks][64][4]),
// into register slots slot0, slot0 + 1, ...
__device__ __forceinline__ int sf_slots(int nc) { return (nc + SF_MW - 1) / SF_MW; }
__device__ __forceinline__ void sf_prefetch(SfA &A, const SfW &W, int mt, int cb, int nc, int mw, int lane, int slot0 = 0)
{
    int c0, c1; sf_chunk_range(nc, mw, c0, c1);
    const float *base = W.w + ((size_t)mt * W.nchunks + cb + c0) * 256 + lane * 4;
#pragma unroll
    for (int i = 0; i < SF_MAXC; i++) if (i >= slot0 && c0 + (i - slot0) < c1) A.a[i] = *reinterpret_cast<const sf_f32x4 *>(base + (size_t)(i - slot0) * 256);
}
// this wave's share of the unit's K: 4 MFMAs per chunk, activations from the LDS tile X[ci][SF_LW] (tile column HALO + j = global
// column col0 + j); operand row k = (ci, tap) with k = ci * KW + tap as in the weight panels (prep_conv), k counted from the first
// chunk of the range (the tile's row 0 is the range's first input channel).  Partial tile -> red[mw].
// (Measured alternative: groups of 6 chunks as straight-line code with zero-padded slots -- the operand reads of a group in flight
// before its MFMAs -- spilled registers into scratch and ran 2-3x slower than this per-chunk form.)
template <int KW>
__device__ __forceinline__ void sf_compute(const SfA &A, const SfW &W, int mt, int cb, int nc, int pad, const float *X, float *red, int mw, int lane, int slot0 = 0)
{
    int c0, c1; sf_chunk_range(nc, mw, c0, c1);
    sf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, kq = lane >> 4;
    const float *xb = X + SF_HALO + li - pad;
#define SF_CHUNK(AV, C)                                                                                  \
    {                                                                                                     \
        const int k0_ = (C) * 16 + kq * 4;                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                   \
            const int k_ = k0_ + j, ci_ = KW == 1 ? k_ : k_ / KW, tap_ = KW == 1 ? 0 : k_ - ci_ * KW;     \
            const float b_ = xb[ci_ * SF_LW + tap_];                                                      \
            if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((AV)[j], b_, acc1, 0, 0, 0);           \
            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32((AV)[j], b_, acc0, 0, 0, 0);                 \
        }                                                                                                 \
    }
#pragma unroll
    for (int i = 0; i < SF_MAXC; i++) if (i >= slot0 && c0 + (i - slot0) < c1) SF_CHUNK(A.a[i], c0 + (i - slot0))
    const float *base = W.w + ((size_t)mt * W.nchunks + cb) * 256 + lane * 4;
    for (int c = c0 + SF_MAXC - slot0; c < c1; c++) { const sf_f32x4 a = *reinterpret_cast<const sf_f32x4 *>(base + (size_t)c * 256); SF_CHUNK(a, c) }
#undef SF_CHUNK
    acc0 += acc1;
#pragma unroll
    for (int r = 0; r < 4; r++) red[mw * 256 + r * 64 + lane] = acc0[r];
}
// element e (0..255) of the reduced tile: row ((e & 63) >> 4) * 4 + (e >> 6), column e & 15 (D layout of the 16x16x4 MFMA)
__device__ __forceinline__ float sf_reduced(const float *red, int e)
{
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SF_MW; w++) v += red[w * 256 + e];
    return v;
}

// ---------------------------------------------------------------------------------------------------------------- stagers
// Copy rows [row0, row0 + rows) x global columns [gc0, gc0 + ncols) of a tagged buffer to dst[row * dst_ld + col] (col from 0), each
// value multiplied by `scale` (ADD: added to what is there).  The (row, column) items are dealt to the 256 stager threads flat,
// SF_NB loads in flight per thread and pass; every sweep re-requests ALL values whose tag did not match yet (one round trip per
// sweep, not one per value; bounded).
template <bool ADD>
__device__ __forceinline__ void sf_stage_flat(SfCtx &c, const SfBuf &src, int row0, int rows, int gc0, int ncols, float *dst, int dst_ld, float scale = 1.0f)
{
    const int total = rows * ncols;
    const float inv = 1.0f / (float)ncols;
    const u64 *sp = src.p + (size_t)row0 * c.NS + gc0;
    for (int base = c.tid; base < total; base += SF_ST * SF_NB) {
        u64 v[SF_NB];
        int off[SF_NB];
        unsigned pend = 0;
#pragma unroll
        for (int i = 0; i < SF_NB; i++) {
            const int idx = base + i * SF_ST;
            if (idx < total) {
                const int r = (int)(((float)idx + 0.5f) * inv), cc = idx - r * ncols;
                off[i] = r * c.NS + cc;
                v[i] = sf_gload(sp + off[i]);
                pend |= 1u << i;
            }
        }
        for (unsigned spins = 0; pend;) {
            unsigned still = 0;
#pragma unroll
            for (int i = 0; i < SF_NB; i++)
                if (pend & (1u << i)) {
                    if ((unsigned)(v[i] >> 32) == src.tag) {
                        const int idx = base + i * SF_ST, r = (int)(((float)idx + 0.5f) * inv), cc = idx - r * ncols;
                        const float val = __uint_as_float((unsigned)v[i]) * scale;
                        if (ADD) dst[r * dst_ld + cc] += val; else dst[r * dst_ld + cc] = val;
                    } else still |= 1u << i;
                }
            pend = still;
            if (!pend || c.dead) break;
            if (++spins > SF_SPIN_LIMIT) { c.dead = true; *c.status = 7; break; }
            __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int i = 0; i < SF_NB; i++) if (pend & (1u << i)) v[i] = sf_gload(sp + off[i]);
        }
    }
}
// Build the LDS tile of this workgroup's unit from a tagged buffer: rows [row0, row0 + rows) x global columns [col0 - pad, col0 + 16 + pad)
// clipped to [0, T).  Tile position SF_HALO + j holds global column col0 + j; the positions of columns outside [0, T) were zeroed
// once (sf_zero_invalid: the convolutions' zero padding) and are never written.
template <bool ADD>
__device__ __forceinline__ void sf_stage(SfCtx &c, const SfBuf &src, int row0, int rows, int pad, float *X)
{
    const int lo = c.col0 - pad < 0 ? 0 : c.col0 - pad, hi = c.col0 + 16 + pad > c.T ? c.T : c.col0 + 16 + pad;
    sf_stage_flat<ADD>(c, src, row0, rows, lo, hi - lo, X + SF_HALO + (lo - c.col0), SF_LW);
}
// zero the tile positions of the columns outside [0, T), rows [row_lo, row_hi), with threads t = 0 .. nt - 1
__device__ __forceinline__ void sf_zero_invalid(const SfCtx &c, int t, int nt, float *X, int row_lo, int row_hi)
{
    for (int idx = t; idx < (row_hi - row_lo) * SF_LW; idx += nt) {
        const int r = idx / SF_LW, q = idx - r * SF_LW, gcol = c.col0 - SF_HALO + q;
        if (gcol < 0 || gcol >= c.T) X[(row_lo + r) * SF_LW + q] = 0.f;
    }
}
// the same from a plain float tensor [rows][ld] that was complete before the launch (the gathered phone features)
__device__ __forceinline__ void sf_stage_plain(SfCtx &c, const float *src, int ld, int rows, float *X)
{
    const int tcol = c.tid & 15, trow = c.tid >> 4, gcol = c.col0 + tcol;
    if (gcol >= c.T) return;
    for (int r = trow; r < rows; r += SF_ST / 16) X[r * SF_LW + SF_HALO + tcol] = src[(size_t)r * ld + gcol];
}
// LayerNorm over the rows of the staged tile, per column, in place (two-pass mean / variance as in oracle layernorm_ct; eps 1e-5).
// Padding positions (columns outside [0, T)) stay zero: the zero padding applies AFTER the norm.  Called by ALL waves (four
// workgroup barriers); the stagers do the arithmetic, scale / shift come from LDS (lnp: g[rows] at 0, b[rows] at 256, written by
// the MFMA waves before their first barrier here).
__device__ __forceinline__ void sf_tile_layernorm(SfCtx &c, int rows, int pad, const float *lnp, float *X, float *part, float *cstat)
{
    const int tcol = c.tid & 31, trow = (c.tid >> 5) & 7, ncol = 16 + 2 * pad;
    const int gcol = c.col0 - pad + tcol;
    const bool act = c.stager && tcol < ncol && gcol >= 0 && gcol < c.T;
    float *col = X + SF_HALO - pad + tcol;
    __syncthreads();
    if (c.stager) {
        float s = 0.f;
        if (act) for (int r = trow; r < rows; r += 8) s += col[r * SF_LW];
        part[trow * 32 + tcol] = s;
    }
    __syncthreads();
    if (c.tid < 32) { float m = 0.f; for (int q = 0; q < 8; q++) m += part[q * 32 + c.tid]; cstat[c.tid] = m / (float)rows; }
    __syncthreads();
    if (c.stager) {
        const float mean = cstat[tcol];
        float q2 = 0.f;
        if (act) for (int r = trow; r < rows; r += 8) { const float d = col[r * SF_LW] - mean; q2 += d * d; }
        part[256 + trow * 32 + tcol] = q2;
    }
    __syncthreads();
    if (c.stager) {
        // every stager thread derives its column's rstd itself (8 partials): no fifth barrier
        float v = 0.f;
        for (int q = 0; q < 8; q++) v += part[256 + q * 32 + tcol];
        const float rstd = 1.0f / sqrtf(v / (float)rows + 1e-5f), mean = cstat[tcol];
        if (act) for (int r = trow; r < rows; r += 8) col[r * SF_LW] = (col[r * SF_LW] - mean) * rstd * lnp[r] + lnp[256 + r];
    }
}

#define SF_SUB(k) do { if (c.sub && threadIdx.x == 0) c.sub[(k)] = wall_clock64(); } while (0)
#define SF_EL(e, mt) const int l_ = (e) & 63, m = (mt) * 16 + (l_ >> 4) * 4 + ((e) >> 6), cl = l_ & 15, n = c.col0 + cl; (void)cl;

// static description of a step for the MFMA waves
struct SfStepW {
    SfW W; int cb, nc;            // panel, chunk range of the unit inside a row block (K halves: cb = half * nc)
    int mt;                       // row block of this workgroup's unit (-1: none in this step)
    const float *ln_g, *ln_b;     // LayerNorm of the step's staging (or nullptr)
};

// One step.  S.mt < 0: this workgroup has no unit in the step (it still requests the next step's weights).
//   stagers:     stage(mt) (polls + tile) ............. B1 ......... B2, epi(mt, e, v) on thread e, stores
//   MFMA waves:  bias / LayerNorm parameters -> LDS ... B1, MFMAs, next step's weights requested, B2
// With S.ln_g the staging ends with sf_tile_layernorm over rows_ln rows (all waves: its barriers are workgroup-wide).
template <int KW, class Stage, class Epi>
__device__ __forceinline__ void sf_step(SfCtx &c, const SfStepW &S, int pad, SfA &A, const SfStepW &N, float *X, float *red, float *bias_s, float *lnp,
                                        float *part, float *cstat, int rows_ln, Stage stage, Epi epi)
{
    if (S.mt < 0) { if (!c.stager && N.mt >= 0) sf_prefetch(A, N.W, N.mt, N.cb, N.nc, c.mw, c.lane); return; }
    if (c.stager) stage(S.mt);
    else {
        // parameters of this step: bias slice of the unit's 16 rows, LayerNorm scale / shift
        if (c.mw == 0 && c.lane < 16 && S.W.b) bias_s[c.lane] = S.W.b[S.mt * 16 + c.lane];
        if (S.ln_g) for (int i = c.tid - SF_ST; i < rows_ln; i += SF_ST) { lnp[i] = S.ln_g[i]; lnp[256 + i] = S.ln_b[i]; }
    }
    if (S.ln_g) sf_tile_layernorm(c, rows_ln, pad, lnp, X, part, cstat);
    SF_SUB(0);
    __syncthreads();                                   // B1: tile (and parameters) ready
    if (!c.stager) {
        sf_compute<KW>(A, S.W, S.mt, S.cb, S.nc, pad, X, red, c.mw, c.lane);
        if (N.mt >= 0) sf_prefetch(A, N.W, N.mt, N.cb, N.nc, c.mw, c.lane);
    }
    SF_SUB(1);
    __syncthreads();                                   // B2: partial tiles ready
    SF_SUB(2);
    if (c.stager) epi(S.mt, c.tid, sf_reduced(red, c.tid));
    SF_SUB(3);
}

__global__ __launch_bounds__(SF_THREADS) void synth_front_kernel(SynFrontP p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *red = smem;                      // [2][SF_MW][256]
    float *part = red + 2 * SF_MW * 256;    // [2][8][32]
    float *cstat = part + 512;              // mean[32] (+32 spare)
    float *bias2 = cstat + 64;              // [2][32]: bias slices, by step parity (the epilogue of step s overlaps the parameter copy of step s + 1)
    float *lnp2 = bias2 + 64;               // [2][512]: LayerNorm scale [0..255], shift [256..511], by step parity
    float *X = lnp2 + 1024;                 // tile [rows][SF_LW]; the attention scratch sits behind the first H rows
    SfCtx c;
    c.g = (int)blockIdx.x; c.lane = (int)threadIdx.x & 63; c.wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    c.stager = c.wave < SF_SW; c.mw = c.wave - SF_SW; c.tid = (int)threadIdx.x;       // (stagers: tid 0..255 = their index)
    c.T = p.T; c.NF = (p.T + 15) / 16; c.NS = c.NF * 16; c.status = p.status; c.dead = false; c.sub = nullptr;
    c.nf = c.g % c.NF; c.col0 = c.nf * 16;
    const int T = p.T, H = p.H, F = p.F, I = p.I, half = p.I / 2, NF = c.NF, NS = c.NS;
    const int HT = H / 16;                  // 16-row blocks of a hidden-size tensor
    const int gm = c.g / NF;                // row-block index of this workgroup's unit in every step
    const int ffn_rows = max(max(H, F / 2), I), max_rows = max(ffn_rows, p.C);
    const unsigned base = *p.epoch;
    unsigned step = 0;
    u64 *ws = p.gran;
    // tuning aid (RVC_FRONT_STAMPS=1): workgroup 0 stamps the device wall clock (10 ns ticks) at the start of every step
#define SF_STAMP() do { if (p.stamps && c.g == 0) { if (threadIdx.x == 0) p.stamps[step] = wall_clock64(); c.sub = p.stamps + 128 + step * 4; } } while (0)
    auto alloc = [&](int rows) { SfBuf b; b.p = ws; b.tag = 0; ws += (size_t)rows * NS; return b; };
    // step descriptors: a plain 1x1 / k-tap layer over the whole K of its panel; units = (M / 16) row blocks
    auto sw = [&](const SfW &W) { SfStepW s; s.W = W; s.cb = 0; s.nc = W.nchunks; s.mt = gm < W.M / 16 ? gm : -1; s.ln_g = s.ln_b = nullptr; return s; };
    // the second FFN convolution in two K halves: row block t of half h is unit h * (M / 16) + t
    auto sw_half = [&](const SfW &W) {
        SfStepW s; s.W = W; s.nc = W.nchunks / 2; const int mtn = W.M / 16;
        s.mt = gm < 2 * mtn ? gm % mtn : -1; s.cb = gm < 2 * mtn ? (gm / mtn) * s.nc : 0; s.ln_g = s.ln_b = nullptr; return s;
    };
    SfA A;
    float *const bias_s[2] = {bias2, bias2 + 32};
    float *const lnp[2] = {lnp2, lnp2 + 512};

    // ---- step 1: x = lrelu((W_phone . phone + b + emb_pitch[pitch]) * sqrt(H), 0.1)              (oracle synth_forward, TextEncoder head)
    SfBuf x = alloc(H);
    x.tag = base + ++step; SF_STAMP();
    SfStepW S = sw(p.phone_w), N = sw(p.layer[0].qkv);
    if (!c.stager && S.mt >= 0) sf_prefetch(A, S.W, S.mt, S.cb, S.nc, c.mw, c.lane);
    sf_zero_invalid(c, (int)threadIdx.x, SF_THREADS, X, 0, max_rows);
    __syncthreads();
    {
        const float sq = sqrtf((float)H);
        float emb = 0.f;
        sf_step<1>(c, S, 0, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, 0,
                   [&](int mt) {
                       SF_EL(c.tid, mt)
                       if (n < T) emb = p.pitch_emb[(size_t)p.pitch[n] * H + m];
                       sf_stage_plain(c, p.phone, p.phone_ld, p.C, X);
                   },
                   [&](int mt, int e, float v) {
                       SF_EL(e, mt)
                       if (n < T) {
                           float a = v + bias_s[step & 1][m & 15] + emb;
                           a *= sq;
                           sf_gstore(x.p + (size_t)m * NS + n, x.tag, a > 0.f ? a : a * 0.1f);
                       }
                   });
    }
    // ---- encoder layers.  xin (+ xin2: second K half of the previous FFN) = this layer's input BEFORE the pending LayerNorm
    SfBuf xin = x, xin2 = {nullptr, 0}; const float *ln_g = nullptr, *ln_b = nullptr;
    for (int l = 0; l < p.n_layers; l++) {
        const SfLayer &L = p.layer[l];
        // -- qkv = W_qkv . LN(xin) + b; the first H/16 row blocks also publish xn = LN(xin) (the attention residual)
        SfBuf qkv = alloc(3 * H), xn = alloc(H);
        qkv.tag = xn.tag = base + ++step; SF_STAMP();
        S = N; S.ln_g = ln_g; S.ln_b = ln_b; N = sw(L.o);
        sf_step<1>(c, S, 0, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, H,
                   [&](int) {
                       sf_stage<false>(c, xin, 0, H, 0, X);
                       if (xin2.p) sf_stage<true>(c, xin2, 0, H, 0, X);
                   },
                   [&](int mt, int e, float v) {
                       SF_EL(e, mt)
                       if (n < T) {
                           sf_gstore(qkv.p + (size_t)m * NS + n, qkv.tag, v + bias_s[step & 1][m & 15]);
                           if (mt < HT) sf_gstore(xn.p + (size_t)m * NS + n, xn.tag, X[m * SF_LW + SF_HALO + cl]);
                       }
                   });
        // -- attention with relative positions (oracle relpos_mha) for this unit's 16 query columns, then y = W_o . att + b + xn
        SfBuf xa = alloc(H);
        xa.tag = base + ++step; SF_STAMP();
        S = N; N = sw(L.ff1);
        if (S.mt < 0) { if (!c.stager && N.mt >= 0) sf_prefetch(A, N.W, N.mt, N.cb, N.nc, c.mw, c.lane); }
        else {
            // Attention on the matrix cores (every product below is a 16x16x4 fp32 MFMA; A[i = lane & 15][k = lane >> 4], B[k][j = lane & 15],
            // D[row = (lane >> 4) * 4 + r][col = lane & 15]):
            //   scores[h][i][j] = sum_d q[d][i] k[d][j]                       (rows = the unit's 16 query columns)
            //   P[h][i][r]      = sum_d q[d][i] rel_k[r][d]                   scores[i][j] += P[i][j - i + W] inside the window
            //   out[c][i]       = sum_j v[c][j] S[i][j] + sum_r rel_v[r][d] Ssk[i][r]      with Ssk[i][r] = S[i][i + r - W] (zero outside [0, T))
            // Padded k (j >= T, r >= NR) multiplies a ZEROED S / Ssk entry by a finite A entry; padded rows / columns of D are not stored.
            const int kc = H / p.heads, TP = T | 1, NR = 2 * p.window + 1, NRP = (NR + 3) / 4 * 4, Wd = p.window;
            const int JF = NF, SW = JF * 16, RF = (NR + 15) / 16, PW = RF * 16;
            float *q = X + H * SF_LW, *kk = q + H * 16, *vv = kk + H * TP, *rk = vv + H * TP, *rv = rk + PW * kc;
            float *Sx = rv + NRP * kc, *P = Sx + p.heads * 16 * SW, *Ssk = P + p.heads * 16 * PW;      // Sx [heads][16][SW], P / Ssk [heads][16][PW]
            const float scale = 1.0f / sqrtf((float)kc);
            const int col0 = c.col0, nq = T - col0 < 16 ? T - col0 : 16, mt = S.mt;
            float *bs = bias_s[step & 1];
            u64 rx = 0; const u64 *rp = nullptr;
            if (c.stager) {
                // residual of this thread's output element: requested first, consumed last; q of the own columns (pre-scaled), k / v of all columns
                { SF_EL(c.tid, mt) if (n < T) { rp = xn.p + (size_t)m * NS + n; rx = sf_gload(rp); } }
                sf_stage_flat<false>(c, qkv, 0, H, col0, nq, q, 16, scale);
                if (nq < 16) for (int i = c.tid; i < H * (16 - nq); i += SF_ST) { const int r = i / (16 - nq), cc = i - r * (16 - nq); q[r * 16 + nq + cc] = 0.f; }
                sf_stage_flat<false>(c, qkv, H, 2 * H, 0, T, kk, TP);             // rows H..3H: k then v (vv = kk + H * TP)
            } else {
                // relative-position tables (rows behind NR are zero / finite padding) and the bias slice: parameters, MFMA waves' queue
                const int t = c.tid - SF_ST;
                if (t < 16) bs[t] = S.W.b[mt * 16 + t];
                for (int i = t; i < PW * kc; i += SF_ST) rk[i] = i < NR * kc ? L.rel_k[i] : 0.f;
                for (int i = t; i < NRP * kc; i += SF_ST) rv[i] = i < NR * kc ? L.rel_v[i] : 0.f;
            }
            __syncthreads();
            const int li = c.lane & 15, kq = c.lane >> 4;
            // scores and P: items (head, 16-column block of keys) then (head, 16-row block of relative positions), one per wave and pass
            for (int it = c.wave; it < p.heads * (JF + RF); it += SF_THREADS / 64) {
                const bool is_p = it >= p.heads * JF;
                const int h = is_p ? (it - p.heads * JF) / RF : it / JF, f = is_p ? (it - p.heads * JF) - h * RF : it - h * JF;
                sf_f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                const float *qa = q + (h * kc + kq) * 16 + li;
                const float *bb = is_p ? rk + (f * 16 + li) * kc + kq : kk + (h * kc + kq) * TP + f * 16 + li;
                const int bst = is_p ? 4 : 4 * TP;
                for (int ks = 0; ks < kc / 4; ks += 2) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[ks * 64], bb[ks * bst], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[(ks + 1) * 64], bb[(ks + 1) * bst], a1, 0, 0, 0);
                }
                a0 += a1;
                float *dst = is_p ? P + (h * 16) * PW + f * 16 : Sx + (h * 16) * SW + f * 16;
                const int dw = is_p ? PW : SW;
#pragma unroll
                for (int r = 0; r < 4; r++) dst[(kq * 4 + r) * dw + li] = a0[r];
            }
            __syncthreads();
            // softmax with the relative-position term: 16 lanes per (head, query) row, keys strided over the lanes
            for (int hi = (int)threadIdx.x >> 4; hi < p.heads * 16; hi += SF_THREADS / 16) {
                const int i = hi & 15, gi = col0 + i;
                float *Sr = Sx + hi * SW, *Kr = Ssk + hi * PW;
                const float *Pr = P + hi * PW;
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
            // attention output of the own columns -> the tile (operand of the projection): items (head, 16-channel block)
            for (int it = c.wave; it < p.heads * (kc / 16); it += SF_THREADS / 64) {
                const int h = it / (kc / 16), cf = it - h * (kc / 16), ch0 = h * kc + cf * 16;
                sf_f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                const float *va = vv + (ch0 + li) * TP + kq, *sb = Sx + (h * 16 + li) * SW + kq;
                for (int ks = 0; ks < (T + 3) / 4; ks++) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(va[ks * 4], sb[ks * 4], a0, 0, 0, 0);
                const float *ra = rv + kq * kc + cf * 16 + li, *kb = Ssk + (h * 16 + li) * PW + kq;
                for (int ks = 0; ks < NRP / 4; ks++) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[ks * 4 * kc], kb[ks * 4], a1, 0, 0, 0);
                a0 += a1;
#pragma unroll
                for (int r = 0; r < 4; r++) if (col0 + li < T) X[(ch0 + kq * 4 + r) * SF_LW + SF_HALO + li] = a0[r];
            }
            SF_SUB(0);
            __syncthreads();                               // B1
            if (!c.stager) {
                sf_compute<1>(A, S.W, mt, 0, S.nc, 0, X, red, c.mw, c.lane);
                if (N.mt >= 0) sf_prefetch(A, N.W, N.mt, N.cb, N.nc, c.mw, c.lane);
            }
            SF_SUB(1);
            __syncthreads();                               // B2
            SF_SUB(2);
            if (c.stager) {
                const float resv = rp ? sf_wait(c, rp, xn.tag, rx) : 0.f;
                SF_EL(c.tid, mt)
                if (n < T) sf_gstore(xa.p + (size_t)m * NS + n, xa.tag, sf_reduced(red, c.tid) + bs[m & 15] + resv);
            } else {
                // the scratch overwrote the zero padding of the tile rows behind H: restore it (those rows are next read by the second FFN
                // convolution; the stagers' next writes there come a barrier later)
                sf_zero_invalid(c, c.tid - SF_ST, SF_THREADS - SF_ST, X, H, ffn_rows);
            }
            SF_SUB(3);
        }
        // -- f1 = relu(conv_k(LN1(xa))); the first H/16 row blocks publish xb = LN1(xa) (the FFN residual)
        SfBuf f1 = alloc(F), xb = alloc(H);
        f1.tag = xb.tag = base + ++step; SF_STAMP();
        S = N; S.ln_g = L.ln1_g; S.ln_b = L.ln1_b; N = sw_half(L.ff2);
        {
            const int pad = p.enc_k / 2;
            auto stage = [&](int) { sf_stage<false>(c, xa, 0, H, pad, X); };
            auto epi = [&](int mt, int e, float v) {
                SF_EL(e, mt)
                if (n < T) {
                    const float a = v + bias_s[step & 1][m & 15];
                    sf_gstore(f1.p + (size_t)m * NS + n, f1.tag, a > 0.f ? a : 0.f);
                    if (mt < HT) sf_gstore(xb.p + (size_t)m * NS + n, xb.tag, X[m * SF_LW + SF_HALO + cl]);
                }
            };
            if (p.enc_k == 3) sf_step<3>(c, S, pad, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, H, stage, epi);
            else sf_step<1>(c, S, pad, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, H, stage, epi);
        }
        // -- x2 = conv_k(f1) + b + xb in two K halves: x2a (half 0, with bias and residual) + x2b; LayerNorm 2 is applied by whoever
        //    consumes x2 (and adds the halves while staging)
        SfBuf x2a = alloc(H), x2b = alloc(H);
        x2a.tag = x2b.tag = base + ++step; SF_STAMP();
        S = N;
        {
            const int pad = p.enc_k / 2;
            const bool last = l + 1 == p.n_layers;
            if (last) { N = sw(p.proj); N.mt = gm < p.proj.M / 32 ? gm : -1; }     // (pair step: a unit owns row block t and t + I/16, see below)
            else N = sw(p.layer[l + 1].qkv);
            const int kh = gm / HT;                    // this unit's K half (0 / 1)
            const int crows = F / 2;                   // input channels of a half
            u64 rx = 0; const u64 *rp = nullptr;
            auto stage = [&](int mt) {
                if (kh == 0) { SF_EL(c.tid, mt) if (n < T) { rp = xb.p + (size_t)m * NS + n; rx = sf_gload(rp); } }
                sf_stage<false>(c, f1, kh * crows, crows, pad, X);
            };
            auto epi = [&](int mt, int e, float v) {
                SF_EL(e, mt)
                if (n < T) {
                    if (kh == 0) sf_gstore(x2a.p + (size_t)m * NS + n, x2a.tag, v + bias_s[step & 1][m & 15] + sf_wait(c, rp, xb.tag, rx));
                    else sf_gstore(x2b.p + (size_t)m * NS + n, x2b.tag, v);
                }
            };
            if (p.enc_k == 3) sf_step<3>(c, S, pad, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, 0, stage, epi);
            else sf_step<1>(c, S, pad, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, 0, stage, epi);
        }
        xin = x2a; xin2 = x2b; ln_g = L.ln2_g; ln_b = L.ln2_b;
    }
    // ---- stats = W_proj . LN2(x) + b -> z_p = m + exp(logs) * eps * 0.66666: a unit owns row block t of m AND row block t + I/16 of logs
    //      (its MFMA waves hold both row blocks' fragments: register slots [0, nc4) and [nc4, 2 nc4), nc4 = a wave's chunk slots)
    SfBuf zA = alloc(half), zB = alloc(half);
    zA.tag = zB.tag = base + ++step; SF_STAMP();
    {
        const int IT = I / 16, mt = gm < IT ? gm : -1;
        const SfStepW Nf = sw(p.flow[p.n_flows - 1].pre);
        const int nc = p.proj.nchunks, nc4 = sf_slots(nc);      // second row block's first register slot
        float *bs = bias_s[step & 1], *lp = lnp[step & 1];
        if (mt < 0) { if (!c.stager && Nf.mt >= 0) sf_prefetch(A, Nf.W, Nf.mt, Nf.cb, Nf.nc, c.mw, c.lane); }
        else {
            if (c.stager) {
                sf_stage<false>(c, xin, 0, H, 0, X);
                if (xin2.p) sf_stage<true>(c, xin2, 0, H, 0, X);
            } else {
                const int t = c.tid - SF_ST;
                if (t < 16) { bs[t] = p.proj.b[mt * 16 + t]; bs[16 + t] = p.proj.b[(mt + IT) * 16 + t]; }
                for (int i = t; i < H; i += SF_ST) { lp[i] = ln_g[i]; lp[256 + i] = ln_b[i]; }
                sf_prefetch(A, p.proj, mt + IT, 0, nc, c.mw, c.lane, nc4);       // (row block mt was requested by the previous step)
            }
            sf_tile_layernorm(c, H, 0, lp, X, part, cstat);
            __syncthreads();
            if (!c.stager) {
                sf_compute<1>(A, p.proj, mt, 0, nc, 0, X, red, c.mw, c.lane);
                sf_compute<1>(A, p.proj, mt + IT, 0, nc, 0, X, red + SF_MW * 256, c.mw, c.lane, nc4);
                if (Nf.mt >= 0) sf_prefetch(A, Nf.W, Nf.mt, Nf.cb, Nf.nc, c.mw, c.lane);
            }
            __syncthreads();
            if (c.stager) {
                const uint32_t seed = p.cp->seed, sid = p.st[0].stream_id, chunk = p.st[0].chunk;
                SF_EL(c.tid, mt)
                if (n < T) {
                    const float mean = sf_reduced(red, c.tid) + bs[m & 15], logs = sf_reduced(red + SF_MW * 256, c.tid) + bs[16 + (m & 15)];
                    const int i = m * T + n;
                    float n4[4];
                    philox_normal4(seed, sid, chunk, 0u, (uint32_t)(i >> 2), n4);
                    const float zv = mean + expf(logs) * n4[i & 3] * 0.66666f;
                    if (m < half) sf_gstore(zA.p + (size_t)m * NS + n, zA.tag, zv); else sf_gstore(zB.p + (size_t)(m - half) * NS + n, zB.tag, zv);
                }
            }
        }
        N = Nf;
    }
    // ---- flows, reverse order (oracle: Flip, then coupling_i reverse; the flips are folded into the pre / post weights, engine.hip ModelSY)
    for (int fi = p.n_flows - 1; fi >= 0; fi--) {
        const SfFlow &Fw = p.flow[fi];
        SfBuf &x0 = Fw.flipped ? zB : zA, &x1 = Fw.flipped ? zA : zB;
        // is this the last flow that rewrites this half?  then its result also goes out as plain floats (the decoder's input)
        bool final_x1 = true;
        for (int q = fi - 1; q >= 0; q--) if (p.flow[q].flipped == Fw.flipped) final_x1 = false;
        const int z_row0 = Fw.flipped ? 0 : half;      // rows of x1 in the latent
        // -- hh = W_pre . x0 + b
        SfBuf hh = alloc(H), skip = alloc(H);
        hh.tag = base + ++step; SF_STAMP(); skip.tag = 0;
        S = N; N = sw(Fw.in[0]);
        sf_step<1>(c, S, 0, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, 0,
                   [&](int) { sf_stage<false>(c, x0, 0, half, 0, X); },
                   [&](int mt, int e, float v) { SF_EL(e, mt) if (n < T) sf_gstore(hh.p + (size_t)m * NS + n, hh.tag, v + bias_s[step & 1][m & 15]); });
        for (int j = 0; j < p.wn_layers; j++) {
            // -- acts = tanh(a_t) * sigmoid(a_s): GLU-packed rows (fragment row kq*4 + r: tanh row of channel f*8 + kq*2 + (r&1) for r < 2, its sigmoid row for r >= 2)
            SfBuf acts = alloc(H);
            acts.tag = base + ++step; SF_STAMP();
            S = N; N = sw(Fw.rs[j]);
            {
                const int pad = (p.wn_k - 1) / 2;
                auto stage = [&](int) { sf_stage<false>(c, hh, 0, H, pad, X); };
                auto epi = [&](int mt, int e, float v) {
                    if (e >= 128) return;                                   // r < 2: this thread pairs element e with e + 128 (r + 2)
                    const float v2 = sf_reduced(red, e + 128);
                    const int l_ = e & 63, r = e >> 6, kq = l_ >> 4, n = c.col0 + (l_ & 15);
                    const float ta = v + bias_s[step & 1][kq * 4 + r], sa = v2 + bias_s[step & 1][kq * 4 + r + 2];
                    const int ch = mt * 8 + kq * 2 + (r & 1);
                    if (n < T) sf_gstore(acts.p + (size_t)ch * NS + n, acts.tag, tanhf(ta) * (1.0f / (1.0f + expf(-sa))));
                };
                if (p.wn_k == 5) sf_step<5>(c, S, pad, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, 0, stage, epi);
                else if (p.wn_k == 3) sf_step<3>(c, S, pad, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, 0, stage, epi);
                else sf_step<1>(c, S, pad, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, 0, stage, epi);
            }
            // -- res / skip: rows < H update hh (all but the last layer), the rest accumulate skip
            const bool lastj = j + 1 == p.wn_layers;
            SfBuf hh2 = alloc(H), skip2 = alloc(H);
            hh2.tag = skip2.tag = base + ++step; SF_STAMP();
            S = N; N = sw(lastj ? Fw.post : Fw.in[j + 1]);
            {
                u64 rx = 0; const u64 *rp = nullptr; unsigned rtag = 0;
                auto stage = [&](int mt) {
                    const bool to_skip = lastj || mt >= HT;
                    {
                        SF_EL(c.tid, mt)
                        const int mm = (to_skip && !lastj) ? m - H : m;
                        if (n < T) {
                            if (!to_skip) { rp = hh.p + (size_t)mm * NS + n; rtag = hh.tag; }
                            else if (skip.tag) { rp = skip.p + (size_t)mm * NS + n; rtag = skip.tag; }
                            if (rp) rx = sf_gload(rp);
                        }
                    }
                    sf_stage<false>(c, acts, 0, H, 0, X);
                };
                auto epi = [&](int mt, int e, float v) {
                    SF_EL(e, mt)
                    const bool to_skip = lastj || mt >= HT;
                    const int mm = (to_skip && !lastj) ? m - H : m;
                    if (n < T) {
                        const float a = (rp ? sf_wait(c, rp, rtag, rx) : 0.f) + (v + bias_s[step & 1][m & 15]);
                        if (to_skip) sf_gstore(skip2.p + (size_t)mm * NS + n, skip2.tag, a); else sf_gstore(hh2.p + (size_t)mm * NS + n, hh2.tag, a);
                    }
                };
                sf_step<1>(c, S, 0, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, 0, stage, epi);
            }
            if (!lastj) hh = hh2;
            skip = skip2;
        }
        // -- x1 -= W_post . skip + b
        SfBuf x1n = alloc(half);
        x1n.tag = base + ++step; SF_STAMP();
        S = N;
        if (fi > 0) N = sw(p.flow[fi - 1].pre); else N.mt = -1;
        {
            u64 rx = 0; const u64 *rp = nullptr;
            sf_step<1>(c, S, 0, A, N, X, red, bias_s[step & 1], lnp[step & 1], part, cstat, 0,
                       [&](int mt) {
                           { SF_EL(c.tid, mt) if (n < T) { rp = x1.p + (size_t)m * NS + n; rx = sf_gload(rp); } }
                           sf_stage<false>(c, skip, 0, H, 0, X);
                       },
                       [&](int mt, int e, float v) {
                           SF_EL(e, mt)
                           if (n < T) {
                               const float a = sf_wait(c, rp, x1.tag, rx) - (v + bias_s[step & 1][m & 15]);
                               sf_gstore(x1n.p + (size_t)m * NS + n, x1n.tag, a);
                               if (final_x1) p.z_out[(size_t)(z_row0 + m) * p.z_ld + n] = a;
                           }
                       });
        }
        x1 = x1n;
    }
    ++step; SF_STAMP();
#undef SF_STAMP
}

// ---------------------------------------------------------------------------------------------------------------- host side
bool synth_front_supported(const SynFrontP &p)
{
    if (p.T < 1 || p.T > SF_MAX_T) return false;
    if (p.H % 16 || p.F % 32 || p.I % 32 || p.C % 16 || p.H % p.heads || (p.H / p.heads) % 16) return false;
    if (p.n_layers < 1 || p.n_layers > SF_MAX_LAYERS || p.n_flows < 1 || p.n_flows > SF_MAX_FLOWS || (p.n_flows & 1) || p.wn_layers < 1 || p.wn_layers > 4) return false;
    if ((p.enc_k != 1 && p.enc_k != 3) || (p.wn_k != 1 && p.wn_k != 3 && p.wn_k != 5)) return false;
    if (synth_front_grid(p) > 256 || p.H > 256 || synth_front_steps(p) > 64) return false;          // one unit per workgroup and step; LayerNorm parameters in LDS
    // every K a whole number of 16-row chunks (no zero-padded operand rows: the LDS tile holds exactly Cin rows); the FFN's K halves too
    if ((p.H * p.enc_k) % 16 || (p.F / 2 * p.enc_k) % 16 || (p.H * p.wn_k) % 16 || (p.I / 2) % 16) return false;
    // the (m, logs) pair step keeps both row blocks' fragments of an MFMA wave in its registers
    if (2 * ((p.H / 16 + SF_MW - 1) / SF_MW) > SF_MAXC) return false;
    // both halves of the latent must be rewritten by some flow (their last writers emit the plain copy the decoder reads)
    bool up = false, lo = false;
    for (int i = 0; i < p.n_flows; i++) { if (p.flow[i].flipped) lo = true; else up = true; }
    if (!up || !lo) return false;
    return synth_front_lds_bytes(p) <= 160 * 1024;
}

// workgroups: the widest step's 16-row blocks (FFN filter, qkv, gated in-layer, or the two K halves of the second FFN convolution)
// x the 16-column blocks of T
int synth_front_grid(const SynFrontP &p) { const int NF = (p.T + 15) / 16; return std::max(std::max(p.F, 3 * p.H), 2 * p.H) / 16 * NF; }

int synth_front_steps(const SynFrontP &p) { return 1 + 4 * p.n_layers + 1 + p.n_flows * (2 + 2 * p.wn_layers); }

size_t synth_front_ws_granules(const SynFrontP &p)
{
    const size_t NS = (size_t)((p.T + 15) / 16) * 16;
    size_t rows = p.H;                                                          // x
    rows += (size_t)p.n_layers * (3 * p.H + p.H + p.H + p.F + p.H + 2 * p.H);   // qkv, xn, xa, f1, xb, x2a, x2b
    rows += p.I;                                                                // zA, zB
    rows += (size_t)p.n_flows * (2 * p.H + (size_t)p.wn_layers * 3 * p.H + p.I / 2);
    return rows * NS;
}

size_t synth_front_lds_bytes(const SynFrontP &p)
{
    const size_t TP = (size_t)p.T | 1, kc = p.H / p.heads, NR = 2 * (size_t)p.window + 1;
    const size_t fixed = 2 * SF_MW * 256 + 512 + 64 + 64 + 1024;
    const size_t rows = (size_t)std::max(std::max(p.H, p.F / 2), std::max(p.C, p.I));
    const size_t tile = rows * SF_LW;
    const size_t NRP = (NR + 3) / 4 * 4, PW = (NR + 15) / 16 * 16, SW = (size_t)((p.T + 15) / 16) * 16;
    const size_t attn = (size_t)p.H * SF_LW + (size_t)p.H * 16 + 2 * p.H * TP + (PW + NRP) * kc + (size_t)p.heads * 16 * (SW + 2 * PW) + 64;
    return (fixed + std::max(tile, attn)) * sizeof(float);
}

void launch_synth_front(const SynFrontP &p, hipStream_t s)
{
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)synth_front_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            throw std::runtime_error("synth_front: cannot raise the dynamic LDS limit");
        attr_done = true;
    }
    hipLaunchKernelGGL(synth_front_kernel, dim3(synth_front_grid(p)), dim3(SF_THREADS), synth_front_lds_bytes(p), s, p);
}

}  // namespace rvc
