// plan.hip -- the planner of the MI355X-native RVC engine: test-hook table, device memory helpers, prepared convolution weights (MFMA-fragment
// order), and the translation of a layer (Conv1d / ConvT1d / Conv2d / ConvT2d / Linear, fused variants) into implicit-GEMM launches: tile
// choice, in-workgroup K split, streams folded into N, staged-tile convolution, 32x32x2 throughput kernels.
#include "engine_int.h"
#include <chrono>

namespace rvc {

static const char *const kTestHooks[] = {"RVC_FORCE_CFG", "RVC_CONV_TILE", "RVC_CONV_TILE_KS", "RVC_NO_LN_FUSE", "RVC_NO_CONV0_MULTI", "RVC_KNN_NO_GEMM",
                                         "RVC_KNN_EXHAUSTIVE", "RVC_STAMPS", "RVC_SERIAL_BRANCHES", "RVC_NO_WN_COMPOSE", "RVC_KNN_LOSE_TICKET", "RVC_FORCE_G2W", "RVC_F0_XCDS", "RVC_CONV32S", "RVC_CONV32S_TILE", "RVC_G32L", "RVC_G32L_TALL", "RVC_G32L_TAB", "RVC_CONV32S_BUF", "RVC_FORCE_CHOICE", "RVC_G32L_PANEL", "RVC_MEAN3", "RVC_RM_FUSE", "RVC_G2W_LN", "RVC_RELPOS_MFMA_MAX"};
std::atomic<unsigned> g_opt_gen{0};       // bumped by every rvc_debug_option call: plans built under another generation are dropped (engine.hip get_plan)
static std::mutex g_opt_mu;
static std::map<std::string, std::string> g_opts;
static bool is_test_hook(const char *name)
{
    for (const char *h : kTestHooks) if (!strcmp(h, name)) return true;
    return false;
}
static const char *opt_lookup(const char *name)
{
    static thread_local std::string buf;
    std::lock_guard<std::mutex> lk(g_opt_mu);
    auto it = g_opts.find(name);
    if (it == g_opts.end()) return nullptr;
    buf = it->second;
    return buf.c_str();
}
#ifdef RVC_TUNING
const char *test_opt(const char *name) { const char *v = opt_lookup(name); return v ? v : getenv(name); }
const char *tune_env(const char *name) { const char *v = opt_lookup(name); return v ? v : getenv(name); }
#else
const char *test_opt(const char *name) { return opt_lookup(name); }
#endif
int test_opt_int(const char *name, int dflt) { const char *v = test_opt(name); return v ? atoi(v) : dflt; }

// Weight memory.  A model is ~1 200 tensors from a few bytes to 9 MB; one hipMalloc each gave every tensor its own mapping, placed wherever the
// driver's VRAM manager had a hole: the weights of an engine that is created after others have come and gone end up in small scattered page
// fragments, and the one-stream f0 branch -- 361 MB of weights streamed once per chunk through the 32 CUs of one XCD behind one L2 TLB -- ran up
// to 18 % slower for the sixth engine of a process than for the first (tests/tools/late_engine.py).  Weights are now
// bump-allocated from 256 MB slabs, each ONE hipMalloc (contiguous, maximal page fragments), reference-counted: a slab is returned to the driver
// when its last tensor is freed.
// Slabs are per (device, class): a process may drive several GPUs (rvc_create(device)), and loads on two devices may interleave -- every device
// has its own open slabs, all of them searched for room (with one global "current" slab, alternating loads opened a fresh 256 MB slab
// per tensor).  Class 0 = model weights (live as long as the model), class 1 = plan-lifetime copies (one-row tap panels, split-bf16 panels): kept in
// slabs of their own so that evicting a plan returns its memory instead of pinning a slab of model weights.
namespace {
struct WSlab { char *base; size_t size, used; long live; int dev, cls; };
std::mutex g_wslab_mu;
std::vector<WSlab> g_wslabs;
const size_t kWSlabBytes = (size_t)256 << 20, kPlanSlabBytes = (size_t)64 << 20;
void *wmalloc_cls(size_t bytes, int cls)
{
    bytes = (std::max<size_t>(bytes, 16) + 255) / 256 * 256;
    std::lock_guard<std::mutex> lk(g_wslab_mu);
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    const size_t slab = cls ? kPlanSlabBytes : kWSlabBytes;
    if (bytes <= slab / 4) {
        for (size_t i = g_wslabs.size(); i-- > 0; ) {          // newest first: the open slab of this (device, class) is usually the last one made
            WSlab &b = g_wslabs[i];
            if (b.dev == dev && b.cls == cls && b.size == slab && b.used + bytes <= b.size) {
                void *r = b.base + b.used; b.used += bytes; b.live++;
                return r;
            }
        }
    }
    const size_t sz = bytes <= slab / 4 ? slab : bytes;
    void *c;
    HIPCHK(hipMalloc(&c, sz));
    g_wslabs.push_back(WSlab{(char *)c, sz, bytes, 1, dev, cls});
    return c;
}
}
void *wmalloc(size_t bytes) { return wmalloc_cls(bytes, 0); }
void *wmalloc_plan(size_t bytes) { return wmalloc_cls(bytes, 1); }
void wfree(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_wslab_mu);
    for (size_t i = 0; i < g_wslabs.size(); i++) {
        WSlab &b = g_wslabs[i];
        if ((char *)p >= b.base && (char *)p < b.base + b.size) {
            if (--b.live == 0) { (void)hipFree(b.base); g_wslabs.erase(g_wslabs.begin() + i); }
            return;
        }
    }
    (void)hipFree(p);          // not from a slab
}
// slabs alive on `dev` (all classes): (count, bytes) -- test aid (rvc_debug_weight_slabs)
void wslab_info(int dev, int *count, size_t *bytes)
{
    std::lock_guard<std::mutex> lk(g_wslab_mu);
    *count = 0; *bytes = 0;
    for (const WSlab &b : g_wslabs) if (b.dev == dev) { (*count)++; *bytes += b.size; }
}

float *upload_f(const std::vector<float> &v, int cls)
{
    float *d = (float *)(cls ? wmalloc_plan(std::max<size_t>(v.size(), 4) * sizeof(float)) : wmalloc(std::max<size_t>(v.size(), 4) * sizeof(float)));
    if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return d;
}
float *upload_f(const float *p, size_t n) { return upload_f(std::vector<float>(p, p + n)); }

// [nphase][M][Kp] row-major panels -> MFMA-fragment-major [nphase][m_tile][chunk][lane][4] (M padded to 16 with zeros)
float *upload_fragments(const std::vector<float> &panel, int nphase, int M, int Kp, int cls)
{
    const int mt = (M + 15) / 16, nch = Kp / 16;
    std::vector<float> out((size_t)nphase * mt * nch * 256, 0.f);
    for (int ph = 0; ph < nphase; ph++)
        for (int t = 0; t < mt; t++)
            for (int c = 0; c < nch; c++)
                for (int l = 0; l < 64; l++) {
                    const int m = t * 16 + (l & 15);
                    if (m >= M) continue;
                    const float *src = &panel[((size_t)ph * M + m) * Kp + c * 16 + (l >> 4) * 4];
                    float *dst = &out[(((size_t)ph * mt + t) * nch + c) * 256 + l * 4];
                    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
                }
    return upload_f(out, cls);
}

// Conv (any rank flattened to K = Cin/groups * KW taps): w [Cout][Cin/groups][KW]
ConvW prep_conv(const float *w, const float *bias, int Cout, int Cin, int KW, int groups)
{
    ConvW c;
    c.Cin = Cin; c.Cout = Cout; c.KW = KW; c.groups = groups; c.nphase = groups;
    int cig = Cin / groups, cog = Cout / groups;
    c.M = cog; c.K = cig * KW; c.Kp = round16(c.K);
    std::vector<float> panel((size_t)Cout * c.Kp, 0.f);
    for (int co = 0; co < Cout; co++) memcpy(&panel[(size_t)co * c.Kp], w + (size_t)co * c.K, (size_t)c.K * sizeof(float));
    c.w = upload_fragments(panel, groups, cog, c.Kp);
    if (bias) c.bias = upload_f(bias, Cout);
    if (KW == 9 && groups == 1 && c.K >= 1024) c.host_w.assign(w, w + (size_t)Cout * c.K);
    return c;
}
// ConvTranspose1d: w [Cin][Cout][K], stride S -> S polyphase sub-convolutions with ntaps = ceil(K/S) taps:
//   out[co][q*S + p - pad] = sum_ci sum_j w[ci][co][p + j*S] * in[ci][q - j]
ConvW prep_convT1d(const float *w, const float *bias, int Cin, int Cout, int K, int S)
{
    ConvW c;
    c.transposed = true; c.Cin = Cin; c.Cout = Cout; c.KW = K; c.S = S; c.ntaps = (K + S - 1) / S; c.nphase = S;
    c.M = Cout; c.K = Cin * c.ntaps; c.Kp = round16(c.K);
    std::vector<float> panel((size_t)S * Cout * c.Kp, 0.f);
    for (int p = 0; p < S; p++)
        for (int co = 0; co < Cout; co++)
            for (int ci = 0; ci < Cin; ci++)
                for (int j = 0; j < c.ntaps; j++) {
                    int k = p + j * S;
                    if (k < K) panel[((size_t)p * Cout + co) * c.Kp + ci * c.ntaps + j] = w[((size_t)ci * Cout + co) * K + k];
                }
    c.w = upload_fragments(panel, S, Cout, c.Kp);
    if (bias) c.bias = upload_f(bias, Cout);
    return c;
}
// ConvTranspose2d 3x3 stride 2 pad 1 output_pad 1: w [Cin][Cout][3][3] -> 4 phases (oh&1, ow&1), 2x2 taps each
//   out[2a+ph][2b+pw] = sum_ci sum_{jh,jw} Wp[ph,pw][co][ci][jh][jw] * in[a+jh][b+jw]
//   even output row: kh = 1 (jh = 0); odd: kh = 2 (jh = 0), kh = 0 (jh = 1); same along w
ConvW prep_convT2d(const float *w, const float *bias, int Cin, int Cout)
{
    ConvW c;
    c.transposed = true; c.Cin = Cin; c.Cout = Cout; c.KW = 9; c.S = 2; c.ntaps = 4; c.nphase = 4;
    c.M = Cout; c.K = Cin * 4; c.Kp = round16(c.K);
    std::vector<float> panel((size_t)4 * Cout * c.Kp, 0.f);
    auto ktap = [](int par, int j) { return par == 0 ? (j == 0 ? 1 : -1) : (j == 0 ? 2 : 0); };
    for (int ph = 0; ph < 2; ph++)
        for (int pw = 0; pw < 2; pw++)
            for (int co = 0; co < Cout; co++)
                for (int ci = 0; ci < Cin; ci++)
                    for (int jh = 0; jh < 2; jh++)
                        for (int jw = 0; jw < 2; jw++) {
                            int kh = ktap(ph, jh), kw = ktap(pw, jw);
                            if (kh < 0 || kw < 0) continue;
                            panel[((size_t)(ph * 2 + pw) * Cout + co) * c.Kp + ci * 4 + jh * 2 + jw] = w[(((size_t)ci * Cout + co) * 3 + kh) * 3 + kw];
                        }
    c.w = upload_fragments(panel, 4, Cout, c.Kp);
    if (bias) c.bias = upload_f(bias, Cout);
    return c;
}
// ---- conv32s_kernel's weight order, built on the device and cached per source panel (every plan build copied every eligible decoder panel to
// the host, repacked it in a six-deep CPU loop and uploaded a private copy -- once per layer, per plan, per trial) ----
static __global__ void c32s_repack_kernel(const float *src, float *dst, int mt, int nchunks, int KW, int nblk)
{
    // dst[((t * nchunks + (blk * KW + tap) * 2 + g) * 64 + l) * 4 + j]  <-  source k = (blk * 32 + g * 16 + (l >> 4) * 4 + j) * KW + tap of the 16-row fragment packing
    const long long total = (long long)mt * nchunks * 256;
    for (long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x; d < total; d += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(d & 3), l = (int)((d >> 2) & 63);
        const long long tc = d >> 8;
        const int c = (int)(tc % nchunks), t = (int)(tc / nchunks);
        const int g = c & 1, bt = c >> 1, tap = bt % KW, blk = bt / KW;
        if (blk >= nblk) { dst[d] = 0.f; continue; }
        const int k = (blk * 32 + g * 16 + (l >> 4) * 4 + j) * KW + tap;
        dst[d] = src[(((long long)t * nchunks + k / 16) * 64 + ((((k % 16) / 4) << 4) | (l & 15))) * 4 + (k % 4)];
    }
}
namespace {
struct PanelKey { const float *src; int M, nchunks, cin, KW; bool operator<(const PanelKey &o) const { return std::tie(src, M, nchunks, cin, KW) < std::tie(o.src, o.M, o.nchunks, o.cin, o.KW); } };
std::mutex g_panel_mu;
std::map<PanelKey, float *> g_panels;
}
const float *c32s_panel(const float *src_frag, int M, int nchunks, int cin, int KW)
{
    std::lock_guard<std::mutex> lk(g_panel_mu);
    const PanelKey key{src_frag, M, nchunks, cin, KW};
    auto it = g_panels.find(key);
    if (it != g_panels.end()) return it->second;
    const int mt = (M + 15) / 16;
    const size_t n = (size_t)mt * nchunks * 256 + (size_t)16 * 256;      // (+ slack: a wave's weight requests run one chunk past its last)
    float *dst = (float *)wmalloc(n * sizeof(float));
    HIPCHK(hipMemsetAsync(dst + (size_t)mt * nchunks * 256, 0, (size_t)16 * 256 * sizeof(float), nullptr));
    hipLaunchKernelGGL(c32s_repack_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, nullptr, src_frag, dst, mt, nchunks, KW, cin / 32);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    g_panels[key] = dst;
    return dst;
}
static void c32s_forget(const float *base, size_t elems)
{
    std::lock_guard<std::mutex> lk(g_panel_mu);
    for (auto it = g_panels.begin(); it != g_panels.end();) {
        if (it->first.src >= base && it->first.src < base + elems) { wfree(it->second); it = g_panels.erase(it); }
        else ++it;
    }
}

void free_conv(ConvW &c)
{
    if (c.w) c32s_forget(c.w, (size_t)c.nphase * (size_t)phase_stride(c));
    if (c.owns) {
        if (c.w) wfree(c.w);
        if (c.bias) wfree(c.bias);
    }
    c.w = c.bias = nullptr;
}
// Re-home the weights of several convolutions in ONE device allocation (the first one owns it), so that a fused launch can
// address them as phases of one weight buffer (PhaseD::w_off / bias_off are offsets from the first conv's pointers).
void merge_convs(const std::vector<ConvW *> &cs)
{
    size_t tw = 0, tb = 0;
    for (ConvW *c : cs) { if (!c->owns || !c->bias) throw std::runtime_error("merge_convs: unexpected conv"); tw += (size_t)c->nphase * phase_stride(*c); tb += (size_t)c->Cout; }
    float *W, *Bv;
    W = (float *)wmalloc(tw * sizeof(float)); Bv = (float *)wmalloc(std::max<size_t>(tb, 4) * sizeof(float));
    size_t ow = 0, ob = 0;
    for (size_t i = 0; i < cs.size(); i++) {
        ConvW *c = cs[i];
        const size_t nw = (size_t)c->nphase * phase_stride(*c);
        HIPCHK(hipMemcpy(W + ow, c->w, nw * sizeof(float), hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(Bv + ob, c->bias, (size_t)c->Cout * sizeof(float), hipMemcpyDeviceToDevice));
        wfree(c->w); wfree(c->bias);
        c->w = W + ow; c->bias = Bv + ob; c->owns = i == 0;
        ow += nw; ob += c->Cout;
    }
}

// (launch_igemm2 / launch_igemm_v1 / launch_igemm_tiled: igemm_launch.h -- the template instantiations are separate translation units)

unsigned long long *g_kprobe = nullptr;   // tuning build (-DRVC_KPROBE): destination of the per-wave phase stamps
int g_last_waves = 0, g_last_wgs = 0;
char g_last_kernel[16] = "";
static void note_kernel(const char *d) { size_t i = 0; for (; i < sizeof(g_last_kernel) - 1 && d[i] && d[i] != ' '; i++) g_last_kernel[i] = d[i]; g_last_kernel[i] = 0; }

// One stream, stride-1 1-D convolution with a long output: conv_tile_kernel (conv_tile.hip.h) stages the input rows once per workgroup.
// Builds the LDS-offset tables (k -> row * RS + tap column) from the layer's gather table and a work-item table that balances the
// unequal phases of a fused launch over the CUs (workgroup b lands on CU b % ncu: tests/tools/place_probe.hip).  false = not eligible.
int g_ncu = 256;

// Plan-time selection by measurement (rvc_set_plan_autotune, queue_igemm below): a trial build of a layer runs under a forced Choice -- the same switch points
// the test hooks RVC_CONV32S_TILE / RVC_FORCE_G2W / RVC_FORCE_CFG use, per layer and thread-local instead of process-wide.  A choice the layer is not eligible
// for falls through to the rules (the trial then reports the same kernel description as the rule-based build and is not timed twice).
struct Choice {
    int kind = 0;      // 0 = the rules; 1 = conv32s_kernel (a = tile | 4 for the buffer-load variant of tile 1); 2 = igemm2w_kernel (tile a, K split b);
    int a = 0, b = 0;  // 3 = workgroup-tiled kernels (lds_cfg a: 3 4 5 7 8); 4 = register-direct kernel (tile a, K split b)
    bool operator==(const Choice &o) const { return kind == o.kind && a == o.a && b == o.b; }
};
static thread_local Choice t_choice;

static bool queue_conv_tile(Plan &pl, IgemmP &p, int B, const std::vector<int> &koff, const std::vector<PhaseD> &phv, double ksum, bool final_out)
{
    const int mode = t_choice.kind ? 0 : test_opt_int("RVC_CONV_TILE", 1);       // test hook: 0 = off, 2 = wherever eligible; read per plan
    auto no = [&](int why) { (void)why; return false; };
    // streams: one always; two to four with the same narrow tiles and the streams in the item table (measured -1 % / -2 % at 2 / 4 streams, nothing at
    // 8; wider tiles for many streams measured slower than the 32x32x2 kernels and are gone)
    if (!mode || p.fold_n || p.x_ld <= 0 || p.x_hs || p.x_ws != 1 || p.y_hm || p.lin_cs4 || p.glu || p.ln_wsum || p.ln_stats_in || p.ln_stats_out || p.part) return no(1);
    if (B > 4) return no(2);
    if (p.M > 128 && mode < 2) return no(3);
    const int kshares = test_opt_int("RVC_CONV_TILE_KS", 2);      // test hook: 1 = one wave per fragment set
    const int tc0 = p.M > 64 ? 0 : (p.M > 32 ? 1 : 2);            // 128 x 16, 64 x 32, 32 x 64
    const int BM = kTileBM[tc0], BN = kTileBN[tc0];
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    if (ntm > 255 || ntn > 32767 || phv.size() > 255) return no(4);
    const long long nitems = (long long)ntm * ntn * (long long)phv.size() * B;
    if (mode < 2 && nitems < 3 * g_ncu / 2) return no(5);                // short outputs: the K-split kernel fills the chip better
    // per phase: (channel, tap) of every k from the gather table (entries are ci * ld + tap * dil - pad, k = ci * KW + tap); the kernel walks K
    // tap-major in chunks of 16 channels, so the phase's weights are repacked: chunk t * G + g, slot kk <- k = (g * 16 + kk) * KW + t
    std::vector<PhaseD> phs(phv);
    std::vector<float> wnew;
    size_t lds_max = 0;
    const int mt = (p.M + 15) / 16;
    for (PhaseD &q : phs) {
        const int K = q.nchunks * 16;
        int cin = 1;
        for (int k = 0; k < K; k++) cin = std::max(cin, (int)std::floor((double)koff[q.koff_off + k] / p.x_ld + 0.5) + 1);
        if (cin % 16 != 0 || K % cin != 0) return no(6);
        const int KW = K / cin;
        if (KW > 255) return no(7);
        const int dmin = koff[q.koff_off];
        const int dil = KW > 1 ? koff[q.koff_off + 1] - koff[q.koff_off] : 1;
        if (dil < 1 || dil > 255 || dmin > 0 || dmin < p.x_lo) return no(8);
        for (int k = 0; k < K; k++) if (koff[q.koff_off + k] != (k / KW) * p.x_ld + dmin + (k % KW) * dil) return no(9);
        const int rl = BN + (KW - 1) * dil, rt = rl | 1, cs = cin + 8;
        q.t_tab = KW | (dil << 8); q.t_cin = cin; q.t_rs = rt; q.t_dmin = dmin;
        if (q.nchunks < 2 || cin / 16 < kshares) return no(10);          // (every K share needs a chunk; the kernel steps its tap / group counters by the share count)
        lds_max = std::max(lds_max, (std::max<size_t>(((size_t)cin * rt + 63) / 64 * 64, (size_t)kTileWF[tc0] * 256) + (size_t)rl * cs) * 4);
        std::vector<float> wold((size_t)mt * q.nchunks * 256);
        HIPCHK(hipMemcpy(wold.data(), p.w + q.w_off, wold.size() * 4, hipMemcpyDeviceToHost));
        const size_t base = wnew.size();
        wnew.resize(base + wold.size());
        const int G = cin / 16;
        for (int t = 0; t < mt; t++)
            for (int tap = 0; tap < KW; tap++)
                for (int g = 0; g < G; g++)
                    for (int l = 0; l < 64; l++)
                        for (int j = 0; j < 4; j++) {
                            const int k = (g * 16 + (l >> 4) * 4 + j) * KW + tap;          // the source's k
                            wnew[base + (((size_t)t * q.nchunks + tap * G + g) * 64 + l) * 4 + j] =
                                wold[(((size_t)t * q.nchunks + k / 16) * 64 + (((k % 16) / 4) << 4 | (l & 15))) * 4 + (k % 4)];
                        }
        q.w_off = (long long)base;
    }
    if (lds_max > 100 * 1024) return no(11);
    wnew.resize(wnew.size() + (size_t)16 * 2 * 256, 0.f);      // slack: the kernel's weight requests run DA x KS chunks past a wave's last chunk
    p.w = pl.arena.upload(wnew);
    // work items, longest first, dealt to the CUs by longest-processing-time; block r * ncu + j = the r-th item of CU j
    struct It { int w, code, b; };
    std::vector<It> items;
    for (int bb = 0; bb < B; bb++)
        for (size_t f = 0; f < phs.size(); f++)
            for (int tm = 0; tm < ntm; tm++)
                for (int tn = 0; tn < ntn; tn++) items.push_back({phs[f].nchunks + 12, (int)f | (tm << 8) | (tn << 16), bb});
    std::stable_sort(items.begin(), items.end(), [](const It &a, const It &b) { return a.w > b.w; });
    const int nb = g_ncu;
    std::vector<std::vector<int>> bins(nb);          // indices into items
    {
        std::vector<std::pair<long long, int>> heap;        // (load, bin): min-heap by load, then bin
        for (int j = 0; j < nb; j++) heap.push_back({0, j});
        auto cmp = [](const std::pair<long long, int> &a, const std::pair<long long, int> &b) { return a > b; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        for (size_t i = 0; i < items.size(); i++) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto &top = heap.back();
            bins[top.second].push_back((int)i); top.first += items[i].w;
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    size_t rounds = 0;
    for (auto &bn : bins) rounds = std::max(rounds, bn.size());
    // Block r * nb + j reaches XCD j % 8 (round-robin in dispatch order), and every XCD has its own L2: with neighbouring n-tiles on different
    // XCDs every tile fetched its own copy of the lines it shares with its neighbours (a 16-column tile with a 50-column halo on each side reads
    // five cache lines and owns half of one; round 4 PMC pass per kernel: 30.6 MB per launch for 2 MB of operands).  Slot j of a round therefore
    // takes bin (j % 8) * (nb / 8) + j / 8: an XCD gets CONSECUTIVE bins = consecutive n-tiles.
    const bool xcd_bins = nb % 8 == 0 && !tune_env("RVC_NO_TILE_XCD");
    auto slot_bin = [&](int j) { return xcd_bins ? (j % 8) * (nb / 8) + j / 8 : j; };
    std::vector<int> order(rounds * nb * 2, -1);
    for (int j = 0; j < nb; j++) {
        const std::vector<int> &bn = bins[slot_bin(j)];
        for (size_t r = 0; r < bn.size(); r++) { order[(r * nb + j) * 2] = items[bn[r]].code; order[(r * nb + j) * 2 + 1] = items[bn[r]].b; }
    }
    p.items = pl.arena.upload(order);
    p.ttab = nullptr;
    p.ph = pl.arena.upload(phs);
    p.nphase = (int)phs.size();
    p.ph0 = phs[0];
    p.ntm = ntm; p.ntn = ntn; p.ksplit = 1; p.nbatch = B; p.m_fast = 0;
    const dim3 grid((unsigned)(order.size() / 2), 1u);
    g_last_wgs = (int)nitems; g_last_waves = 4;
    const double flops = 2.0 * p.M * (double)p.N * ksum * B;
    pl.igemm_flops += flops; pl.n_igemm++;
    Plan *plp = &pl;
    { char d[200]; snprintf(d, sizeof d, "tile M=%d N=%d K=%d B=%d nph=%d tile=%dx%d items=%lld grid=%u lds=%zu pre=%d ksum=%.0f", p.M, p.N, p.K, B, p.nphase, BM, BN, nitems, grid.x, lds_max, (int)(p.pre_act != ACT_NONE), ksum); pl.descs.push_back(d); note_kernel(d); }
    const int desc_id = (int)pl.descs.size() - 1;
    const IgemmP pc = p;
    if (final_out) pl.final_out_honoured = true;
    pl.ops.push_back([=](hipStream_t s) {
        ProfEvent *pe = nullptr;
        if (plp->profile) {
            if (plp->prof_used == plp->prof.size()) { ProfEvent e; HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b)); e.flops = 0; e.bytes = 0; plp->prof.push_back(e); }
            pe = &plp->prof[plp->prof_used++]; pe->flops = flops; pe->bytes = 0; pe->desc = desc_id;
        }
        hipEvent_t ea = pe ? pe->a : nullptr, eb = pe ? pe->b : nullptr;
        if (final_out && plp->cur_out) { IgemmP q = pc; q.y = plp->cur_out; q.y_bs = plp->cur_out_bs; launch_conv_tile(tc0, kshares, q, grid, lds_max, s, ea, eb); }
        else launch_conv_tile(tc0, kshares, pc, grid, lds_max, s, ea, eb);
    });
    return true;
}

// More than 4 streams, stride-1 1-D convolution whose input channels come in 32s: conv32s_kernel (conv32s.hip.h) stages the input rows of a 32-channel
// block once per workgroup and walks the taps from LDS (32x32x2 MFMAs); igemm32_kernel re-gathers the activation tile for every 16-deep K step, i.e.
// once per tap.  The streams stay a grid dimension (tiles never straddle streams).  Test hook RVC_CONV32S: 0 = off, 2 = wherever eligible (any stream
// count, any size), "RVC_CONV32S_TILE" forces a tile (0..2).  false = not eligible.
static bool queue_conv32s(Plan &pl, IgemmP &p, int B, const std::vector<int> &koff, const std::vector<PhaseD> &phv, double ksum, bool final_out)
{
    if (t_choice.kind != 0 && t_choice.kind != 1) return false;          // (a trial of another kernel family)
    const bool chosen = t_choice.kind == 1;                               // (a trial of THIS family: the size rules below are what is being measured)
    const int mode = chosen ? 2 : test_opt_int("RVC_CONV32S", 1);
    if (!mode || p.fold_n || p.x_ld <= 0 || p.x_hs || p.x_ws != 1 || p.y_hm || p.y_ws != 1 || p.glu || p.ln_wsum || p.ln_stats_in || p.ln_stats_out || p.part || p.bf3 || pl.bf3) return false;
    if (mode < 2 && (B <= 4 || p.N < 200 || p.M < 32)) return false;
    for (const PhaseD &q : phv) if (q.act_p1 != 0 || q.y_off != 0 || q.y_pos != 0) return false;
    // tile by the height of the weight panel; every wave owns 32 x 64 outputs (round 5 sweep of six tiles per layer at 8 / 16 / 32 / 64 streams, gpurun_out
    // of tests/tools/c32s_layers.py: 2 x 2 accumulator blocks per wave -- 128 x 128, 64 x 256 -- lose to these at every count but 64, where they tie)
    auto wgs_of = [&](int t) { return (long long)((p.M + kC32sBM[t] - 1) / kC32sBM[t]) * ((p.N + kC32sBN[t] - 1) / kC32sBN[t]) * (long long)phv.size() * B; };
    int tile = p.M <= 32 ? 0 : (p.M <= 64 ? 1 : 2);
    const int forced = chosen ? (t_choice.a & 3) : test_opt_int("RVC_CONV32S_TILE", -1);
    if (forced >= 0 && forced <= 2) tile = forced;
    // under two workgroups per CU the register-direct kernels with their K split win (256-row stage at 32 streams: 684 vs 720 us)
    if (mode < 2 && wgs_of(tile) < 2 * g_ncu) return false;
    const int BM = kC32sBM[tile], BN = kC32sBN[tile], CB = kC32sCB;
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    if (phv.size() > 65535 || B > 65535) return false;
    std::vector<PhaseD> phs(phv);
    size_t lds_max = 0;
    for (PhaseD &q : phs) {
        const int K = q.nchunks * 16;
        int cin = 1;
        for (int k = 0; k < K; k++) cin = std::max(cin, (int)std::floor((double)koff[q.koff_off + k] / p.x_ld + 0.5) + 1);
        if (cin % CB != 0 || K % cin != 0) return false;
        const int KW = K / cin;
        if (KW > 255) return false;
        const int dmin = koff[q.koff_off];
        const int dil = KW > 1 ? koff[q.koff_off + 1] - koff[q.koff_off] : 1;
        if (dil < 1 || dil > 255 || dmin > 0 || dmin < p.x_lo) return false;
        for (int k = 0; k < K; k++) if (koff[q.koff_off + k] != (k / KW) * p.x_ld + dmin + (k % KW) * dil) return false;
        if ((KW - 1) * dil > 64) return false;          // (the kernel's staging grid covers BN + 64 columns)
        q.t_tab = KW | (dil << 8); q.t_cin = cin; q.t_rs = 0; q.t_dmin = dmin;
        lds_max = std::max(lds_max, (size_t)(BN + (KW - 1) * dil) * kC32sCS * 4);
        // K order of the kernel: chunk (block * KW + tap) * 2 + group, slot kk of a chunk <- source k = (block * 32 + group * 16 + kk) * KW + tap: the phase's
        // panel in that order comes from the per-model cache (c32s_panel: built once on the device, shared by all plans)
        q.t_rs = KW;          // (scratch until the panels are fetched below: the staged kernel does not read t_rs)
    }
    if (lds_max > 60 * 1024) return false;
    // three-tap layers of the 64- / 128- / 256-row panels at 24 streams and more stay on igemm32_kernel: a channel block is only six chunks there, shorter than
    // the latency of the next block's staging loads that the first weight wait behind them has to sit out (32 / 64 streams: 546 vs 496, 998 vs 919 us for the
    // six 128-row layers, 351 vs 331, 592 vs 564 for the 64-row ones; at 16 streams this kernel wins them too: 294 vs 301, 182 vs 207)
    {
        int kw_max = 0;
        for (const PhaseD &q : phs) kw_max = std::max(kw_max, q.t_tab & 0xff);
        if (mode < 2 && kw_max <= 3 && p.M >= 64 && B >= 24) return false;
    }
    {
        const float *w0 = nullptr;
        for (PhaseD &q : phs) {
            const float *panel = c32s_panel(p.w + q.w_off, p.M, q.nchunks, q.t_cin, q.t_rs);
            if (!w0) w0 = panel;
            q.w_off = panel - w0;          // (device pointers of one flat address space)
            q.t_rs = 0;
        }
        p.w = w0;
    }
    p.koff = nullptr; p.items = nullptr; p.ttab = nullptr;
    p.ph = pl.arena.upload(phs);
    p.nphase = (int)phs.size();
    p.ph0 = phs[0];
    p.ntm = ntm; p.ntn = ntn; p.ksplit = 1; p.nbatch = B; p.m_fast = 0;
    // conv32s_kernel's three ablation branches (timing only: they drop loads, i.e. change results) are reachable in the -DRVC_TUNING build alone; the product
    // always passes 0.  The never-taken branches stay in the kernel because they delimit the scheduler's regions (finding 6).
    p.pad2_ = tune_env("RVC_C32S_DBG") ? atoi(tune_env("RVC_C32S_DBG")) : 0;
    // the 64 x 128 tile takes the buffer-load kernel below 24 streams (us per six launches, all-buffer build against this one: 8 streams 437 vs 474, 16 streams 461 vs 479,
    // 32 streams 842 vs 775, 64 streams 1 543 vs 1 510-1 533); test hook RVC_CONV32S_BUF: 0 never, 2 always
    const int buf_opt = chosen ? ((t_choice.a & 4) ? 2 : 0) : test_opt_int("RVC_CONV32S_BUF", 1);
    const int ltile = tile | ((tile == 1 && (buf_opt == 2 || (buf_opt == 1 && B < 24))) ? 4 : 0);
    const dim3 grid((unsigned)(ntm * ntn), (unsigned)B, (unsigned)p.nphase);
    g_last_wgs = (int)(grid.x * grid.y * grid.z); g_last_waves = 4;
    const double flops = 2.0 * p.M * (double)p.N * ksum * B;
    pl.igemm_flops += flops; pl.n_igemm++;
    Plan *plp = &pl;
    { char d[200]; snprintf(d, sizeof d, "c32s M=%d N=%d K=%d B=%d nph=%d tile=%dx%d%s grid=%ux%ux%u lds=%zu pre=%d ksum=%.0f", p.M, p.N, p.K, B, p.nphase, BM, BN, (ltile & 4) ? "b" : "", grid.x, grid.y, grid.z, lds_max, (int)(p.pre_act != ACT_NONE), ksum); pl.descs.push_back(d); note_kernel(d); }
    const int desc_id = (int)pl.descs.size() - 1;
    const IgemmP pc = p;
    if (final_out) pl.final_out_honoured = true;
    pl.ops.push_back([=](hipStream_t s) {
        ProfEvent *pe = nullptr;
        if (plp->profile) {
            if (plp->prof_used == plp->prof.size()) { ProfEvent e; HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b)); e.flops = 0; e.bytes = 0; plp->prof.push_back(e); }
            pe = &plp->prof[plp->prof_used++]; pe->flops = flops; pe->bytes = 0; pe->desc = desc_id;
        }
        hipEvent_t ea = pe ? pe->a : nullptr, eb = pe ? pe->b : nullptr;
        if (final_out && plp->cur_out) { IgemmP q = pc; q.y = plp->cur_out; q.y_bs = plp->cur_out_bs; launch_conv32s(ltile, q, grid, lds_max, s, ea, eb); }
        else launch_conv32s(ltile, pc, grid, lds_max, s, ea, eb);
    });
    return true;
}

// generic: the caller fills geometry (N, NW, strides, koff, phases); this picks the tile + split-K and queues the op
// Which table-free 1x1 layers take igemm2w_kernel, and with what tile / K split (filled in from per-layer measurements: tests/tools/g2w_sweep.py).
// gt < 0: not this kernel.
// Round 5, tests/tools/g2w_sweep.py (isolated launches, one box; us, planner's previous choice -> this kernel):
//   streams            2              4              8              16             32
//   3072 x  768   20.4 -> 14.7   43.1 -> 27.0   57.5 -> 48.8   94.7 (-> 95.4)  182.6 (-> 173.6)
//   2304 x  768   14.4 -> 11.0   28.8 -> 19.9   43.3 -> 36.2   67.9 -> 64.6    124.3 (-> 129.5)
//    768 x 3072   22.1 -> 16.5   40.7 -> 29.8   59.8 -> 45.2  114.2 -> 86.5    209.8 -> 162.2
//    768 x  768    8.9 ->  7.1   13.7 -> 10.4   20.7 -> 15.2   42.4 -> 28.1     57.0 -> 49.2
//    768 x  512    7.5 ->  6.1   11.3 ->  8.7   16.7 -> 11.9   34.0 -> 20.9     39.4 -> 36.3
// The 32 x 32 wave tile wins everywhere (64 x 32 / 64 x 64 leave too few waves); the K split that wins puts ~4 000-5 000 waves into the launch
// (six waves per workgroup -- 2, 2, 1, 1 over the SIMDs -- always loses to four or eight).  One stream keeps its own kernels (folded LayerNorm,
// latency-tuned splits: 13.5 -> 15.0 us for the 768 x 3072 projection); the tall panels from 16 streams on stay on the LDS-staged 32x32x2 tiles.
static void g2w_rule(const IgemmP &p, int nchunks, int &gt, int &gk)
{
    gt = -1; gk = 1;
    if (!p.fold_n || p.M < 256 || nchunks < 16) return;
    // (round 5, after igemm32l_kernel: from 24 streams -- 2 664 columns -- the 768-row panels run faster on its 64 x 64 / 128 x 64 tiles: step time with this kernel
    //  / without, same process: 20 streams 14.73 / 14.91, 24 streams 16.19 / 15.81, 32 streams 20.16 / 19.70 ms; at 16 streams 11.34 / 11.47-11.60)
    if (p.M > 1024 ? p.N > 1000 : p.N > 2400) return;
    const long long tiles = (long long)((p.M + 31) / 32) * ((p.N + 31) / 32);
    const long long want = (4800 + tiles / 2) / tiles;
    int ks = want <= 1 ? 1 : (want == 2 ? 2 : (want == 3 ? 3 : (want <= 4 ? 4 : 8)));
    while (ks > 1 && nchunks / ks < 4) ks = ks == 8 ? 4 : ks - 1;
    gt = 0; gk = ks;
}

static void queue_igemm_impl(Plan &pl, IgemmP p, int B, const std::vector<int> &koff, const std::vector<PhaseD> &phases, bool final_out)
{
    p.probe = g_kprobe;
    // many streams: fold them into the N axis (one launch-wide column index instead of a grid dimension), so that tiles are cut from
    // B * N columns -- the ContentVec window (N = 111), the text encoder (N = 21) or RMVPE's deep levels (N = 4..64) no longer pad
    // every stream up to a tile.  All offsets stay below 2^31 bytes / elements for every geometry the plugin can ask for (checked).
    const int streams = B;
    if (B > 1) {
        // two to four streams, stride-1 1-D convolution: the staged-tile kernel with the streams in its work-item table (tried before the fold)
        std::vector<PhaseD> phq(phases);
        double ks0 = 0;
        for (PhaseD &q : phq) { if (q.nchunks == 0) q.nchunks = p.K / 16; ks0 += q.nchunks * 16.0; }
        std::stable_sort(phq.begin(), phq.end(), [](const PhaseD &a, const PhaseD &b) { return a.nchunks > b.nchunks; });
        IgemmP pt = p;
        if (queue_conv_tile(pl, pt, B, koff, phq, ks0, final_out)) return;
        // five streams and more: the staged 32x32x2 convolution (streams as a grid dimension: also before the fold)
        pt = p;
        if (queue_conv32s(pl, pt, B, koff, phq, ks0, final_out)) return;
        // (round 6: the same structure for RMVPE's Conv2d 3x3 layers -- conv2d32s_kernel, the padded planes as flat 1-D rows -- was built, parity-green and
        //  SLOWER than the register-direct kernel at 16 / 64 / 128 streams, 41-54 against 60-67 TF/s: the layers are 0.6-1.2 GFLOP with K = 288-576, a tile's
        //  K loop is 18-36 chunks behind a 50 KB staging prologue.  Not in the tree; DESIGN.md section 7 round 6, profiles/r06_conv2d32s_*.txt)
    }
    if (B == 1 && test_opt_int("RVC_CONV32S", 1) == 2) {           // test hook: the kernel forced onto one stream
        std::vector<PhaseD> phq(phases);
        double ks0 = 0;
        for (PhaseD &q : phq) { if (q.nchunks == 0) q.nchunks = p.K / 16; ks0 += q.nchunks * 16.0; }
        IgemmP pt = p;
        if (queue_conv32s(pl, pt, B, koff, phq, ks0, final_out)) return;
    }
    if (B > 1 && !tune_env("RVC_NO_FOLD")) {
        const long long lim = (1LL << 29);
        if ((long long)B * p.x_bs < lim && (long long)B * p.y_bs < lim && (long long)B * (p.res ? p.res_bs : 0) < lim && (long long)B * p.N < (1LL << 30) &&
            (size_t)(p.K / 16) * 64 <= 60 * 1024) {      // (the two-stage grid split-K fallback keeps the batch as a grid dimension)
            p.fold_n = p.N; p.N = B * p.N; B = 1;
        }
    }
    (void)streams;
    // table entries become non-negative byte offsets; the kernel moves the base pointer back by koff_bias bytes
    std::vector<int> kb(koff);
    int kmin = 0;
    for (int v : kb) kmin = std::min(kmin, v);
    for (int &v : kb) v = (v - kmin) * 4;
    p.koff_bias = -kmin * 4;
    const bool pre = p.pre_act != ACT_NONE;
    {
        auto it = pl.koff_tabs.find(kb);
        if (it == pl.koff_tabs.end()) it = pl.koff_tabs.emplace(kb, pl.arena.upload(kb)).first;
        p.koff = it->second;
    }
    std::vector<PhaseD> phv(phases);
    double ksum = 0;   // sum of the phases' K (phases of a fused launch may differ; p.K is the maximum)
    for (PhaseD &q : phv) { if (q.nchunks == 0) q.nchunks = p.K / 16; ksum += q.nchunks * 16.0; }
    // phases of unequal length (the fused ResBlock chains: kernel sizes 3 / 7 / 11) are dispatched longest first: the grid's z axis
    // is walked last, so the workgroups of phase 0 start first and the short phases fill the tail instead of the long one forming it
    if (!tune_env("RVC_NO_LPT"))
        std::stable_sort(phv.begin(), phv.end(), [](const PhaseD &a, const PhaseD &b) { return a.nchunks > b.nchunks; });
    p.ph = pl.arena.upload(phv);
    p.nphase = (int)phv.size();
    p.ph0 = phv[0];
    const int nchunks = p.K / 16;
    auto tiles = [&](int c) {
        long long tm = (p.M + 16 * kMF[c] - 1) / (16 * kMF[c]), tn = (p.N + 16 * kNF[c] - 1) / (16 * kNF[c]);
        return tm * tn * B * p.nphase;
    };
    // Pick the largest tile that still yields >= 1024 waves (one per SIMD), using the in-workgroup K split
    // (KS = 4/8/16 waves per tile) when the layer has too few tiles.  A wave keeps >= 4 chunks of K.
    const int order_big[3] = {4, 3, 0}, order_small[3] = {2, 1, 0};
    // a panel whose 32-row tiling would be >= 25 % padding (48 rows: the grouped positional convolution) takes the 16-row tiles
    // (measured at one stream: 16 x 32, K split 8: 25 us against 37 us for the 32 x 32 tile the size rule picked)
    const bool pad32 = p.M > 16 && (((p.M + 31) / 32 * 32 - p.M) * 4 >= p.M);
    const int *order = (p.M > 16 && !pad32) ? order_big : order_small;
    int cfg = 0, wg_ks = 1;
    long long best_waves = -1;
    bool found = false;
    // phases of unequal length (fused ResBlock chains, kernel sizes 3/7/11) are all co-resident: finer tiles even out the
    // per-SIMD load (measured on the decoder: 32x32 tiles 185 vs 200 us at C = 128, 127 vs 133 us at C = 64; folding the
    // chains' average into one K-concatenated GEMM was also measured: no gain)
    bool uneven = false;
    for (const PhaseD &q : phv) uneven = uneven || q.nchunks != phv[0].nchunks;
    long long want_waves = (uneven && p.M >= 64) ? 2048 : 1024;
    if (const char *f = tune_env("RVC_WANT_WAVES")) { if (p.fold_n) want_waves = atoll(f); }      // tuning aid
    for (int oi = 0; oi < 3 && !found; oi++) {
        const int c = order[oi];
        for (int ks = 1; ks <= 16; ks = ks == 1 ? 4 : ks * 2) {
            if (ks > 1 && (nchunks / ks < 4 || ks * kMF[c] * kNF[c] > 32)) break;
            if ((size_t)nchunks * 64 + (ks > 1 ? (size_t)ks * kMF[c] * kNF[c] * 1024 : 0) > 60 * 1024) break;
            const long long w = tiles(c) * ks;
            if (w > best_waves) { best_waves = w; cfg = c; wg_ks = ks; }
            // streams folded into N: the workgroups must also spread evenly over the CUs (768 x 3072 at 8 streams: 336 workgroups of 32 x 64 tiles
            // are one or two per CU, 52 TF/s; 672 of 32 x 32 tiles 68 TF/s).  Below four rounds a last round under 80 % full sends the choice on
            // to the next smaller tile.  (One stream keeps its own, latency-tuned rule.)
            if (p.fold_n && oi == 0 && !tune_env("RVC_NO_BALANCE")) {         // (one step down only: the 16 x 16 tile loses more than an uneven last round costs)
                const long long wgs = ks > 1 ? tiles(c) : (tiles(c) + 3) / 4, rounds = (wgs + g_ncu - 1) / g_ncu;
                if (rounds < 4 && wgs * 5 < rounds * g_ncu * 4) continue;
            }
            if (w >= want_waves) { cfg = c; wg_ks = ks; found = true; break; }
        }
    }
    // throughput mode (many streams): workgroup-tiled kernel with the activation tile shared through LDS
    int lds_cfg = -1;
    bool phase_epi = false;                               // per-phase activation / output tensor: igemm2 only
    for (const PhaseD &q : phv) phase_epi = phase_epi || q.act_p1 != 0 || q.y_off != 0;
    if (queue_conv_tile(pl, p, B, koff, phv, ksum, final_out)) return;
    const bool ln_fold = p.ln_wsum || p.ln_stats_in || phase_epi;      // folded LayerNorm lives in the register-direct kernel's K-split epilogue
    if (!ln_fold && !tune_env("RVC_NO_LDS_GEMM") && nchunks >= 2 && (size_t)nchunks * 64 + 2 * 256 * 20 * 4 <= 60 * 1024) {
        int bm = p.M >= 96 ? 128 : (p.M >= 48 ? 64 : (p.M > 16 ? 32 : 0));
        if (const char *f = tune_env("RVC_G32_BM")) { const int v = atoi(f); if (v == 32 || v == 64 || v == 128) bm = v; }   // tuning aid
        const int bn = bm == 128 ? 128 : 256;
        if (bm) {
            const long long wgs = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * B * p.nphase;
            // isolated B = 64 timings (tests/tools/gemm_microbench.py): the LDS-tiled kernel wins for very tall (M >= 2048) and very short
            // (M <= 64) weight panels, the register-direct kernel in between (cv_ff2 91 vs 73 TF/s, cv_o 77 vs 62, enc_ff1 24 vs 13)
            const bool lds_wins = p.M >= 2048 || p.M <= 64;
            if (wgs >= 384 && lds_wins) lds_cfg = bm == 128 ? 0 : (bm == 64 ? 1 : 2);
            // 32x32x2 kernel (igemm32): RVC_GEMM32 = 0 off, 1 wherever the old workgroup-tiled kernel was chosen, 2 (default) for every
            // layer with enough workgroups to fill the chip
            static const int g32 = tune_env("RVC_GEMM32") ? atoi(tune_env("RVC_GEMM32")) : 2;
            static const long long g32_min = tune_env("RVC_GEMM32_MIN") ? atoll(tune_env("RVC_GEMM32_MIN")) : 768;   // fewer workgroups balance badly over 256 CUs (measured: 336 -> slower)
            if (g32 == 1 && lds_cfg >= 0 && !p.glu) lds_cfg += 3;
            else if (g32 >= 2 && wgs >= g32_min && !p.glu) lds_cfg = 3 + (bm == 128 ? 0 : (bm == 64 ? 1 : 2));   // (gated layers stay on the kernels that are tested with the gate)
            // ... and from 500 workgroups wherever the 16x16x4 LDS kernel would run (tall panels at 32 streams: 2133 -> 2081, 1470 -> 1454 us)
            else if (g32 >= 2 && lds_cfg >= 0 && wgs >= 500 && !p.glu && !tune_env("RVC_NO_G32_SHORT")) lds_cfg = 3 + (bm == 128 ? 0 : (bm == 64 ? 1 : 2));
            else if (g32 >= 2 && !p.glu && p.M <= 64 && p.M > 16 && !tune_env("RVC_NO_G32_SHORT")) {
                // short weight panels below that count (the decoder's 64- and 32-channel stages at 16-32 streams): 32-row tiles double the
                // workgroups of a 64-channel layer, and from ~600 of them the 32x32x2 kernel beats both the register-direct kernel and the
                // 16x16x4 LDS kernel (round 4, 16 streams, per layer: M = 64: 807 / 542 / 307 -> 606 / 408 / 210 us; M = 32: 354 / 247 / 143 ->
                // 330 / 232 / 132 us; with 64-row tiles -- 315 workgroups -- 688 / 460 / 236)
                const long long wgs32 = (long long)((p.M + 31) / 32) * ((p.N + 255) / 256) * B * p.nphase;
                if (wgs32 >= 600) lds_cfg = 3 + 2;
            }
        }
    }
    // 48-row panels (ContentVec's grouped positional convolution: 16 groups of 48 channels, K = 6144 each): three 16-row fragments
    // exactly, instead of a 64-row tile with a quarter of its MFMAs on padding
    if (lds_cfg == 1 && p.M == 48 && !tune_env("RVC_NO_BM48")) lds_cfg = 6;
    // mid-size panels (M = 768 at 64 streams: 336 tiles of 128 x 128 balance badly over 256 CUs, and the register-direct 2 x 4 tile runs
    // at two waves per SIMD): 128 x 64 tiles of the 32x32x2 kernel, four waves stacked in M over one 64-column activation tile
    // (768 x 3072 projection at 64 streams: 361 -> 342 us; small, but the same kernel)
    // (round 4, per layer at 8 / 32 streams: from 250 workgroups for panels of >= 512 rows -- 3072 x 768 at 8 streams 886 -> 800 us, 768 x 3072 at 32
    // streams 3006 -> 2774, 768 x 768 883 -> 771 --; a 256-row panel with 252 workgroups loses to the register-direct kernel, 731 -> 1103)
    static const long long g32_narrow_env = tune_env("RVC_G32_NARROW") ? atoll(tune_env("RVC_G32_NARROW")) : -1;     // 0 = off
    const long long g32_narrow_min = g32_narrow_env >= 0 ? g32_narrow_env : (p.M >= 512 ? 250 : 500);
    if (lds_cfg < 0 && !ln_fold && g32_narrow_min > 0 && !tune_env("RVC_NO_LDS_GEMM") && !p.glu && nchunks >= 2 && p.M >= 96 && (size_t)nchunks * 64 + 2 * 64 * 20 * 4 <= 60 * 1024) {
        const long long wgs = (long long)((p.M + 127) / 128) * ((p.N + 63) / 64) * B * p.nphase;
        if (wgs >= g32_narrow_min) lds_cfg = 7;
        // between one and two workgroups per CU the 128 x 64 tile leaves half of the slots empty: 64 x 64 tiles (one 32 x 32 accumulator per wave) double them
        // (round 4, same box: 3072 x 768 / 2304 x 768 at 8 streams 803 / 660 -> 692 / 575 us, 768 x 3072 / 768 x 768 at 32 streams 2761 / 771 -> 2413 / 675 us)
        // (round 5: only the table-free 1x1 layers, where that was measured.  The 512-row stem convolutions with ~7 160 columns -- 448 workgroups of
        //  128 x 64 -- run 119-120 us on the 128 x 64 tile against 148-152 on 64 x 64, isolated, at 4 and 16 streams: tests/tools/tile_sweep.py)
        if (lds_cfg == 7 && wgs < 500 && p.lin_cs4 != 0 && !tune_env("RVC_NO_G32_SQ64")) lds_cfg = 8;
    }
    // exploratory split-bf16 GEMM (rvc_set_gemm_precision(e, 1); never the default): every 1-D layer the 32x32x2 kernel could take with >= 128 rows and
    // >= 250 workgroups of 128 x 128 (below that the fp32 kernels with their finer tiles win -- 16 streams, 768-row panels, 84 workgroups: 162 vs 86 us)
    if ((p.bf3 || pl.bf3) && !p.glu && !ln_fold && B == 1 && p.x_hs == 0 && p.y_hm == 0 && p.M >= 128 && nchunks >= 2 && !p.accumulate &&
        (size_t)nchunks * 64 + 2 * 2 * 128 * 48 <= 60 * 1024 &&
        (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.nphase >= 250) {
        const bool lin = p.lin_cs4 != 0 && p.nphase == 1 && !pre;
        const int nblk = (p.M + 31) / 32;
        // every phase's fp32 fragment panel -> its split panels ([32-row block][chunk][hi | lo][lane][8 bf16]); phases keep their own K
        size_t tot = 0;
        std::vector<size_t> off(phv.size());
        for (size_t f = 0; f < phv.size(); f++) { off[f] = tot; tot += (size_t)nblk * phv[f].nchunks * 2048; }
        float *wsplit = (float *)wmalloc_plan(tot);
        for (size_t f = 0; f < phv.size(); f++) bf3_pack(p.w + phv[f].w_off, p.M, phv[f].nchunks, (char *)wsplit + off[f], nullptr);
        HIPCHK(hipDeviceSynchronize());
        pl.owned_dev.push_back(wsplit);
        std::vector<PhaseD> ph3(phv);
        for (size_t f = 0; f < ph3.size(); f++) ph3[f].w_off = (long long)(off[f] / 4);
        p.w = wsplit; p.ph = pl.arena.upload(ph3); p.ph0 = ph3[0];
        p.ksplit = 1; p.chunks_per_split = nchunks;
        p.ntm = (p.M + 127) / 128; p.ntn = (p.N + 127) / 128;
        p.m_fast = p.fold_n ? 1 : 0;
        const dim3 grid((unsigned)(p.ntm * p.ntn), (unsigned)p.nphase);
        const size_t lds = (lin ? 0 : (size_t)nchunks * 64) + (size_t)2 * 2 * 128 * 48;
        g_last_wgs = (int)(grid.x * grid.y); g_last_waves = 4;
        const double flops = 2.0 * p.M * (double)p.N * ksum;
        pl.igemm_flops += flops; pl.n_igemm++;
        Plan *plp = &pl;
        { char d[176]; snprintf(d, sizeof d, "bf3 M=%d N=%d K=%d B=1 nph=%d tile=128x128 grid=%ux%u lin=%d pre=%d", p.M, p.N, p.K, p.nphase, grid.x, grid.y, (int)lin, (int)pre); pl.descs.push_back(d); note_kernel(d); }
        const int desc_id = (int)pl.descs.size() - 1;
        if (final_out) pl.final_out_honoured = true;
        pl.ops.push_back([=](hipStream_t s) {
            ProfEvent *pe = nullptr;
            if (plp->profile) {
                if (plp->prof_used == plp->prof.size()) { ProfEvent e; HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b)); e.flops = 0; e.bytes = 0; plp->prof.push_back(e); }
                pe = &plp->prof[plp->prof_used++]; pe->flops = flops; pe->bytes = 0; pe->desc = desc_id;
            }
            hipEvent_t ea = pe ? pe->a : nullptr, eb = pe ? pe->b : nullptr;
            if (final_out && plp->cur_out) { IgemmP q = p; q.y = plp->cur_out; q.y_bs = plp->cur_out_bs; launch_igemm_bf3(lin, pre, q, grid, lds, s, ea, eb); }
            else launch_igemm_bf3(lin, pre, p, grid, lds, s, ea, eb);
        });
        return;
    }
    // igemm2w_kernel: register-direct 32x32x2 tiles for the table-free 1x1 layers at a few streams (igemm.hip.h).  Test hook RVC_FORCE_G2W = "tile,ks"
    // (tile 0 = 32 x 32 per wave, 1 = 64 x 32, 2 = 64 x 64; ks = 1 / 2 / 3 / 4 / 6 / 8 waves splitting K) forces it wherever it is eligible.
    {
        // (round 6) a layer that consumes a not yet normalised tensor (IgemmP::ln_wsum) can take the kernel's LayerNorm-consumer variant: 32 x 32 wave tile, four or
        // eight K shares.  One stream, tests/tools/g2w_sweep.py: the 2304- / 3072-row projections 12.4 / 13.4 -> 9.5 / 10.0 us against igemm2_kernel's LNB tiles;
        // the 768-row layers (output projection, second FFN layer, feature projection) stay: 6.0 / 13.5 / 5.0 vs 6.3 / 15.0 / 5.4.  Test hook RVC_G2W_LN = 0: never.
        const bool g2w_ln = p.ln_wsum && !p.ln_stats_in && !phase_epi;
        const bool g2w_ok = p.lin_cs4 != 0 && p.nphase == 1 && !pre && !p.glu && (!ln_fold || g2w_ln) && B == 1 && nchunks >= 1;
        int gt = -1, gk = 1;
        if (t_choice.kind == 2) { gt = t_choice.a; gk = t_choice.b; }
        else if (t_choice.kind != 0) gt = -1;
        else if (const char *f = test_opt("RVC_FORCE_G2W")) { if (sscanf(f, "%d,%d", &gt, &gk) < 1) gt = -1; }
        else if (g2w_ln) { if (streams == 1 && p.M >= 2048 && nchunks >= 32 && test_opt_int("RVC_G2W_LN", 1) != 0) { gt = 0; gk = 8; } }
        else g2w_rule(p, nchunks, gt, gk);
        if (g2w_ln && (gt != 0 || (gk != 4 && gk != 8) || nchunks < gk)) gt = -1;          // (only those two instantiations exist)
        if (g2w_ok && gt >= 0 && gt <= 2) {
            if (!(gt == 0 && (gk == 12 || gk == 16)) && gk != 1 && gk != 2 && gk != 3 && gk != 4 && gk != 6 && gk != 8) gk = gk > 8 ? 8 : 4;
            while (gk > 1 && nchunks < gk) gk = gk == 16 ? 12 : (gk == 12 ? 8 : (gk == 8 ? 6 : (gk == 6 ? 4 : gk - 1)));
            const int bm = 32 * kG2wMT[gt], bn = 32 * kG2wNT[gt];
            p.ksplit = 1; p.chunks_per_split = nchunks;
            p.ntm = (p.M + bm - 1) / bm; p.ntn = (p.N + bn - 1) / bn;
            const bool wh = p.ntm >= 2;           // all column tiles of one weight-row block on one XCD (its L2 is private)
            p.m_fast = wh ? (p.ntm + 7) / 8 * 8 : 0;
            unsigned gx = (unsigned)(wh ? p.ntm : p.ntn);
            if (wh && gx >= 8) gx = (gx + 7) / 8 * 8;
            const dim3 grid(gx, (unsigned)(wh ? p.ntn : p.ntm), 1);
            if (grid.y > 65535) throw ShapeError("implicit GEMM grid too large");
            p.nbatch = 1;
            const size_t lds = gk > 1 ? (size_t)gk * kG2wMT[gt] * kG2wNT[gt] * 1024 * sizeof(float) + (g2w_ln ? (size_t)gk * 32 * 2 * sizeof(float) : 0) : 0;
            g_last_wgs = (int)(grid.x * grid.y); g_last_waves = gk;
            const double flops = 2.0 * p.M * (double)p.N * ksum;
            pl.igemm_flops += flops; pl.n_igemm++;
            Plan *plp = &pl;
            { char d[176]; snprintf(d, sizeof d, "g2w M=%d N=%d K=%d B=1 nph=1 tile=%dx%d ks=%d grid=%ux%u%s", p.M, p.N, p.K, bm, bn, gk, grid.x, grid.y, g2w_ln ? " ln=1" : ""); pl.descs.push_back(d); note_kernel(d); }
            const int desc_id = (int)pl.descs.size() - 1;
            if (final_out) pl.final_out_honoured = true;
            pl.ops.push_back([=](hipStream_t s) {
                ProfEvent *pe = nullptr;
                if (plp->profile) {
                    if (plp->prof_used == plp->prof.size()) { ProfEvent e; HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b)); e.flops = 0; e.bytes = 0; plp->prof.push_back(e); }
                    pe = &plp->prof[plp->prof_used++]; pe->flops = flops; pe->bytes = 0; pe->desc = desc_id;
                }
                hipEvent_t ea = pe ? pe->a : nullptr, eb = pe ? pe->b : nullptr;
                if (g2w_ln) launch_igemm2w_ln(gk, p, grid, lds, s, ea, eb);
                else if (final_out && plp->cur_out) { IgemmP q = p; q.y = plp->cur_out; q.y_bs = plp->cur_out_bs; launch_igemm2w(gt, gk, q, grid, lds, s, ea, eb); }
                else launch_igemm2w(gt, gk, p, grid, lds, s, ea, eb);
            });
            return;
        }
    }
    // tuning aid: RVC_G32W = "lc" forces one of the 32x32x2 tiles (3 4 5 7 8) on every layer that kernel can take, "-1" the register-direct kernel
    // (then RVC_FORCE_CFG picks its tile): tests/tools/tile_sweep.py
    if (const char *f = tune_env("RVC_G32W")) {
        const int wl = atoi(f);
        const bool g32_ok = !ln_fold && !p.glu && nchunks >= 2 && (size_t)nchunks * 64 + 2 * 256 * 20 * 4 <= 60 * 1024;
        if (g32_ok && (wl == 3 || wl == 4 || wl == 5 || wl == 7 || wl == 8)) lds_cfg = wl;
        else if (wl == -1) lds_cfg = -1;
    }
    if (t_choice.kind == 3) {
        const int wl = t_choice.a;
        const bool g32_ok = !ln_fold && !p.glu && nchunks >= 2 && (size_t)nchunks * 64 + 2 * 256 * 20 * 4 <= 60 * 1024;
        const int need_m = (wl == 3 || wl == 7) ? 96 : ((wl == 4 || wl == 8) ? 48 : 17);
        if (g32_ok && (wl == 3 || wl == 4 || wl == 5 || wl == 7 || wl == 8) && p.M >= need_m) lds_cfg = wl;
    } else if (t_choice.kind == 4) lds_cfg = -1;
    // igemm32l_kernel (one-phase 1-D layers, buffer loads with scalar offsets): its 128 x 64 tile beats the 128 x 128 tile of either kernel on every layer it can
    // take -- the 3072-row projection (64 streams 36.12 -> 35.81 ms), the 2304-row one (35.31 -> 35.05), the strided stem (35.28 -> 35.05; 16 / 32 streams
    // 11.35 / 19.70 -> 11.23 / 19.50) -- so those layers move there (test hook RVC_G32L_TALL = 0: keep the 128 x 128 tile)
    const bool g32l_on = p.lin_cs4 != 0 && p.nphase == 1 && !pre && !p.bf3 && !pl.bf3 && test_opt_int("RVC_G32L", 1) != 0;
    const bool g32t_on = !g32l_on && p.lin_cs4 == 0 && p.nphase == 1 && B == 1 && p.x_hs == 0 && p.y_hm == 0 && !p.bf3 && !pl.bf3 && !p.glu &&
                         test_opt_int("RVC_G32L", 1) != 0 && test_opt_int("RVC_G32L_TAB", 1) != 0;
    if ((g32l_on || (g32t_on && !pre)) && lds_cfg == 3 && t_choice.kind != 3 && test_opt_int("RVC_G32L_TALL", 1) != 0) lds_cfg = 7;
    if (lds_cfg >= 0) {
        const int bm = lds_cfg == 8 ? 64 : (lds_cfg == 7 ? 128 : (lds_cfg == 6 ? 48 : (lds_cfg % 3 == 0 ? 128 : (lds_cfg % 3 == 1 ? 64 : 32))));
        const int bn = (lds_cfg == 7 || lds_cfg == 8) ? 64 : ((lds_cfg != 6 && lds_cfg % 3 == 0) ? 128 : 256);
        p.ksplit = 1; p.chunks_per_split = nchunks;
        p.ntm = (p.M + bm - 1) / bm; p.ntn = (p.N + bn - 1) / bn;
        // tile order of the tiled kernels when the streams are folded into N: m fastest over XCD-local tile ids (1), or over the raw block index (2)
        // where that keeps a large weight matrix partitioned over the XCDs (igemm.hip.h, xcd_tile_id)
        p.m_fast = p.fold_n ? ((p.ntm % 8 == 0 && (size_t)p.M * (size_t)ksum * sizeof(float) > ((size_t)4 << 20)) ? 2 : 1) : 0;
        dim3 grid(p.ntm * p.ntn, B * p.nphase);
        const bool g32k = lds_cfg >= 3 && lds_cfg != 6;            // igemm32_kernel keeps its activation tile column-major, [2][bn][20]
        const size_t lds = (size_t)nchunks * 64 + (g32k ? (size_t)2 * bn * 20 * 4 : (size_t)2 * 16 * (bn + 4) * 4);
        g_last_wgs = (int)(grid.x * grid.y); g_last_waves = 4;
        const double flops = 2.0 * p.M * (double)p.N * ksum * B;
        pl.igemm_flops += flops; pl.n_igemm++;
        Plan *plp = &pl;
        const int lc = lds_cfg;
        // table-free 1x1 layers on tiles 3 / 7 / 8: igemm32l_kernel (buffer loads with scalar row offsets, no offset table in LDS); test hook RVC_G32L = 0: off
        // ... and the one-phase 1-D layers WITH a table (the strided stem of ContentVec, three-tap decoder layers): the same kernel with the table entries as
        // scalar loads (test hook RVC_G32L_TAB = 0: off)
        const bool g32t = (lc == 3 || lc == 7 || lc == 8) && g32t_on &&
                          !(lc == 3 && pre);           // (the decoder's three-tap 128-row layers with the fused input activation: 167 vs 152 us on the 128 x 128 tile)
        const bool g32l = ((lc == 3 || lc == 7 || lc == 8) && g32l_on && !(lc == 3 && p.m_fast == 2)) || g32t;
        const int g32l_mode = g32t ? (pre ? 2 : 1) : 0;
        const size_t lds_l = (size_t)2 * bn * 20 * 4;
        // panel order inside the XCDs (igemm32l.hip.h, m_fast = 3) for tall table-free panels whose weights exceed an L2: mp m-tiles = the largest panel of
        // <= 2.5 MB; test hook RVC_G32L_PANEL = 0: the orders of round 4 (m fastest over XCD-local ids / the raw block index)
        if (g32l && !g32t && p.fold_n && B == 1 && (lc == 7 || lc == 8) && p.ntm >= 8 && p.ntn >= 64 && test_opt_int("RVC_G32L_PANEL", 1) != 0) {          // (from ~37 streams: at 16 streams -- 28 n-tiles, 3.5 per XCD -- the padded grid costs 2.4 %)
            const size_t per_tile = (size_t)bm * (size_t)ksum * sizeof(float);
            const int mp = (int)std::max<size_t>(1, ((size_t)5 << 19) / per_tile);
            if ((size_t)p.M * (size_t)ksum * sizeof(float) > ((size_t)5 << 19) && mp < p.ntm) {
                p.m_fast = 3; p.pad2_ = mp;
                const int nx_max = (p.ntn + 7) / 8;
                grid = dim3((unsigned)(8 * nx_max * p.ntm), 1u);
            }
        }
        { char d[160]; snprintf(d, sizeof d, "%s M=%d N=%d K=%d B=%d nph=%d tile=%dx%d grid=%ux%u", g32t ? "g32t" : g32l ? "g32l" : (lds_cfg >= 3 && lds_cfg != 6) ? "g32" : "lds", p.M, p.N, p.K, B, p.nphase, bm, bn, grid.x, grid.y); pl.descs.push_back(d); note_kernel(d); }
        const int desc_id = (int)pl.descs.size() - 1;
        if (final_out) pl.final_out_honoured = true;
        pl.ops.push_back([=](hipStream_t s) {
            ProfEvent *pe = nullptr;
            if (plp->profile) {
                if (plp->prof_used == plp->prof.size()) { ProfEvent e; HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b)); e.flops = 0; e.bytes = 0; plp->prof.push_back(e); }
                pe = &plp->prof[plp->prof_used++]; pe->flops = flops; pe->bytes = 0; pe->desc = desc_id;
            }
            hipEvent_t ea = pe ? pe->a : nullptr, eb = pe ? pe->b : nullptr;
            if (g32l) {
                if (final_out && plp->cur_out) { IgemmP q = p; q.y = plp->cur_out; q.y_bs = plp->cur_out_bs; launch_igemm32l(lc, g32l_mode, q, grid, lds_l, s, ea, eb); }
                else launch_igemm32l(lc, g32l_mode, p, grid, lds_l, s, ea, eb);
                return;
            }
            if (final_out && plp->cur_out) { IgemmP q = p; q.y = plp->cur_out; q.y_bs = plp->cur_out_bs; launch_igemm_tiled(lc, pre, q, grid, lds, s, ea, eb); }
            else launch_igemm_tiled(lc, pre, p, grid, lds, s, ea, eb);
        });
        return;
    }
    // one stream, table-free layers of the ContentVec window (N = 111) that the size rule sends to lone 16 x 16 fragments: every B fragment costs
    // four dword gathers (9-12 clocks each on the CU's single vector-memory path) for ONE MFMA row block; two fragments along N per wave and eight
    // K shares halve the weight loads per MFMA (isolated: 768 x 3072 18.5 -> 14.9 us, 768 x 768 6.5 -> 5.7 us; in the chain: ContentVec -22 us)
    if (cfg == 0 && wg_ks == 4 && B == 1 && !p.fold_n && p.lin_cs4 && p.nphase == 1 && p.M >= 256 && p.N > 64 && p.N <= 128 && nchunks >= 32 && !p.ln_wsum && !tune_env("RVC_NO_LIN_16x32")) { cfg = 1; wg_ks = 8; }
    if (const char *f = tune_env("RVC_TUNE")) {        // tuning aid: "M,K:cfg,ks;M,K:cfg,ks;..." overrides the tile choice of matching layers
        for (const char *q = f; q && *q; ) {
            int tm = 0, tk = 0, tc = 0, tks = 1;
            if (sscanf(q, "%d,%d:%d,%d", &tm, &tk, &tc, &tks) == 4 && tm == p.M && tk == p.K) { cfg = tc; wg_ks = tks; }
            q = strchr(q, ';'); if (q) q++;
        }
    }
    if (const char *f = test_opt("RVC_FORCE_CFG")) {   // tuning aid: "cfg,ks[,mfast]"
        int fc = 0, fk = 1; if (sscanf(f, "%d,%d", &fc, &fk) >= 1) { cfg = fc; wg_ks = fk; }
    }
    if (t_choice.kind == 4 && !(p.ln_wsum || p.ln_stats_in)) {
        const int fc = t_choice.a, fk = t_choice.b;
        const bool ok = fc >= 0 && fc <= 4 && (fk == 1 || ((fk == 4 || fk == 8 || fk == 16) && nchunks / fk >= 4 && fk * kMF[fc] * kNF[fc] <= 32)) &&
                        (size_t)nchunks * 64 + (fk > 1 ? (size_t)fk * kMF[fc] * kNF[fc] * 1024 : 0) <= 60 * 1024;
        if (ok) { cfg = fc; wg_ks = fk; }
    }
    if (p.ln_wsum || p.ln_stats_in) {
        // folded LayerNorm: one stream or a few folded into N (statistics are per launch column), in-workgroup K split (the statistics / the
        // normalised residual live in that epilogue)
        if (lds_cfg >= 0 || B != 1 || p.nphase != 1) throw std::logic_error("folded LayerNorm outside its supported launch shape");
        if (wg_ks == 1) {
            wg_ks = 4;
            while (cfg > 0 && (nchunks / wg_ks < 4 || wg_ks * kMF[cfg] * kNF[cfg] > 32)) cfg = cfg == 4 ? 3 : (cfg == 3 ? 1 : 0);
        }
        if (p.ln_wsum && (p.lin_cs4 == 0 || pre || nchunks / wg_ks < 1)) throw std::logic_error("LayerNorm consumer must be a table-free 1x1 layer");
    }
    int ksplit = 1;
    if ((size_t)nchunks * 64 > 60 * 1024 && (p.glu || phase_epi)) throw ShapeError("fused conv too long for the in-workgroup K split");
    if ((size_t)nchunks * 64 > 60 * 1024) {     // koff slice would not fit in LDS: grid-level split (two-stage, rare)
        ksplit = (int)(((size_t)nchunks * 64 + 60 * 1024 - 1) / (60 * 1024));
        cfg = 0; wg_ks = 1;
    }
    int cps = (nchunks + ksplit - 1) / ksplit;
    ksplit = (nchunks + cps - 1) / cps;
    p.ksplit = ksplit; p.chunks_per_split = cps;
    p.ntm = (p.M + 16 * kMF[cfg] - 1) / (16 * kMF[cfg]);
    p.ntn = (p.N + 16 * kNF[cfg] - 1) / (16 * kNF[cfg]);
    if (ksplit > 1) p.part = pl.arena.floats((size_t)B * p.nphase * ksplit * p.M * p.N);
    // weight-heavy layers (short N: the transformer at T=111, RMVPE's deep levels, the synth encoder): keep all tiles that
    // read the same weight rows on one XCD so each weight byte crosses the fabric once (per-XCD L2s are private)
    bool weight_heavy = (p.N <= 512 && (long long)p.M * p.K >= 64 * 1024 && p.ntm >= 8) || (p.fold_n && p.ntm >= 2);
    if (const char *f = tune_env("RVC_FORCE_MFAST")) weight_heavy = atoi(f) != 0;
    p.m_fast = weight_heavy ? (p.ntm + 7) / 8 * 8 : 0;
    const int ntiles = weight_heavy ? p.m_fast * p.ntn : p.ntm * p.ntn;
    dim3 grid(wg_ks > 1 ? ntiles : (ntiles + 3) / 4, B * p.nphase * ksplit);
    dim3 egrid((unsigned)(((long long)p.M * p.N + 255) / 256), B * p.nphase);
    // lean kernel: x = fast tile axis (m when m_fast, else n; 4 tiles per workgroup without the in-workgroup K split), y = slow axis
    const bool lean = ksplit == 1 && !tune_env("RVC_OLD_IGEMM");
    const bool lin = lean && p.lin_cs4 != 0 && p.nphase == 1 && !pre && !tune_env("RVC_NO_LIN");
    size_t lds2 = 0;
    if (lean) {
        const int fast_n = weight_heavy ? p.ntm : p.ntn, slow_n = weight_heavy ? p.ntn : p.ntm;
        unsigned gx = (unsigned)(wg_ks > 1 ? fast_n : (fast_n + 3) / 4);
        // workgroup (x, y) runs on XCD x % 8 when gridDim.x is a multiple of 8: all tiles of one weight-row block then share one
        // XCD's L2.  Only when the padding is cheap and every XCD still gets live workgroups (a short axis padded to 8 would park
        // all the work on a few XCDs: measured 3.6x slower at 64 streams)
        if (weight_heavy && ((wg_ks > 1 && gx >= 8) || gx >= 16)) gx = (gx + 7) / 8 * 8;
        grid = dim3(gx, (unsigned)slow_n, (unsigned)(B * p.nphase));
        if (grid.y > 65535 || grid.z > 65535) throw ShapeError("implicit GEMM grid too large");
        p.nbatch = B;
        lds2 = (lin ? 0 : (size_t)nchunks * 64) + (wg_ks > 1 ? (size_t)wg_ks * kMF[cfg] * kNF[cfg] * 1024 : 0) + (p.ln_wsum ? (size_t)wg_ks * kNF[cfg] * 16 * 2 * 4 : 0);
        if (p.ln_wsum && !lin) throw std::logic_error("LayerNorm consumer did not get the table-free kernel");
    }
    g_last_wgs = (int)(grid.x * grid.y * grid.z); g_last_waves = wg_ks > 1 ? wg_ks : 4;
    const double flops = 2.0 * p.M * (double)p.N * ksum * B;
    pl.igemm_flops += flops;
    pl.n_igemm++;
    Plan *plp = &pl;
    { char d[200]; snprintf(d, sizeof d, "reg M=%d N=%d K=%d B=%d nph=%d tile=%dx%d ks=%d mfast=%d grid=%ux%ux%u pre=%d lin=%d ksum=%.0f", p.M, p.N, p.K, B, p.nphase, 16 * kMF[cfg], 16 * kNF[cfg], wg_ks, p.m_fast, grid.x, grid.y, grid.z, (int)pre, (int)lin, ksum); pl.descs.push_back(d); note_kernel(d); }
    const int desc_id = (int)pl.descs.size() - 1;
    if (final_out && lean) pl.final_out_honoured = true;      // (the two-stage grid split-K fallback writes through a second kernel: it keeps the plan's own tensor)
    pl.ops.push_back([=](hipStream_t s) {
        ProfEvent *pe = nullptr;
        if (plp->profile) {
            if (plp->prof_used == plp->prof.size()) {
                ProfEvent e; HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b)); e.flops = 0; e.bytes = 0; plp->prof.push_back(e);
            }
            pe = &plp->prof[plp->prof_used++];
            pe->flops = flops; pe->bytes = 0; pe->desc = desc_id;
            if (ksplit > 1) HIPCHK(hipEventRecord(pe->a, s));
        }
        if (lean && final_out && plp->cur_out) { IgemmP q = p; q.y = plp->cur_out; q.y_bs = plp->cur_out_bs; launch_igemm2(cfg, wg_ks, pre, lin, q, grid, lds2, s, pe ? pe->a : nullptr, pe ? pe->b : nullptr); }
        else if (lean) launch_igemm2(cfg, wg_ks, pre, lin, p, grid, lds2, s, pe ? pe->a : nullptr, pe ? pe->b : nullptr);
        else launch_igemm_v1(pre, p, grid, s);
        if (ksplit > 1) hipLaunchKernelGGL(splitk_epilogue_kernel, egrid, dim3(256), 0, s, p);
        if (pe && ksplit > 1) HIPCHK(hipEventRecord(pe->b, s));
    });
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Plan-time selection by measurement (the rules above are thresholds fitted on one box at the bench's stream counts and geometry; every
// plugin instance has its own geometry, obs-rvc/src/lib.rs:200-227, and boxes of one pool differ by 10 %).  With Plan::autotune (rvc_set_plan_autotune,
// default on above 4 streams) every layer that has more than one eligible kernel / tile is built as the rule-based choice AND as its neighbours across the
// nearest thresholds -- each in a scratch plan, launched on the engine's device: one warm-up, then the best of two timed launches (three when two candidates
// are within 8 %) -- and the fastest is queued.  Results are cached per process by (device, layer signature, streams), so the twelve identical transformer
// layers, the second plan slot and later engines cost nothing, and two engines of one process always agree.  Every candidate is a kernel the parity tests
// cover; the choice changes fp32 summation order at most.  A plan built while a planner test hook is set is never tuned (the hook IS the choice).
namespace {
struct TuneRec { Choice c; std::string desc; double us; int ncand; };
std::mutex g_tune_mu;
std::map<std::string, TuneRec> g_tune;
std::atomic<long long> g_tune_trials{0};
hipStream_t tune_stream()
{
    static thread_local std::map<int, hipStream_t> st;
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    auto it = st.find(dev);
    if (it != st.end()) return it->second;
    hipStream_t s; HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    st[dev] = s;
    return s;
}
bool planner_hook_set()
{
    static const char *const names[] = {"RVC_FORCE_CFG", "RVC_CONV_TILE", "RVC_CONV_TILE_KS", "RVC_FORCE_G2W", "RVC_CONV32S", "RVC_CONV32S_TILE", "RVC_G32L", "RVC_G32L_TALL", "RVC_G32L_TAB", "RVC_CONV32S_BUF", "RVC_FORCE_CHOICE", "RVC_G32L_PANEL"};
    for (const char *n : names) if (test_opt(n)) return true;
    return false;
}
}

// one candidate in a scratch plan: -> microseconds per launch (best of `reps`), < 0 if it cannot be built; *desc = the kernel it really became
static double tune_trial(const Plan &pl, const IgemmP &p, int B, const std::vector<int> &koff, const std::vector<PhaseD> &phases, const Choice &c, int reps, std::string *desc)
{
    // one scratch plan per thread, rewound between trials (its arena holds a layer's tables only)
    // (a plain pointer, never freed at thread / process exit: a destructor that calls hipFree behind the runtime's own teardown crashed the process
    //  at exit under rocprofv3 -- round 6, profiles of the first autotuned build)
    static thread_local Plan *scratch = nullptr;
    static thread_local int scratch_dev = -1;
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    if (!scratch || scratch_dev != dev) { delete scratch; scratch = new Plan(); scratch->arena.set_chunk_min((size_t)8 << 20); scratch_dev = dev; }
    Plan &tp = *scratch;
    HIPCHK(hipStreamSynchronize(tune_stream()));          // (the previous trial's launches still read the tables that are about to be overwritten)
    tp.ops = OpList(); tp.descs.clear(); tp.koff_tabs.clear(); tp.arena.rewind();
    tp.B = pl.B; tp.bf3 = pl.bf3; tp.autotune = false; tp.profile = false; tp.igemm_flops = 0; tp.n_igemm = 0;
    const Choice saved = t_choice;
    t_choice = c;
    try { queue_igemm_impl(tp, p, B, koff, phases, false); }
    catch (...) { t_choice = saved; return -1.0; }
    t_choice = saved;
    if (tp.ops.v.empty() || tp.descs.empty()) return -1.0;
    *desc = tp.descs.back();
    hipStream_t st = tune_stream();
    static thread_local hipEvent_t ea = nullptr, eb = nullptr;
    if (!ea) { HIPCHK(hipEventCreate(&ea)); HIPCHK(hipEventCreate(&eb)); }
    Op &op = tp.ops.v.back();
    // the first launch is a warm-up (code object, tables and weights into the caches) -- and, timed, already the answer for a long kernel, whose cold-start
    // share is below the 1.5 % a candidate must win by
    double best = 1e30;
    for (int r = 0; r <= reps; r++) {
        HIPCHK(hipEventRecord(ea, st));
        op(st);
        HIPCHK(hipEventRecord(eb, st));
        HIPCHK(hipEventSynchronize(eb));
        float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, ea, eb));
        if (r == 0 && ms < 0.4f) continue;          // short kernels: the warm-up does not count
        best = std::min(best, (double)ms * 1e3);
        if (ms >= 0.4f && r + 1 >= reps) break;          // long kernels: one launch fewer
    }
    HIPCHK(hipGetLastError());
    g_tune_trials.fetch_add(1);
    return best;
}

void queue_igemm(Plan &pl, IgemmP p, int B, const std::vector<int> &koff, const std::vector<PhaseD> &phases, bool final_out)
{
    if (const char *f = test_opt("RVC_FORCE_CHOICE")) {          // test hook "kind,a,b": every layer built under ONE of the choices the tuner can make (tests/test_gpu_tiles.py)
        Choice c; if (sscanf(f, "%d,%d,%d", &c.kind, &c.a, &c.b) < 1) c = Choice();
        const Choice saved = t_choice;
        t_choice = c;
        try { queue_igemm_impl(pl, p, B, koff, phases, final_out); } catch (...) { t_choice = saved; throw; }
        t_choice = saved;
        return;
    }
    if (!pl.autotune || B <= 4 || pl.bf3 || p.bf3 || p.ln_wsum || p.ln_stats_in || p.ln_stats_out || planner_hook_set() || t_choice.kind != 0) return queue_igemm_impl(pl, p, B, koff, phases, final_out);
    // layer signature: everything the kernels' speed depends on (shape, strides, taps, epilogue class), not the tensors
    std::string key;
    {
        int dev = 0; HIPCHK(hipGetDevice(&dev));
        unsigned long long h = 1469598103934665603ull;
        auto mix = [&](long long v) { h ^= (unsigned long long)v; h *= 1099511628211ull; };
        for (size_t i = 0; i < koff.size(); i += std::max<size_t>(1, koff.size() / 97)) mix(koff[i]);
        mix((long long)koff.size());
        for (const PhaseD &q : phases) { mix(q.nchunks); mix(q.y_pos); mix(q.y_h0); mix(q.act_p1); mix(q.y_off != 0); mix(q.x_off); }
        char k[256];
        snprintf(k, sizeof k, "d%d B%d M%d N%d K%d ph%zu NW%d xs%d,%d ys%d,%d ld%d lin%d pre%d act%d res%d acc%d glu%d fo%d h%llx", dev, B, p.M, p.N, p.K, phases.size(), p.NW, p.x_hs, p.x_ws, p.y_hm, p.y_ws,
                 p.x_ld, p.lin_cs4 != 0, p.pre_act != ACT_NONE, p.act, p.res != nullptr, p.accumulate, p.glu, (int)final_out, h);
        key = k;
    }
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        auto it = g_tune.find(key);
        if (it != g_tune.end()) {
            const Choice c = it->second.c;
            t_choice = c;
            try { queue_igemm_impl(pl, p, B, koff, phases, final_out); } catch (...) { t_choice = Choice(); throw; }
            t_choice = Choice();
            pl.tune_hits++;
            return;
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    // the rule-based build first: its family decides which neighbours are worth a trial
    std::string d0;
    const double us0 = tune_trial(pl, p, B, koff, phases, Choice(), 2, &d0);
    if (us0 < 0) return queue_igemm_impl(pl, p, B, koff, phases, final_out);
    const std::string fam = d0.substr(0, d0.find(' '));
    std::vector<Choice> cands;
    auto add = [&](int kind, int a, int b) { Choice c; c.kind = kind; c.a = a; c.b = b; for (const Choice &o : cands) if (o == c) return; cands.push_back(c); };
    const bool one_d = p.x_ld > 0 && p.x_hs == 0 && p.x_ws == 1 && p.y_hm == 0 && p.y_ws == 1 && !p.glu;      // conv32s_kernel's domain (it checks the rest itself)
    const bool lin = p.lin_cs4 != 0 && phases.size() == 1 && p.pre_act == ACT_NONE && !p.glu;
    const int t32 = p.M <= 32 ? 0 : (p.M <= 64 ? 1 : 2);                                              // conv32s tile by panel height
    const int lc_m = p.M >= 96 ? 7 : (p.M >= 48 ? 8 : 5);                                             // a 32x32x2 workgroup tile by panel height
    int nchunks = p.K / 16;
    auto g2w_ks = [&](int &ks) {          // g2w_rule's K split on the folded column count (the rule itself sees the layer after the fold)
        const long long tiles = (long long)((p.M + 31) / 32) * (((long long)B * p.N + 31) / 32), want = (4800 + tiles / 2) / std::max<long long>(tiles, 1);
        ks = want <= 1 ? 1 : (want == 2 ? 2 : (want == 3 ? 3 : (want <= 4 ? 4 : 8)));
        while (ks > 1 && nchunks / ks < 4) ks = ks == 8 ? 4 : ks - 1;
    };
    if (fam == "c32s") {
        add(1, t32 == 2 ? 1 : t32 + 1, 0); if (t32 > 0) add(1, t32 - 1, 0);
        if (t32 == 1) { add(1, 1 | 4, 0); add(1, 1, 0); }            // both load variants of the 64 x 128 tile
        add(3, lc_m, 0);
    } else if (fam == "g2w") {
        int ks; g2w_ks(ks);
        add(2, 0, ks == 1 ? 2 : ks / 2); if (ks < 8) add(2, 0, ks == 3 ? 4 : ks * 2);
        add(3, p.M >= 96 ? 8 : lc_m, 0); add(4, 3, 4);
    } else if (fam == "g32l" || fam == "g32t" || fam == "g32" || fam == "lds") {
        // the neighbouring workgroup tiles, the staged convolution / the register-direct 32x32x2 kernel where the layer is in their domain, the register-direct 16x16x4 kernel
        if (p.M >= 96) { add(3, 7, 0); add(3, 3, 0); add(3, 8, 0); } else if (p.M >= 48) { add(3, 4, 0); add(3, 8, 0); add(3, 5, 0); } else add(3, 5, 0);
        if (one_d && !lin) add(1, t32, 0);
        if (lin && p.M >= 256) { int ks; g2w_ks(ks); add(2, 0, ks); }
        add(4, p.M > 16 ? 3 : 1, nchunks >= 16 ? 4 : 1);
    } else if (fam == "reg") {
        if (one_d && !lin) add(1, t32, 0);
        if (lin && p.M >= 256) { int ks; g2w_ks(ks); add(2, 0, ks); }
        if (p.M > 16) add(3, lc_m, 0);
        // the other K split of the same tile (the rule wants >= 1024 waves; with streams folded into N a wave count between the steps goes either way)
        {
            int ks = 1; const char *q = strstr(d0.c_str(), " ks="); if (q) ks = atoi(q + 4);
            int cf = 0; const char *t = strstr(d0.c_str(), " tile="); int bm = 16, bn = 16; if (t && sscanf(t + 6, "%dx%d", &bm, &bn) == 2) { for (int c = 0; c < 5; c++) if (16 * kMF[c] == bm && 16 * kNF[c] == bn) cf = c; }
            add(4, cf, ks == 1 ? 4 : (ks == 4 ? 8 : 4));
            if (cf == 4 || cf == 3) add(4, cf == 4 ? 3 : 0, ks);
        }
    }
    struct Res { Choice c; std::string d; double us; };
    std::vector<Res> res;
    res.push_back({Choice(), d0, us0});
    for (const Choice &c : cands) {
        std::string d;
        // skip what cannot beat the leader by a wide margin: a single timed launch first
        double us = tune_trial(pl, p, B, koff, phases, c, 1, &d);
        if (us < 0) continue;
        bool dup = false;
        for (const Res &r : res) dup = dup || r.d == d;
        if (dup) continue;
        double lead = 1e30; for (const Res &r : res) lead = std::min(lead, r.us);
        if (us < lead * 1.08) us = std::min(us, tune_trial(pl, p, B, koff, phases, c, 2, &d));
        res.push_back({c, d, us});
    }
    size_t bi = 0;
    for (size_t i = 1; i < res.size(); i++) if (res[i].us < res[bi].us) bi = i;
    // the rules keep a tie: a candidate must win by more than the noise of two launches
    if (bi != 0 && res[bi].us > res[0].us * 0.985) bi = 0;
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        g_tune[key] = TuneRec{res[bi].c, res[bi].d, res[bi].us, (int)res.size()};
    }
    pl.tune_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    pl.tuned_layers++;
    if (bi != 0) pl.tune_changed++;
    t_choice = res[bi].c;
    try { queue_igemm_impl(pl, p, B, koff, phases, final_out); } catch (...) { t_choice = Choice(); throw; }
    t_choice = Choice();
}

// the autotuner's decisions of this process, one line each: "<key> -> <kernel description> <us> (<candidates>)"; returns the number of entries (test / tool aid)
extern "C" int rvc_debug_autotune_dump(char *buf, size_t cap)
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    std::string out;
    for (const auto &kv : g_tune) {
        char ln[640];
        snprintf(ln, sizeof ln, "%s -> [%d,%d,%d] %s | %.1f us (%d candidates)\n", kv.first.c_str(), kv.second.c.kind, kv.second.c.a, kv.second.c.b, kv.second.desc.c_str(), kv.second.us, kv.second.ncand);
        out += ln;
    }
    if (buf && out.size() + 1 <= cap) memcpy(buf, out.c_str(), out.size() + 1);
    return (int)g_tune.size();
}
extern "C" void rvc_debug_autotune_reset(void)
{
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tune.clear();
}

void fill_epilogue(IgemmP &p, const ConvW &cw, const ConvOpts &o)
{
    p.bias = (o.no_bias || !cw.bias) ? nullptr : cw.bias + o.m_off;
    p.res = o.res; p.res_cs = o.res_cs; p.res_bs = o.res_bs; p.res_rs = o.res_rs;
    p.act = o.act; p.slope = o.slope; p.scale = o.scale; p.accumulate = o.accumulate ? 1 : 0;
    if (o.pre_act != ACT_NONE && o.pre_act != ACT_LRELU) throw std::runtime_error("only LeakyReLU can be fused on the input side");
    p.pre_act = o.pre_act; p.pre_slope = o.pre_act == ACT_LRELU ? o.pre_slope : 1.0f;
    p.part = nullptr;
    p.glu = o.glu ? 1 : 0;
    p.ln_wsum = o.ln_wsum; p.ln_stats_out = o.ln_stats_out; p.ln_stats_in = o.ln_stats_in; p.ln_g = o.ln_g; p.ln_bt = o.ln_b;
    p.ln_eps = 1e-5f; p.ln_inv_rows = o.ln_rows > 0 ? 1.0f / (float)o.ln_rows : 0.f;
    if (o.ln_stats_in && !(o.res && o.ln_g && o.ln_b)) throw std::logic_error("normalised residual without residual / scale / shift");
    if (o.glu && (p.bias == nullptr || o.res || o.accumulate || o.act != ACT_NONE)) throw std::runtime_error("glu epilogue takes bias only");
}

// Conv1d (stride s, dilation d, symmetric zero padding pad, groups) on halo'd rows
void add_conv1d(Plan &pl, const ConvW &cw, const T1 &x, const T1 &y, int stride, int pad, int dil, ConvOpts o)
{
    if (cw.transposed) throw std::runtime_error("add_conv1d on transposed weights");
    const int cig = cw.Cin / cw.groups, KW = cw.KW;
    const int Tout = (x.T + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
    if (Tout != y.T && !(Tout == y.T + 1)) throw ShapeError("conv1d output length mismatch");
    if (x.halo < pad || (y.T - 1) * stride + (KW - 1) * dil - pad > x.T - 1 + x.halo) throw ShapeError("conv1d halo too small");
    IgemmP p{};
    if (o.m_off % 16 != 0) throw std::runtime_error("output-row sub-range must start at a multiple of 16");
    p.x = x.p; p.w = cw.w + (long long)o.m_off * cw.Kp; p.y = y.p;
    p.M = o.m_cnt >= 0 ? o.m_cnt : cw.M; p.N = y.T; p.K = cw.Kp;
    p.NW = y.T; p.x_hs = 0; p.x_ws = stride; p.y_hm = 0; p.y_ws = 1; p.OW = y.T;
    p.x_bs = x.bs; p.y_bs = y.bs; p.y_cs = y.ld; p.y_rs = 0;
    p.x_ld = x.ld; p.x_lo = -x.halo; p.x_lim = x.T - 1 + x.halo;
    fill_epilogue(p, cw, o);
    p.bf3 = o.bf3 ? 1 : 0;
    // 1x1 convolution with a whole number of 16-row chunks: operand row k sits at k * channel stride, no offset table (igemm2 LIN)
    if (KW == 1 && cw.groups == 1 && pad == 0 && cw.K == cw.Kp) p.lin_cs4 = x.ld * 4;
    if (cw.groups > 1 && (o.m_off != 0 || o.m_cnt >= 0)) throw std::runtime_error("row sub-range on grouped conv");
    std::vector<int> koff(cw.Kp, 0);
    for (int ci = 0; ci < cig; ci++) for (int k = 0; k < KW; k++) koff[ci * KW + k] = ci * x.ld + k * dil - pad;
    std::vector<PhaseD> ph(cw.groups);
    for (int g = 0; g < cw.groups; g++) {
        ph[g] = PhaseD{};
        ph[g].w_off = (long long)g * phase_stride(cw);
        ph[g].x_off = g * cig * x.ld;
        ph[g].y_c0 = g * cw.M;
        ph[g].y_pos = 0;
        ph[g].bias_off = g * cw.M;
        ph[g].koff_off = 0;
    }
    queue_igemm(pl, p, x.B, koff, ph, o.final_out);
}

// Several stride-1 convs of the same Cin/Cout but different kernel size / dilation as ONE launch (phase j = conv j): the
// HiFiGAN stage's parallel ResBlock chains.  x is either shared by all convs or a [n*Cin] tensor holding conv j's input in rows
// j*Cin..; y is a [n*Cout] tensor (conv j writes rows j*Cout..); the residual is shared or grouped likewise.
void add_conv1d_multi(Plan &pl, const std::vector<const ConvW *> &cws, const T1 &x, bool x_grouped, const T1 &y,
                             const std::vector<int> &pads, const std::vector<int> &dils, ConvOpts o, bool res_grouped)
{
    const int n = (int)cws.size();
    const ConvW &c0 = *cws[0];
    IgemmP p{};
    p.x = x.p; p.w = c0.w; p.y = y.p;
    p.M = c0.M; p.N = y.T; p.K = 0;
    p.NW = y.T; p.x_hs = 0; p.x_ws = 1; p.y_hm = 0; p.y_ws = 1; p.OW = y.T;
    p.x_bs = x.bs; p.y_bs = y.bs; p.y_cs = y.ld; p.y_rs = 0;
    p.x_ld = x.ld; p.x_lo = -x.halo; p.x_lim = x.T - 1 + x.halo;
    fill_epilogue(p, c0, o);
    p.res_nogroup = res_grouped ? 0 : 1;
    std::vector<int> koff;
    std::vector<PhaseD> ph(n);
    for (int j = 0; j < n; j++) {
        const ConvW &cw = *cws[j];
        if (cw.transposed || cw.groups != 1 || cw.Cin != c0.Cin || cw.Cout != c0.Cout || (x_grouped ? x.C != n * cw.Cin : x.C != cw.Cin) || y.C != n * cw.Cout)
            throw std::runtime_error("add_conv1d_multi: incompatible convs");
        const int KW = cw.KW, pad = pads[j], dil = dils[j];
        if (x.T + 2 * pad - dil * (KW - 1) != y.T) throw ShapeError("conv1d_multi output length mismatch");
        if (x.halo < pad || (KW - 1) * dil - pad > x.halo) throw ShapeError("conv1d_multi halo too small");
        ph[j] = PhaseD{};
        ph[j].w_off = cw.w - c0.w;                       // same allocation (merge_convs)
        ph[j].bias_off = (int)(cw.bias - c0.bias);
        ph[j].x_off = x_grouped ? j * cw.Cin * x.ld : 0;
        ph[j].y_c0 = j * cw.Cout;
        ph[j].koff_off = (int)koff.size();
        ph[j].nchunks = cw.Kp / 16;
        p.K = std::max(p.K, cw.Kp);
        const size_t base = koff.size();
        koff.resize(base + cw.Kp, 0);
        for (int ci = 0; ci < cw.Cin; ci++) for (int k = 0; k < KW; k++) koff[base + ci * KW + k] = ci * x.ld + k * dil - pad;
    }
    queue_igemm(pl, p, x.B, koff, ph);
}

// ConvTranspose1d (polyphase), pad = (K - S) / 2 as in HiFiGAN
void add_convT1d(Plan &pl, const ConvW &cw, const T1 &x, const T1 &y, int pad, ConvOpts o)
{
    const int S = cw.S, nt = cw.ntaps;
    const int Tout = (x.T - 1) * S - 2 * pad + cw.KW;
    if (Tout != y.T) throw ShapeError("convT1d output length mismatch");
    if (x.halo < nt) throw ShapeError("convT1d halo too small");
    IgemmP p{};
    p.x = x.p; p.w = cw.w; p.y = y.p;
    p.M = cw.M; p.N = x.T + nt - 1; p.K = cw.Kp;
    p.NW = p.N; p.x_hs = 0; p.x_ws = 1; p.y_hm = 0; p.y_ws = S; p.OW = y.T;
    p.x_bs = x.bs; p.y_bs = y.bs; p.y_cs = y.ld; p.y_rs = 0;
    fill_epilogue(p, cw, o);
    std::vector<int> koff(cw.Kp, 0);
    for (int ci = 0; ci < cw.Cin; ci++) for (int j = 0; j < nt; j++) koff[ci * nt + j] = ci * x.ld - j;
    std::vector<PhaseD> ph(S);
    for (int q = 0; q < S; q++) {
        ph[q] = PhaseD{};
        ph[q].w_off = (long long)q * phase_stride(cw);
        ph[q].x_off = 0;
        ph[q].y_pos = q - pad;
        ph[q].bias_off = 0;
        ph[q].koff_off = 0;
    }
    queue_igemm(pl, p, x.B, koff, ph);
}

// Conv2d 3x3 pad 1 (KW = 9) or 1x1 (KW = 1) on halo'd images
void add_conv2d(Plan &pl, const ConvW &cw, const T2 &x, const T2 &y, ConvOpts o)
{
    if (x.H != y.H || x.W != y.W) throw ShapeError("conv2d shape mismatch");
    IgemmP p{};
    p.x = x.p; p.w = cw.w; p.y = y.p;
    p.M = cw.M; p.N = x.H * x.W; p.K = cw.Kp;
    p.NW = x.W; p.x_hs = x.ld; p.x_ws = 1; p.y_hm = 1; p.y_ws = o.y_ws > 0 ? o.y_ws : 1; p.OW = o.y_ws > 0 ? y.W * o.y_ws : y.W;      // (OW bounds the column OFFSET)
    p.x_bs = x.bs; p.y_bs = y.bs; p.y_cs = y.cs; p.y_rs = y.ld;
    fill_epilogue(p, cw, o);
    std::vector<int> koff;
    if (cw.KW == 9 && x.H == 1 && !cw.host_w.empty() && !tune_env("RVC_NO_TAP_PRUNE")) {
        // one-row image (RMVPE's bottleneck at Tm = 32): the kh = 0 and kh = 2 taps only ever read the zero halo rows, so two
        // thirds of the weight stream is dead.  Repack the middle row of every 3x3 filter once per plan (K = Cin*3).
        const int K3 = cw.Cin * 3, Kp3 = round16(K3);
        std::vector<float> panel((size_t)cw.M * Kp3, 0.f);
        for (int mo = 0; mo < cw.M; mo++)
            for (int ci = 0; ci < cw.Cin; ci++)
                for (int kw = 0; kw < 3; kw++) panel[(size_t)mo * Kp3 + ci * 3 + kw] = cw.host_w[(size_t)mo * cw.K + ci * 9 + 3 + kw];
        float *dw = upload_fragments(panel, 1, cw.M, Kp3, 1);      // plan-lifetime copy
        pl.owned_dev.push_back(dw);
        p.w = dw; p.K = Kp3;
        koff.assign(Kp3, 0);
        for (int ci = 0; ci < cw.Cin; ci++) for (int kw = 0; kw < 3; kw++) koff[ci * 3 + kw] = ci * x.cs + (kw - 1);
    } else {
        koff.assign(cw.Kp, 0);
        if (cw.KW == 9) { for (int ci = 0; ci < cw.Cin; ci++) for (int k = 0; k < 9; k++) koff[ci * 9 + k] = ci * x.cs + (k / 3 - 1) * x.ld + (k % 3 - 1); }
        else { for (int ci = 0; ci < cw.Cin; ci++) koff[ci] = ci * x.cs; }
    }
    std::vector<PhaseD> ph(1);
    ph[0] = PhaseD{};
    queue_igemm(pl, p, x.B, koff, ph);
}

void add_convT2d(Plan &pl, const ConvW &cw, const T2 &x, const T2 &y, ConvOpts o)
{
    if (y.H != 2 * x.H || y.W != 2 * x.W) throw ShapeError("convT2d shape mismatch");
    IgemmP p{};
    p.x = x.p; p.w = cw.w; p.y = y.p;
    p.M = cw.M; p.N = x.H * x.W; p.K = cw.Kp;
    p.NW = x.W; p.x_hs = x.ld; p.x_ws = 1; p.y_hm = 2; p.y_ws = 2; p.OW = y.W;
    p.x_bs = x.bs; p.y_bs = y.bs; p.y_cs = y.cs; p.y_rs = y.ld;
    fill_epilogue(p, cw, o);
    std::vector<int> koff(cw.Kp, 0);
    for (int ci = 0; ci < cw.Cin; ci++) for (int jh = 0; jh < 2; jh++) for (int jw = 0; jw < 2; jw++) koff[ci * 4 + jh * 2 + jw] = ci * x.cs + jh * x.ld + jw;
    std::vector<PhaseD> ph(4);
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) {
        PhaseD d{};
        d.w_off = (long long)(a * 2 + b) * phase_stride(cw);
        d.y_h0 = a;
        d.y_pos = b;
        ph[a * 2 + b] = d;
    }
    queue_igemm(pl, p, x.B, koff, ph);
}

void add_layernorm(Plan &pl, const T1 &x, const float *g, const float *b)
{
    dim3 grid((x.T + 3) / 4, x.B);
    if (x.C > 1024) throw ShapeError("layernorm: more than 1024 channels");
    const bool small = x.C <= 256;
    // many streams: 16-column strips held in registers (float4 rows; needs 16-byte aligned rows, which every plan tensor has: ld and
    // halo are multiples of 4).  Reading the padding columns behind T is safe (inside the row), they are never written.
    if (x.B >= 16 && x.ld % 4 == 0 && x.halo % 4 == 0 && ((x.T + 3) / 4 * 4 <= x.ld - x.halo) && !tune_env("RVC_NO_LN_STRIP")) {
        // grid x = stream, y = strip: workgroup (b, strip) runs on XCD (strip * B + b) % 8 = b % 8 when B is a multiple of 8, so the two
        // 64-byte halves of every 128-byte line (adjacent strips of one stream) are fetched by the same XCD's L2, once
        dim3 sg(x.B, (x.T + 15) / 16);
        const int nr = (x.C + 63) / 64;
        pl.ops.push_back([=](hipStream_t s) {
            if (nr <= 4) hipLaunchKernelGGL((layernorm_strip_kernel<4>), sg, dim3(256), 0, s, x.p, x.p, g, b, x.C, x.T, x.ld, x.bs, x.ld, x.bs);
            else if (nr <= 12) hipLaunchKernelGGL((layernorm_strip_kernel<12>), sg, dim3(256), 0, s, x.p, x.p, g, b, x.C, x.T, x.ld, x.bs, x.ld, x.bs);
            else hipLaunchKernelGGL((layernorm_strip_kernel<16>), sg, dim3(256), 0, s, x.p, x.p, g, b, x.C, x.T, x.ld, x.bs, x.ld, x.bs);
        });
        return;
    }
    if (x.B >= 16 && x.C > 256 && (size_t)x.C * 33 * 4 <= 150 * 1024 && !tune_env("RVC_NO_LN_TILE")) {
        dim3 tg((x.T + 31) / 32, x.B);
        const size_t lds = (size_t)x.C * 33 * sizeof(float);
        pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(layernorm_tile_kernel, tg, dim3(256), lds, s, x.p, x.p, g, b, x.C, x.T, x.ld, x.bs, x.ld, x.bs); });
        return;
    }
    pl.ops.push_back([=](hipStream_t s) {
        if (small) hipLaunchKernelGGL((layernorm_ct_kernel<4>), grid, dim3(256), 0, s, x.p, x.p, g, b, x.C, x.T, x.ld, x.bs, x.ld, x.bs);
        else hipLaunchKernelGGL((layernorm_ct_kernel<16>), grid, dim3(256), 0, s, x.p, x.p, g, b, x.C, x.T, x.ld, x.bs, x.ld, x.bs);
    });
}

void add_stamp(Plan &pl, const char *name)
{
    const bool on = test_opt("RVC_STAMPS") != nullptr;
    if (!on) return;
    if (!pl.d_stamps) pl.d_stamps = reinterpret_cast<unsigned long long *>(pl.arena.floats(2 * 256));
    if (pl.stamp_names.size() >= 256) return;
    unsigned long long *slot = pl.d_stamps + pl.stamp_names.size();
    pl.stamp_names.push_back(name);
    pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, slot); });
}
void add_tap(Plan &pl, const char *name, const T1 &t)
{
    add_stamp(pl, name);
    if (!pl.with_taps) return;
    // snapshot into a private contiguous-row tensor so later in-place ops do not clobber it
    T1 snap = make_t1(pl.arena, 1, t.C, t.T, 0);
    pl.ops.push_back([=](hipStream_t s) {
        HIPCHK(hipMemcpy2DAsync(snap.p, (size_t)snap.ld * 4, t.p, (size_t)t.ld * 4, (size_t)t.T * 4, t.C, hipMemcpyDeviceToDevice, s));
    });
    TapRec r; r.name = name; r.rank = 1; r.t1 = snap; pl.taps.push_back(r);
}
void add_tap2(Plan &pl, const char *name, const T2 &t)
{
    add_stamp(pl, name);
    if (!pl.with_taps) return;
    TapRec r; r.name = name; r.rank = 2; r.t2 = t; pl.taps.push_back(r);   // RMVPE images are never overwritten
}

// Two 1x1 convolutions of ONE input with the same M and K into two output tensors, as two phases of one launch (the flows' merged post / next-pre
// layer).  pair_bias = [c0's bias | c1's bias].
void add_conv1d_two(Plan &pl, const ConvW &c0, const ConvW &c1, const float *pair_bias, const T1 &x, const T1 &y0, const T1 &y1)
{
    if (c0.M != c1.M || c0.Kp != c1.Kp || c0.KW != 1 || c1.KW != 1 || c0.Cin != x.C || y0.ld != y1.ld || y0.T != y1.T || y0.bs != y1.bs || y0.C != c0.M || y1.C != c1.M)
        throw ShapeError("conv1d pair: shapes differ");
    IgemmP p{};
    p.x = x.p; p.w = c0.w; p.y = y0.p;
    p.M = c0.M; p.N = y0.T; p.K = c0.Kp;
    p.NW = y0.T; p.x_hs = 0; p.x_ws = 1; p.y_hm = 0; p.y_ws = 1; p.OW = y0.T;
    p.x_bs = x.bs; p.y_bs = y0.bs; p.y_cs = y0.ld; p.y_rs = 0;
    ConvOpts o;
    fill_epilogue(p, c0, o);
    p.bias = pair_bias;
    std::vector<int> koff(c0.Kp, 0);
    for (int ci = 0; ci < c0.Cin; ci++) koff[ci] = ci * x.ld;
    std::vector<PhaseD> ph(2);
    ph[0] = PhaseD{}; ph[1] = PhaseD{};
    ph[0].nchunks = c0.Kp / 16;
    ph[1].w_off = c1.w - c0.w;          // both are device pointers of one flat address space
    ph[1].nchunks = c1.Kp / 16;
    ph[1].koff_off = 0;
    ph[1].bias_off = c0.M;
    ph[1].act_p1 = ACT_NONE + 1;
    ph[1].y_off = y1.p - y0.p;
    queue_igemm(pl, p, x.B, koff, ph);
}


void plan_kernel_attrs()
{
    HIPCHK(hipFuncSetAttribute((const void *)layernorm_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));   // + 1.3 KB static
    conv_tile_prepare_device();
    igemm2w_prepare_device();
}

}  // namespace rvc

// test hook (see "switches" in engine_int.h): set (value != NULL) or clear one of the named hooks; 0 = done, -1 = unknown name.
// Hooks are read when a model is loaded (RVC_NO_LN_FUSE) or a plan is built -- set them before.
extern "C" int rvc_debug_weight_slabs(int device, int *count, size_t *bytes)
{
    int c = 0; size_t b = 0;
    rvc::wslab_info(device, &c, &b);
    if (count) *count = c;
    if (bytes) *bytes = b;
    return 0;
}

extern "C" int rvc_debug_option(const char *name, const char *value)
{
    using namespace rvc;
    if (!name) return -1;
#ifndef RVC_TUNING
    if (!is_test_hook(name)) return -1;
#endif
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (value) g_opts[name] = value; else g_opts.erase(name);
    if (!strcmp(name, "RVC_KNN_LOSE_TICKET")) g_knn_test_lose.store(value && atoi(value) ? 1 : 0);      // (read per launch: kept out of the string table's lock)
    g_opt_gen.fetch_add(1);
    return 0;
}
