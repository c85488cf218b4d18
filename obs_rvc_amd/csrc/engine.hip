// engine.hip -- host side of the MI355X-native RVC streaming inference engine + its C ABI.
//
// Mirrors rvc::RvcInfer (reference: rvc/src/rvc.rs:18-220): model handles, the 1024-entry
// pitch cache, and the per-chunk pipeline hubert -> (retrieval) -> pitch -> synthesizer.
// All compute is launched as hand-written gfx950 kernels (kernels.hip.h); nothing here falls
// back to a CPU path: without a HIP device every entry point returns RVC_BACKEND.
#include "../../include/rvc_mi355x.h"
#include "blob.h"
#include "kernels.hip.h"
#include "igemm_launch.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <thread>

// The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The engine runs four
// streams concurrently per chunk; a second engine in the process, or the streams an RCCL communicator leaves behind, then share
// queues with them and the per-chunk latency rises by 5-20 % (measured: 2.38 -> 2.84 ms for an engine created after a communicator;
// flat 2.38 ms with 16 queues).  The variable is read when the runtime initialises, so a default is planted when this library is
// loaded -- it does not override a value the host has set, and a host that has already initialised HIP should set it itself
// (the variable then has no effect: rvc_create says so once on stderr).  A host that wants its environment left alone sets
// RVC_NO_RUNTIME_DEFAULTS=1 (documented in include/rvc_mi355x.h and INTEGRATION.md).
static bool g_planted_queues = false;
__attribute__((constructor)) static void rvc_runtime_defaults()
{
    if (getenv("RVC_NO_RUNTIME_DEFAULTS")) return;
    if (!getenv("GPU_MAX_HW_QUEUES")) { setenv("GPU_MAX_HW_QUEUES", "16", 0); g_planted_queues = true; }
}

namespace rvc {

// ---------------------------------------------------------------------------------------
// switches
// ---------------------------------------------------------------------------------------
// The product library reads four environment variables and no others: GPU_MAX_HW_QUEUES (a default is planted, see above),
// RVC_NO_RUNTIME_DEFAULTS, LOCAL_RANK (rvc_create with device < 0) and RVC_RCCL_LIB (rccl_bcast.hip.h); the rvc-rpc executable adds
// RVC_NOISE_SEED and RVC_USE_GRAPH.  Every other switch is
//   * a TEST HOOK (kTestHooks): set with rvc_debug_option(name, value) by the parity tests and the profiling tools -- an explicit call,
//     never inherited from a host's environment -- to force a code path the planner would not pick for the geometry at hand; or
//   * a TUNING switch (tune_env): compiled out of the product (the call is a constant nullptr, its branch disappears); only builds with
//     -DRVC_TUNING (tests/tools/build_tuning.py -> librvc_tuning.so) have them, and there both kinds also fall back to the environment
//     variable of the same name.
static const char *const kTestHooks[] = {"RVC_FORCE_CFG", "RVC_CONV_TILE", "RVC_CONV_TILE_KS", "RVC_NO_LN_FUSE", "RVC_NO_CONV0_MULTI", "RVC_KNN_NO_GEMM",
                                         "RVC_KNN_EXHAUSTIVE", "RVC_STAMPS", "RVC_SERIAL_BRANCHES", "RVC_NO_WN_COMPOSE"};
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
static const char *test_opt(const char *name) { const char *v = opt_lookup(name); return v ? v : getenv(name); }
static const char *tune_env(const char *name) { const char *v = opt_lookup(name); return v ? v : getenv(name); }
#else
static const char *test_opt(const char *name) { return opt_lookup(name); }
static inline const char *tune_env(const char *) { return nullptr; }
#endif
static int test_opt_int(const char *name, int dflt) { const char *v = test_opt(name); return v ? atoi(v) : dflt; }

#define HIPCHK(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

struct ShapeError : std::runtime_error { using std::runtime_error::runtime_error; };
struct PanicError : std::runtime_error { using std::runtime_error::runtime_error; };

// ---------------------------------------------------------------------------------------
// device memory
// ---------------------------------------------------------------------------------------
class Arena {
public:
    ~Arena() { for (void *c : chunks_) (void)hipFree(c); }
    void *alloc(size_t bytes)
    {
        bytes = (bytes + 255) / 256 * 256;
        if (bytes > left_) {
            size_t sz = std::max(bytes, (size_t)64 << 20);
            void *c;
            HIPCHK(hipMalloc(&c, sz));
            HIPCHK(hipMemset(c, 0, sz));
            chunks_.push_back(c);
            cur_ = (char *)c;
            left_ = sz;
            total_ += sz;
        }
        void *r = cur_;
        cur_ += bytes;
        left_ -= bytes;
        return r;
    }
    float *floats(size_t n) { return (float *)alloc(n * sizeof(float)); }
    template <typename T> T *upload(const std::vector<T> &v)
    {
        T *d = (T *)alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
        if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return d;
    }
    size_t total() const { return total_; }

private:
    std::vector<void *> chunks_;
    char *cur_ = nullptr;
    size_t left_ = 0, total_ = 0;
};

// 1-D activation [B][C][ld]: row = halo | T | halo (halo stays zero)
struct T1 {
    float *p = nullptr;   // -> [0][0][0]
    int B = 1, C = 0, T = 0, ld = 0, halo = 0;
    long long bs = 0;
    T1 rows(int c0, int n) const { T1 r = *this; r.p = p + (long long)c0 * ld; r.C = n; return r; }
};
// 2-D activation [B][C][H+2][W+2]
struct T2 {
    float *p = nullptr;   // -> interior (0,0) of channel 0
    int B = 1, C = 0, H = 0, W = 0, ld = 0, cs = 0;
    long long bs = 0;
    T2 chans(int c0, int n) const { T2 r = *this; r.p = p + (long long)c0 * cs; r.C = n; return r; }
};

static T1 make_t1(Arena &a, int B, int C, int T, int halo)
{
    T1 t;
    t.B = B; t.C = C; t.T = T; t.halo = halo;
    t.ld = (T + 2 * halo + 3) / 4 * 4;
    t.bs = (long long)C * t.ld;
    // guard rows in front and behind so clamped/garbage tail reads stay inside the allocation
    size_t guard = (size_t)t.ld + 64;
    float *base = a.floats((size_t)B * t.bs + 2 * guard);
    t.p = base + guard + halo;
    return t;
}
static T2 make_t2(Arena &a, int B, int C, int H, int W)
{
    T2 t;
    t.B = B; t.C = C; t.H = H; t.W = W;
    t.ld = W + 2;
    t.cs = (H + 2) * t.ld;
    t.bs = (long long)C * t.cs;
    size_t guard = (size_t)t.ld * 2 + 64;
    float *base = a.floats((size_t)B * t.bs + 2 * guard);
    t.p = base + guard + t.ld + 1;
    return t;
}

// ---------------------------------------------------------------------------------------
// prepared convolution weights: [nphase][M][Kp] panels (Kp = K rounded up to 16, zero padded)
// ---------------------------------------------------------------------------------------
struct ConvW {
    float *w = nullptr, *bias = nullptr;
    int M = 0, K = 0, Kp = 0, nphase = 1;
    int Cin = 0, Cout = 0, KW = 1, groups = 1;
    int S = 1, ntaps = 1;      // transposed convs
    bool transposed = false;
    bool owns = true;          // false: w / bias point into a buffer owned by another ConvW (merge_convs)
    std::vector<float> host_w; // row-major [Cout][K] copy kept for big 3x3 convs (plan-time tap pruning)
};

static int round16(int k) { return (k + 15) / 16 * 16; }

static float *upload_f(const std::vector<float> &v)
{
    float *d;
    HIPCHK(hipMalloc(&d, std::max<size_t>(v.size(), 4) * sizeof(float)));
    if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return d;
}
static float *upload_f(const float *p, size_t n) { return upload_f(std::vector<float>(p, p + n)); }

// [nphase][M][Kp] row-major panels -> MFMA-fragment-major [nphase][m_tile][chunk][lane][4] (M padded to 16 with zeros)
static float *upload_fragments(const std::vector<float> &panel, int nphase, int M, int Kp)
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
    return upload_f(out);
}
static long long phase_stride(const struct ConvW &c);

// Conv (any rank flattened to K = Cin/groups * KW taps): w [Cout][Cin/groups][KW]
static ConvW prep_conv(const float *w, const float *bias, int Cout, int Cin, int KW, int groups)
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
static ConvW prep_convT1d(const float *w, const float *bias, int Cin, int Cout, int K, int S)
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
static ConvW prep_convT2d(const float *w, const float *bias, int Cin, int Cout)
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
static long long phase_stride(const ConvW &c) { return (long long)((c.M + 15) / 16 * 16) * c.Kp; }
static void free_conv(ConvW &c)
{
    if (c.owns) {
        if (c.w) (void)hipFree(c.w);
        if (c.bias) (void)hipFree(c.bias);
    }
    c.w = c.bias = nullptr;
}
// Re-home the weights of several convolutions in ONE device allocation (the first one owns it), so that a fused launch can
// address them as phases of one weight buffer (PhaseD::w_off / bias_off are offsets from the first conv's pointers).
static void merge_convs(const std::vector<ConvW *> &cs)
{
    size_t tw = 0, tb = 0;
    for (ConvW *c : cs) { if (!c->owns || !c->bias) throw std::runtime_error("merge_convs: unexpected conv"); tw += (size_t)c->nphase * phase_stride(*c); tb += (size_t)c->Cout; }
    float *W, *Bv;
    HIPCHK(hipMalloc(&W, tw * sizeof(float))); HIPCHK(hipMalloc(&Bv, std::max<size_t>(tb, 4) * sizeof(float)));
    size_t ow = 0, ob = 0;
    for (size_t i = 0; i < cs.size(); i++) {
        ConvW *c = cs[i];
        const size_t nw = (size_t)c->nphase * phase_stride(*c);
        HIPCHK(hipMemcpy(W + ow, c->w, nw * sizeof(float), hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(Bv + ob, c->bias, (size_t)c->Cout * sizeof(float), hipMemcpyDeviceToDevice));
        (void)hipFree(c->w); (void)hipFree(c->bias);
        c->w = W + ow; c->bias = Bv + ob; c->owns = i == 0;
        ow += nw; ob += c->Cout;
    }
}

// ---------------------------------------------------------------------------------------
// op list ("plan") construction
// ---------------------------------------------------------------------------------------
struct ConvOpts {
    int act = ACT_NONE; float slope = 0.f; float scale = 1.f; bool accumulate = false;
    int pre_act = ACT_NONE; float pre_slope = 0.f;
    const float *res = nullptr; int res_cs = 0; long long res_bs = 0; int res_rs = 0;
    int m_off = 0, m_cnt = -1;   // output-row sub-range of the weight panel
    bool no_bias = false;
    bool glu = false;            // GLU-packed weight rows, gate fused into the epilogue (ModelSY flows)
    bool final_out = false;      // the chunk's last convolution: writes the caller's device buffer when the call provides one (Plan::cur_out)
    // LayerNorm folded into its neighbours (IgemmP::ln_*): this layer consumes a not-yet-normalised tensor (weights pre-scaled, wsum
    // per output row, optional (mean, rstd) output) / this layer's residual is LayerNorm(stored tensor) with published statistics
    const float *ln_wsum = nullptr; float *ln_stats_out = nullptr; int ln_rows = 0;
    const float *ln_stats_in = nullptr, *ln_g = nullptr, *ln_b = nullptr;
};

struct ProfEvent { hipEvent_t a, b; double flops; double bytes; int desc = -1; };   // bytes > 0: HBM-bound retrieval scan (flops = 0)

struct Plan;
typedef std::function<void(hipStream_t)> Op;

struct TapRec { std::string name; int rank; T1 t1; T2 t2; };

// ops are tagged with the HIP stream they run on: 0 = main, 1 = auxiliary (the RMVPE branch runs
// concurrently with ContentVec; fork/join through events, captured as parallel branches of the hipGraph)
struct OpList {
    std::vector<Op> v;
    std::vector<int> sid;    // stream of the op (0 = main, 1..3 auxiliary)
    std::vector<int> kind;   // 0 = op, 1 = fork(sid): stream sid waits for main, 2 = join(sid): main waits for stream sid
    // issue order (indices into v): host launch order decides which concurrent branch is fed first.  Eager launches follow it
    // exactly; a captured hipGraph is submitted branch by branch, the branch of the first created node first.
    std::vector<int> order_eager, order_graph;
    int cur = 0;
    void push_back(Op o) { v.push_back(std::move(o)); sid.push_back(cur); kind.push_back(0); }
    void fork(int s) { v.push_back(Op()); sid.push_back(s); kind.push_back(1); }
    void join(int s) { v.push_back(Op()); sid.push_back(s); kind.push_back(2); }
};

struct Plan {
    Arena arena;
    OpList ops;
    std::vector<TapRec> taps;
    // geometry
    int B = 1; size_t L = 0, frame16k = 0; uint32_t skip_head = 0, R = 0;
    int T = 0, Tm = 0, C = 0; size_t N = 0;
    bool with_index = false, with_taps = false;
    bool bucket = false;          // a plan of rvc_infer_batch_g: built for a subset of the streams on the gathered state block (rvc_engine::d_state_bucket)
    bool plain_plan = false;      // taps level 1: the explicit plan (LayerNorm launches, WaveNets layer by layer); level 2 taps the production plan
    int mode = 0;   // 0 infer, 1 hubert only, 2 pitch only
    // I/O tensors
    float *d_in = nullptr;  // [B][L]
    T1 cv_out, audio;
    float *d_f0 = nullptr;  // [B][Tm]
    float *d_feat = nullptr; // extract_feature output (1,2T+1,C)
    int *d_knn_idx = nullptr; float *d_knn_dist = nullptr;
    // profiling
    bool profile = false;
    std::vector<ProfEvent> prof;
    std::vector<std::string> descs;   // per-op description for rvc_debug_profile_dump
    size_t prof_used = 0;
    double igemm_flops = 0;
    int n_igemm = 0;
    std::vector<float *> owned_dev;   // plan-time repacked weights
    // timeline probe (RVC_STAMPS=1): one device timestamp per section boundary
    unsigned long long *d_stamps = nullptr; std::vector<std::string> stamp_names;
    // chunk pipelining (rvc_set_pipeline): plans alternate between two slots; ev_done marks the end of this plan's previous chunk
    int slot = 0; hipEvent_t ev_done = nullptr; bool ev_done_valid = false;
    // per-call pointers (eager launches): the kernels that read the input / write the audio take them at launch time, so a device-resident
    // caller needs no staging copy in front of the chunk and no copy behind it (a captured graph bakes pointers: it keeps d_in / audio)
    const float *cur_in = nullptr; float *cur_out = nullptr; long long cur_out_bs = 0;
    bool in_direct_ok = true, out_direct_ok = false;
    // graph
    hipGraphExec_t graph_exec = nullptr;
    ~Plan()
    {
        for (float *p : owned_dev) (void)hipFree(p);
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (ev_done) (void)hipEventDestroy(ev_done);
        for (auto &e : prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    }
};

// (launch_igemm2 / launch_igemm_v1 / launch_igemm_tiled: igemm_launch.h -- the template instantiations are separate translation units)

static unsigned long long *g_kprobe = nullptr;   // tuning build (-DRVC_KPROBE): destination of the per-wave phase stamps
static int g_last_waves = 0, g_last_wgs = 0;

// One stream, stride-1 1-D convolution with a long output: conv_tile_kernel (conv_tile.hip.h) stages the input rows once per workgroup.
// Builds the LDS-offset tables (k -> row * RS + tap column) from the layer's gather table and a work-item table that balances the
// unequal phases of a fused launch over the CUs (workgroup b lands on CU b % ncu: tests/tools/place_probe.hip).  false = not eligible.
static int g_ncu = 256;
static bool queue_conv_tile(Plan &pl, IgemmP &p, int B, const std::vector<int> &koff, const std::vector<PhaseD> &phv, double ksum, bool final_out)
{
    const int mode = test_opt_int("RVC_CONV_TILE", 1);       // test hook: 0 = off, 2 = wherever eligible; read per plan
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
    std::vector<int> order(rounds * nb * 2, -1);
    for (int j = 0; j < nb; j++)
        for (size_t r = 0; r < bins[j].size(); r++) { order[(r * nb + j) * 2] = items[bins[j][r]].code; order[(r * nb + j) * 2 + 1] = items[bins[j][r]].b; }
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
    { char d[200]; snprintf(d, sizeof d, "tile M=%d N=%d K=%d B=%d nph=%d tile=%dx%d items=%lld grid=%u lds=%zu pre=%d ksum=%.0f", p.M, p.N, p.K, B, p.nphase, BM, BN, nitems, grid.x, lds_max, (int)(p.pre_act != ACT_NONE), ksum); pl.descs.push_back(d); }
    const int desc_id = (int)pl.descs.size() - 1;
    const IgemmP pc = p;
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

// generic: the caller fills geometry (N, NW, strides, koff, phases); this picks the tile + split-K and queues the op
static void queue_igemm(Plan &pl, IgemmP p, int B, const std::vector<int> &koff, const std::vector<PhaseD> &phases, bool final_out = false)
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
    p.koff = pl.arena.upload(kb);
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
    const long long want_waves = (uneven && p.M >= 64) ? 2048 : 1024;
    for (int oi = 0; oi < 3 && !found; oi++) {
        const int c = order[oi];
        for (int ks = 1; ks <= 16; ks = ks == 1 ? 4 : ks * 2) {
            if (ks > 1 && (nchunks / ks < 4 || ks * kMF[c] * kNF[c] > 32)) break;
            if ((size_t)nchunks * 64 + (ks > 1 ? (size_t)ks * kMF[c] * kNF[c] * 1024 : 0) > 60 * 1024) break;
            const long long w = tiles(c) * ks;
            if (w > best_waves) { best_waves = w; cfg = c; wg_ks = ks; }
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
        }
    }
    // 48-row panels (ContentVec's grouped positional convolution: 16 groups of 48 channels, K = 6144 each): three 16-row fragments
    // exactly, instead of a 64-row tile with a quarter of its MFMAs on padding
    if (lds_cfg == 1 && p.M == 48 && !tune_env("RVC_NO_BM48")) lds_cfg = 6;
    // mid-size panels (M = 768 at 64 streams: 336 tiles of 128 x 128 balance badly over 256 CUs, and the register-direct 2 x 4 tile runs
    // at two waves per SIMD): 128 x 64 tiles of the 32x32x2 kernel, four waves stacked in M over one 64-column activation tile
    // (768 x 3072 projection at 64 streams: 361 -> 342 us; small, but the same kernel)
    static const long long g32_narrow_min = tune_env("RVC_G32_NARROW") ? atoll(tune_env("RVC_G32_NARROW")) : 500;     // 0 = off
    if (lds_cfg < 0 && !ln_fold && g32_narrow_min > 0 && !tune_env("RVC_NO_LDS_GEMM") && !p.glu && nchunks >= 2 && p.M >= 96 && (size_t)nchunks * 64 + 2 * 64 * 20 * 4 <= 60 * 1024) {
        const long long wgs = (long long)((p.M + 127) / 128) * ((p.N + 63) / 64) * B * p.nphase;
        if (wgs >= g32_narrow_min) lds_cfg = 7;
    }
    if (lds_cfg >= 0) {
        const int bm = lds_cfg == 7 ? 128 : (lds_cfg == 6 ? 48 : (lds_cfg % 3 == 0 ? 128 : (lds_cfg % 3 == 1 ? 64 : 32)));
        const int bn = lds_cfg == 7 ? 64 : ((lds_cfg != 6 && lds_cfg % 3 == 0) ? 128 : 256);
        p.ksplit = 1; p.chunks_per_split = nchunks;
        p.ntm = (p.M + bm - 1) / bm; p.ntn = (p.N + bn - 1) / bn;
        p.m_fast = p.fold_n ? p.ntm : 0;
        dim3 grid(p.ntm * p.ntn, B * p.nphase);
        const bool g32k = lds_cfg >= 3 && lds_cfg != 6;            // igemm32_kernel keeps its activation tile column-major, [2][bn][20]
        const size_t lds = (size_t)nchunks * 64 + (g32k ? (size_t)2 * bn * 20 * 4 : (size_t)2 * 16 * (bn + 4) * 4);
        g_last_wgs = (int)(grid.x * grid.y); g_last_waves = 4;
        const double flops = 2.0 * p.M * (double)p.N * ksum * B;
        pl.igemm_flops += flops; pl.n_igemm++;
        Plan *plp = &pl;
        const int lc = lds_cfg;
        { char d[160]; snprintf(d, sizeof d, "%s M=%d N=%d K=%d B=%d nph=%d tile=%dx%d grid=%ux%u", (lds_cfg >= 3 && lds_cfg != 6) ? "g32" : "lds", p.M, p.N, p.K, B, p.nphase, bm, bn, grid.x, grid.y); pl.descs.push_back(d); }
        const int desc_id = (int)pl.descs.size() - 1;
        pl.ops.push_back([=](hipStream_t s) {
            ProfEvent *pe = nullptr;
            if (plp->profile) {
                if (plp->prof_used == plp->prof.size()) { ProfEvent e; HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b)); e.flops = 0; e.bytes = 0; plp->prof.push_back(e); }
                pe = &plp->prof[plp->prof_used++]; pe->flops = flops; pe->bytes = 0; pe->desc = desc_id;
            }
            hipEvent_t ea = pe ? pe->a : nullptr, eb = pe ? pe->b : nullptr;
            if (final_out && plp->cur_out) { IgemmP q = p; q.y = plp->cur_out; q.y_bs = plp->cur_out_bs; launch_igemm_tiled(lc, pre, q, grid, lds, s, ea, eb); }
            else launch_igemm_tiled(lc, pre, p, grid, lds, s, ea, eb);
        });
        return;
    }
    // one stream, table-free layers of the ContentVec window (N = 111) that the size rule sends to lone 16 x 16 fragments: every B fragment costs
    // four dword gathers (9-12 clocks each on the CU's single vector-memory path) for ONE MFMA row block; two fragments along N per wave and eight
    // K shares halve the weight loads per MFMA (isolated: 768 x 3072 18.5 -> 14.9 us, 768 x 768 6.5 -> 5.7 us; in the chain: ContentVec -22 us)
    if (cfg == 0 && wg_ks == 4 && B == 1 && !p.fold_n && p.lin_cs4 && p.nphase == 1 && p.M >= 256 && p.N > 64 && p.N <= 128 && nchunks >= 32 && !p.ln_wsum && !getenv("RVC_NO_LIN_16x32")) { cfg = 1; wg_ks = 8; }
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
    if (p.ln_wsum || p.ln_stats_in) {
        // folded LayerNorm: one stream, in-workgroup K split (the statistics / the normalised residual live in that epilogue)
        if (lds_cfg >= 0 || B != 1 || p.fold_n || p.nphase != 1) throw std::logic_error("folded LayerNorm outside its supported launch shape");
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
    { char d[200]; snprintf(d, sizeof d, "reg M=%d N=%d K=%d B=%d nph=%d tile=%dx%d ks=%d mfast=%d grid=%ux%ux%u pre=%d lin=%d ksum=%.0f", p.M, p.N, p.K, B, p.nphase, 16 * kMF[cfg], 16 * kNF[cfg], wg_ks, p.m_fast, grid.x, grid.y, grid.z, (int)pre, (int)lin, ksum); pl.descs.push_back(d); }
    const int desc_id = (int)pl.descs.size() - 1;
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

static void fill_epilogue(IgemmP &p, const ConvW &cw, const ConvOpts &o)
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
static void add_conv1d(Plan &pl, const ConvW &cw, const T1 &x, const T1 &y, int stride, int pad, int dil, ConvOpts o = ConvOpts())
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
static void add_conv1d_multi(Plan &pl, const std::vector<const ConvW *> &cws, const T1 &x, bool x_grouped, const T1 &y,
                             const std::vector<int> &pads, const std::vector<int> &dils, ConvOpts o = ConvOpts(), bool res_grouped = true)
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
static void add_convT1d(Plan &pl, const ConvW &cw, const T1 &x, const T1 &y, int pad, ConvOpts o = ConvOpts())
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
static void add_conv2d(Plan &pl, const ConvW &cw, const T2 &x, const T2 &y, ConvOpts o = ConvOpts())
{
    if (x.H != y.H || x.W != y.W) throw ShapeError("conv2d shape mismatch");
    IgemmP p{};
    p.x = x.p; p.w = cw.w; p.y = y.p;
    p.M = cw.M; p.N = x.H * x.W; p.K = cw.Kp;
    p.NW = x.W; p.x_hs = x.ld; p.x_ws = 1; p.y_hm = 1; p.y_ws = 1; p.OW = y.W;
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
        float *dw = upload_fragments(panel, 1, cw.M, Kp3);
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

static void add_convT2d(Plan &pl, const ConvW &cw, const T2 &x, const T2 &y, ConvOpts o = ConvOpts())
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

static void add_layernorm(Plan &pl, const T1 &x, const float *g, const float *b)
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

static void add_stamp(Plan &pl, const char *name)
{
    const bool on = test_opt("RVC_STAMPS") != nullptr;
    if (!on) return;
    if (!pl.d_stamps) pl.d_stamps = reinterpret_cast<unsigned long long *>(pl.arena.floats(2 * 256));
    if (pl.stamp_names.size() >= 256) return;
    unsigned long long *slot = pl.d_stamps + pl.stamp_names.size();
    pl.stamp_names.push_back(name);
    pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, slot); });
}
static void add_tap(Plan &pl, const char *name, const T1 &t)
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
static void add_tap2(Plan &pl, const char *name, const T2 &t)
{
    add_stamp(pl, name);
    if (!pl.with_taps) return;
    TapRec r; r.name = name; r.rank = 2; r.t2 = t; pl.taps.push_back(r);   // RMVPE images are never overwritten
}

// ---------------------------------------------------------------------------------------
// models
// ---------------------------------------------------------------------------------------
struct DevVec { float *p = nullptr; };
static float *dv(const Blob &b, const std::string &name) { const BlobTensor &t = b.t(name); return upload_f(t.data, t.nelem); }

struct ModelCV {
    int conv_dim, embed, heads, ffn, run_layers, pos_k, pos_groups, out_dim;
    int conv_k[7], conv_s[7];
    ConvW conv[7], proj, pos, final_proj;
    float *gn_g, *gn_b, *ln0_g, *ln0_b, *encln_g, *encln_b;
    float *conv0_raw = nullptr;     // [conv_dim][conv_k0] row-major copy of the first conv (fused conv + GroupNorm + GELU kernel)
    // qkv_f / ff1_f: the same projections with the PRECEDING LayerNorm folded in (W' = W diag(g), b' = b + W beta, wsum = row sums of W'):
    // one-stream plans feed them the not-yet-normalised tensor and drop the LayerNorm launches (build_contentvec)
    struct Layer { ConvW qkv, o, ff1, ff2, qkv_f, ff1_f; float *qkv_wsum = nullptr, *ff1_wsum = nullptr; float *ln1_g, *ln1_b, *ln2_g, *ln2_b; };
    bool has_folded = false;
    ConvW proj_f; float *proj_wsum = nullptr;      // feature projection with the LayerNorm over the conv features folded in
    static ConvW fold_ln(const float *w, const float *bias, int M, int K, const float *g, const float *beta, float **wsum_dev)
    {
        std::vector<float> wf((size_t)M * K), bf(M), ws(M);
        for (int m = 0; m < M; m++) {
            double sb = bias ? bias[m] : 0.0, sw = 0.0;
            for (int k = 0; k < K; k++) {
                const float v = w[(size_t)m * K + k] * g[k];
                wf[(size_t)m * K + k] = v;
                sb += (double)w[(size_t)m * K + k] * beta[k];
                sw += v;
            }
            bf[m] = (float)sb; ws[m] = (float)sw;
        }
        *wsum_dev = upload_f(ws);
        return prep_conv(wf.data(), bf.data(), M, K, 1, 1);
    }
    std::vector<Layer> layers;
    std::vector<float *> owned;
    size_t weight_bytes = 0;
    explicit ModelCV(const Blob &b)
    {
        conv_dim = b.icfg("conv_dim"); embed = b.icfg("embed"); heads = b.icfg("heads"); ffn = b.icfg("ffn");
        run_layers = b.icfg("run_layers"); pos_k = b.icfg("pos_k"); pos_groups = b.icfg("pos_groups"); out_dim = b.icfg("out_dim");
        int cin = 1;
        for (int i = 0; i < 7; i++) {
            conv_k[i] = b.icfg(fmt("conv_k%d", i)); conv_s[i] = b.icfg(fmt("conv_s%d", i));
            conv[i] = prep_conv(b.w(fmt("cv.conv%d.w", i)), nullptr, conv_dim, cin, conv_k[i], 1);
            cin = conv_dim;
        }
        auto own = [&](const std::string &n) { float *p = dv(b, n); owned.push_back(p); return p; };
        conv0_raw = own("cv.conv0.w");
        gn_g = own("cv.gn.g"); gn_b = own("cv.gn.b"); ln0_g = own("cv.ln0.g"); ln0_b = own("cv.ln0.b");
        proj = prep_conv(b.w("cv.proj.w"), b.w("cv.proj.b"), embed, conv_dim, 1, 1);
        pos = prep_conv(b.w("cv.pos.w"), b.w("cv.pos.b"), embed, embed, pos_k, pos_groups);
        encln_g = own("cv.enc_ln.g"); encln_b = own("cv.enc_ln.b");
        const int E = embed;
        for (int l = 0; l < run_layers; l++) {
            Layer L;
            std::vector<float> w((size_t)3 * E * E), bb((size_t)3 * E);
            const char *nm[3] = {"q", "k", "v"};
            for (int j = 0; j < 3; j++) {
                memcpy(&w[(size_t)j * E * E], b.w(fmt("cv.l%d.", l) + nm[j] + ".w"), (size_t)E * E * 4);
                memcpy(&bb[(size_t)j * E], b.w(fmt("cv.l%d.", l) + nm[j] + ".b"), (size_t)E * 4);
            }
            L.qkv = prep_conv(w.data(), bb.data(), 3 * E, E, 1, 1);
            L.o = prep_conv(b.w(fmt("cv.l%d.o.w", l)), b.w(fmt("cv.l%d.o.b", l)), E, E, 1, 1);
            L.ff1 = prep_conv(b.w(fmt("cv.l%d.ff1.w", l)), b.w(fmt("cv.l%d.ff1.b", l)), ffn, E, 1, 1);
            L.ff2 = prep_conv(b.w(fmt("cv.l%d.ff2.w", l)), b.w(fmt("cv.l%d.ff2.b", l)), E, ffn, 1, 1);
            L.ln1_g = own(fmt("cv.l%d.ln1.g", l)); L.ln1_b = own(fmt("cv.l%d.ln1.b", l));
            L.ln2_g = own(fmt("cv.l%d.ln2.g", l)); L.ln2_b = own(fmt("cv.l%d.ln2.b", l));
            if (E >= 256 && E % 64 == 0 && ffn % 64 == 0 && !test_opt("RVC_NO_LN_FUSE")) {
                has_folded = true;
                L.ff1_f = fold_ln(b.w(fmt("cv.l%d.ff1.w", l)), b.w(fmt("cv.l%d.ff1.b", l)), ffn, E, b.w(fmt("cv.l%d.ln1.g", l)), b.w(fmt("cv.l%d.ln1.b", l)), &L.ff1_wsum);
                if (l > 0) L.qkv_f = fold_ln(w.data(), bb.data(), 3 * E, E, b.w(fmt("cv.l%d.ln2.g", l - 1)), b.w(fmt("cv.l%d.ln2.b", l - 1)), &L.qkv_wsum);
                else L.qkv_f = fold_ln(w.data(), bb.data(), 3 * E, E, b.w("cv.enc_ln.g"), b.w("cv.enc_ln.b"), &L.qkv_wsum);      // layer 0: the encoder's input LayerNorm
            }
            layers.push_back(L);
        }
        if (out_dim != E) final_proj = prep_conv(b.w("cv.final_proj.w"), b.w("cv.final_proj.b"), out_dim, E, 1, 1);
        if (has_folded && conv_dim % 64 == 0) proj_f = fold_ln(b.w("cv.proj.w"), b.w("cv.proj.b"), embed, conv_dim, b.w("cv.ln0.g"), b.w("cv.ln0.b"), &proj_wsum);
        weight_bytes = b.bytes();
    }
    ~ModelCV()
    {
        for (auto &c : conv) free_conv(c);
        free_conv(proj); free_conv(pos); free_conv(final_proj); free_conv(proj_f);
        if (proj_wsum) (void)hipFree(proj_wsum);
        for (auto &L : layers) {
            free_conv(L.qkv); free_conv(L.o); free_conv(L.ff1); free_conv(L.ff2); free_conv(L.qkv_f); free_conv(L.ff1_f);
            if (L.qkv_wsum) (void)hipFree(L.qkv_wsum);
            if (L.ff1_wsum) (void)hipFree(L.ff1_wsum);
        }
        for (float *p : owned) (void)hipFree(p);
    }
    int out_frames(size_t L) const
    {
        long long T = (long long)L;
        for (int i = 0; i < 7; i++) { if (T < conv_k[i]) return 0; T = (T - conv_k[i]) / conv_s[i] + 1; }
        return (int)T;
    }
};

struct ResBlockW { ConvW c1, c2, sc; bool has_sc = false; int ci = 0, co = 0; float *pair_bias = nullptr; };    // pair_bias: [c1.bias; sc.bias] for the fused c1 + shortcut launch
struct ModelRM {
    int en_out, levels, n_blocks, inter_layers, n_mels, gru_hidden, n_out;
    float bn_scale, bn_shift;
    std::vector<std::vector<ResBlockW>> enc, inter, dec;
    std::vector<ConvW> up;
    ConvW cnn, gru_ih, fc;
    float *whhT = nullptr, *bhh = nullptr, *whh = nullptr;
    size_t weight_bytes = 0;
    static ResBlockW block(const Blob &b, const std::string &pre, int ci, int co)
    {
        ResBlockW r; r.ci = ci; r.co = co;
        r.c1 = prep_conv(b.w(pre + "c1.w"), b.w(pre + "c1.b"), co, ci, 9, 1);
        r.c2 = prep_conv(b.w(pre + "c2.w"), b.w(pre + "c2.b"), co, co, 9, 1);
        if (ci != co) {
            r.has_sc = true; r.sc = prep_conv(b.w(pre + "sc.w"), b.w(pre + "sc.b"), co, ci, 1, 1);
            std::vector<float> pb(b.w(pre + "c1.b"), b.w(pre + "c1.b") + co);
            pb.insert(pb.end(), b.w(pre + "sc.b"), b.w(pre + "sc.b") + co);
            r.pair_bias = upload_f(pb);
        }
        return r;
    }
    explicit ModelRM(const Blob &b)
    {
        en_out = b.icfg("en_out"); levels = b.icfg("levels"); n_blocks = b.icfg("n_blocks"); inter_layers = b.icfg("inter_layers");
        n_mels = b.icfg("n_mels"); gru_hidden = b.icfg("gru_hidden"); n_out = b.icfg("n_out");
        bn_scale = b.w("rm.bn0")[0]; bn_shift = b.w("rm.bn0")[1];
        int ci = 1, co = en_out;
        for (int lv = 0; lv < levels; lv++) {
            std::vector<ResBlockW> v;
            for (int j = 0; j < n_blocks; j++) v.push_back(block(b, fmt("rm.enc%d.b%d.", lv, j), j == 0 ? ci : co, co));
            enc.push_back(v);
            ci = co; co *= 2;
        }
        for (int lv = 0; lv < inter_layers; lv++) {
            std::vector<ResBlockW> v;
            for (int j = 0; j < n_blocks; j++) v.push_back(block(b, fmt("rm.int%d.b%d.", lv, j), j == 0 ? (lv == 0 ? ci : co) : co, co));
            inter.push_back(v);
        }
        ci = co;
        for (int lv = 0; lv < levels; lv++) {
            co = ci / 2;
            up.push_back(prep_convT2d(b.w(fmt("rm.dec%d.up.w", lv)), b.w(fmt("rm.dec%d.up.b", lv)), ci, co));
            std::vector<ResBlockW> v;
            for (int j = 0; j < n_blocks; j++) v.push_back(block(b, fmt("rm.dec%d.b%d.", lv, j), j == 0 ? 2 * co : co, co));
            dec.push_back(v);
            ci = co;
        }
        cnn = prep_conv(b.w("rm.cnn.w"), b.w("rm.cnn.b"), 3, en_out, 9, 1);
        const int H = gru_hidden, I = 3 * n_mels;
        std::vector<float> wih((size_t)6 * H * I), bih((size_t)6 * H), wt((size_t)2 * H * 3 * H), bh((size_t)6 * H);
        const char *sfx[2] = {"f", "b"};
        for (int d = 0; d < 2; d++) {
            memcpy(&wih[(size_t)d * 3 * H * I], b.w(std::string("rm.gru.w_ih_") + sfx[d]), (size_t)3 * H * I * 4);
            memcpy(&bih[(size_t)d * 3 * H], b.w(std::string("rm.gru.b_ih_") + sfx[d]), (size_t)3 * H * 4);
            memcpy(&bh[(size_t)d * 3 * H], b.w(std::string("rm.gru.b_hh_") + sfx[d]), (size_t)3 * H * 4);
            const float *whh = b.w(std::string("rm.gru.w_hh_") + sfx[d]);
            for (int r = 0; r < 3 * H; r++) for (int j = 0; j < H; j++) wt[((size_t)d * H + j) * 3 * H + r] = whh[(size_t)r * H + j];
        }
        gru_ih = prep_conv(wih.data(), bih.data(), 6 * H, I, 1, 1);
        whhT = upload_f(wt); bhh = upload_f(bh);
        {
            std::vector<float> wr((size_t)2 * 3 * H * H);
            for (int d = 0; d < 2; d++) memcpy(&wr[(size_t)d * 3 * H * H], b.w(std::string("rm.gru.w_hh_") + sfx[d]), (size_t)3 * H * H * 4);
            whh = upload_f(wr);
        }
        fc = prep_conv(b.w("rm.fc.w"), b.w("rm.fc.b"), n_out, 2 * H, 1, 1);
        weight_bytes = b.bytes();
    }
    ~ModelRM()
    {
        auto fb = [](std::vector<std::vector<ResBlockW>> &vv) { for (auto &v : vv) for (auto &r : v) { free_conv(r.c1); free_conv(r.c2); free_conv(r.sc); if (r.pair_bias) (void)hipFree(r.pair_bias); } };
        fb(enc); fb(inter); fb(dec);
        for (auto &u : up) free_conv(u);
        free_conv(cnn); free_conv(gru_ih); free_conv(fc);
        if (whhT) (void)hipFree(whhT);
        if (bhh) (void)hipFree(bhh);
        if (whh) (void)hipFree(whh);
    }
};

struct ModelSY {
    int phone_dim, hidden, inter, filter, heads, enc_layers, enc_k, window, flow_n, wn_layers, wn_k, gin, up_init, n_ups, n_rb, n_rbd, sr;
    int up_rate[8], up_kernel[8], rb_k[8], rb_d[8];
    ConvW phone, proj;
    float *pitch_emb = nullptr;
    // qkv_f: the projection with the previous layer's second LayerNorm folded in (ModelCV::fold_ln); proj_f likewise for the last layer.
    // (The first LayerNorm of a layer feeds a 3-tap convolution with zero padding: padded positions are zero AFTER the norm, so it stays.)
    struct Layer {
        ConvW qkv, o, ff1, ff2, qkv_f; float *qkv_wsum = nullptr; float *rel_k, *rel_v, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    };
    ConvW proj_f; float *proj_wsum = nullptr; bool has_folded = false;
    std::vector<Layer> layers;
    struct Flow {
        ConvW pre, post; std::vector<ConvW> in, rs; bool flipped = false;
        // one stream: the WaveNet with its 1x1 res_skip layers composed into the following in-layers (compose_flows): host copies of the
        // layer weights in model order, and the composed panels
        std::vector<float> h_pre_w, h_pre_b, h_post_w, h_post_b;
        std::vector<std::vector<float>> h_in_w, h_in_b, h_rs_w, h_rs_b;
        ConvW pre1, postc, posth; std::vector<ConvW> inc; float *pair_bias = nullptr;      // postc: z_next rows, posth: the next flow's h0 rows
    };
    bool composed = false;
    std::vector<Flow> flows;
    ConvW dec_pre, dec_post;
    std::vector<ConvW> ups, ncs;
    std::vector<std::vector<std::vector<std::pair<ConvW, ConvW>>>> rbs;   // [stage][kernel][dilation] -> (c1, c2)
    float src_w, src_b;
    std::vector<float *> owned;
    size_t weight_bytes = 0;
    explicit ModelSY(const Blob &b)
    {
        phone_dim = b.icfg("phone_dim"); hidden = b.icfg("hidden"); inter = b.icfg("inter"); filter = b.icfg("filter"); heads = b.icfg("heads");
        enc_layers = b.icfg("enc_layers"); enc_k = b.icfg("enc_k"); window = b.icfg("window"); flow_n = b.icfg("flow_n");
        wn_layers = b.icfg("wn_layers"); wn_k = b.icfg("wn_k"); gin = b.icfg("gin"); up_init = b.icfg("up_init"); n_ups = b.icfg("n_ups");
        n_rb = b.icfg("n_rb"); n_rbd = b.icfg("n_rbd"); sr = b.icfg("sr");
        for (int i = 0; i < n_ups; i++) { up_rate[i] = b.icfg(fmt("up_rate%d", i)); up_kernel[i] = b.icfg(fmt("up_kernel%d", i)); }
        for (int j = 0; j < n_rb; j++) rb_k[j] = b.icfg(fmt("rb_k%d", j));
        for (int m = 0; m < n_rbd; m++) rb_d[m] = b.icfg(fmt("rb_d%d", m));
        auto own = [&](const std::string &n) { float *p = dv(b, n); owned.push_back(p); return p; };
        const int H = hidden, G = gin;
        const float *g = b.w("sy.g");
        phone = prep_conv(b.w("sy.enc.phone.w"), b.w("sy.enc.phone.b"), H, phone_dim, 1, 1);
        pitch_emb = own("sy.enc.pitch_emb");
        for (int l = 0; l < enc_layers; l++) {
            Layer L;
            std::vector<float> w((size_t)3 * H * H), bb((size_t)3 * H);
            const char *nm[3] = {"q", "k", "v"};
            for (int j = 0; j < 3; j++) {
                memcpy(&w[(size_t)j * H * H], b.w(fmt("sy.enc.l%d.", l) + nm[j] + ".w"), (size_t)H * H * 4);
                memcpy(&bb[(size_t)j * H], b.w(fmt("sy.enc.l%d.", l) + nm[j] + ".b"), (size_t)H * 4);
            }
            L.qkv = prep_conv(w.data(), bb.data(), 3 * H, H, 1, 1);
            L.o = prep_conv(b.w(fmt("sy.enc.l%d.o.w", l)), b.w(fmt("sy.enc.l%d.o.b", l)), H, H, 1, 1);
            L.ff1 = prep_conv(b.w(fmt("sy.enc.l%d.ff1.w", l)), b.w(fmt("sy.enc.l%d.ff1.b", l)), filter, H, enc_k, 1);
            L.ff2 = prep_conv(b.w(fmt("sy.enc.l%d.ff2.w", l)), b.w(fmt("sy.enc.l%d.ff2.b", l)), H, filter, enc_k, 1);
            L.rel_k = own(fmt("sy.enc.l%d.rel_k", l)); L.rel_v = own(fmt("sy.enc.l%d.rel_v", l));
            L.ln1_g = own(fmt("sy.enc.l%d.ln1.g", l)); L.ln1_b = own(fmt("sy.enc.l%d.ln1.b", l));
            L.ln2_g = own(fmt("sy.enc.l%d.ln2.g", l)); L.ln2_b = own(fmt("sy.enc.l%d.ln2.b", l));
            if (H >= 128 && H % 16 == 0 && !test_opt("RVC_NO_LN_FUSE")) {
                has_folded = true;
                if (l > 0) L.qkv_f = ModelCV::fold_ln(w.data(), bb.data(), 3 * H, H, b.w(fmt("sy.enc.l%d.ln2.g", l - 1)), b.w(fmt("sy.enc.l%d.ln2.b", l - 1)), &L.qkv_wsum);
            }
            layers.push_back(L);
        }
        proj = prep_conv(b.w("sy.enc.proj.w"), b.w("sy.enc.proj.b"), 2 * inter, H, 1, 1);
        if (has_folded)
            proj_f = ModelCV::fold_ln(b.w("sy.enc.proj.w"), b.w("sy.enc.proj.b"), 2 * inter, H, b.w(fmt("sy.enc.l%d.ln2.g", enc_layers - 1)), b.w(fmt("sy.enc.l%d.ln2.b", enc_layers - 1)), &proj_wsum);
        const int half = inter / 2;
        for (int i = 0; i < flow_n; i++) {
            Flow F;
            // Flip layers are folded into the weights: the latent stays in its physical channel order and a flow that sees it
            // flipped (inference runs flip -> coupling from the last flow to the first: flow i after flow_n - i flips) reads its
            // x0 from the upper half with reversed input columns and writes x1 to the lower half with reversed output rows
            F.flipped = ((flow_n - i) & 1) != 0;
            {
                // rows H..2H are zero: the launch also clears the skip accumulator that sits behind hh in one tensor
                std::vector<float> w((size_t)2 * H * half, 0.f), bb((size_t)2 * H, 0.f);
                const float *pw = b.w(fmt("sy.flow%d.pre.w", i)), *pb = b.w(fmt("sy.flow%d.pre.b", i));
                for (int r = 0; r < H; r++) {
                    bb[r] = pb[r];
                    for (int q = 0; q < half; q++) w[(size_t)r * half + q] = pw[(size_t)r * half + (F.flipped ? half - 1 - q : q)];
                }
                F.pre = prep_conv(w.data(), bb.data(), 2 * H, half, 1, 1);
                F.h_pre_w.assign(w.begin(), w.begin() + (size_t)H * half); F.h_pre_b.assign(bb.begin(), bb.begin() + H);
            }
            // speaker conditioning is a load-time constant (sid baked, rvc.rs:186-187): fold cond(g) into the in-layer biases
            const float *cw = b.w(fmt("sy.flow%d.cond.w", i)), *cb = b.w(fmt("sy.flow%d.cond.b", i));
            for (int j = 0; j < wn_layers; j++) {
                std::vector<float> bias(2 * H);
                const float *ib = b.w(fmt("sy.flow%d.in%d.b", i, j));
                for (int r = 0; r < 2 * H; r++) {
                    float a = cb[j * 2 * H + r];
                    for (int q = 0; q < G; q++) a += cw[(size_t)(j * 2 * H + r) * G + q] * g[q];
                    bias[r] = ib[r] + a;
                }
                {
                    // GLU row packing (kernels.hip.h glu_store): packed row f*16 + kq*4 + r <- channel f*8 + kq*2 + (r&1), sigmoid half for r >= 2
                    if (H % 8 != 0) throw std::runtime_error("synth hidden size must be a multiple of 8");
                    const float *iw = b.w(fmt("sy.flow%d.in%d.w", i, j));
                    const size_t Kin = (size_t)H * wn_k;
                    std::vector<float> w((size_t)2 * H * Kin), pb((size_t)2 * H);
                    for (int r = 0; r < 2 * H; r++) {
                        const int f = r >> 4, kq = (r & 15) >> 2, rr = r & 3;
                        const int src = f * 8 + kq * 2 + (rr & 1) + (rr >= 2 ? H : 0);
                        memcpy(&w[(size_t)r * Kin], iw + (size_t)src * Kin, Kin * sizeof(float));
                        pb[r] = bias[src];
                    }
                    F.in.push_back(prep_conv(w.data(), pb.data(), 2 * H, H, wn_k, 1));
                    F.h_in_w.emplace_back(iw, iw + (size_t)2 * H * Kin); F.h_in_b.push_back(bias);
                }
                int rs_c = j < wn_layers - 1 ? 2 * H : H;
                F.rs.push_back(prep_conv(b.w(fmt("sy.flow%d.rs%d.w", i, j)), b.w(fmt("sy.flow%d.rs%d.b", i, j)), rs_c, H, 1, 1));
                { const float *rw = b.w(fmt("sy.flow%d.rs%d.w", i, j)), *rb = b.w(fmt("sy.flow%d.rs%d.b", i, j)); F.h_rs_w.emplace_back(rw, rw + (size_t)rs_c * H); F.h_rs_b.emplace_back(rb, rb + rs_c); }
            }
            {
                const float *pw = b.w(fmt("sy.flow%d.post.w", i)), *pb = b.w(fmt("sy.flow%d.post.b", i));
                std::vector<float> w((size_t)half * H), bb(half);
                for (int r = 0; r < half; r++) {
                    const int src = F.flipped ? half - 1 - r : r;
                    memcpy(&w[(size_t)r * H], pw + (size_t)src * H, (size_t)H * sizeof(float));
                    bb[r] = pb[src];
                }
                F.post = prep_conv(w.data(), bb.data(), half, H, 1, 1);
                F.h_post_w = w; F.h_post_b = bb;
            }
            flows.push_back(F);
        }
        // composed WaveNets (one to eight streams): built with the model, 20 tasks on the host's cores, so that no first chunk pays for them
        if (hidden % 16 == 0 && inter == hidden && !test_opt("RVC_NO_WN_COMPOSE")) compose_flows();
        {
            std::vector<float> bias(up_init);
            const float *cw = b.w("sy.dec.cond.w"), *cb = b.w("sy.dec.cond.b"), *pb = b.w("sy.dec.pre.b");
            for (int c = 0; c < up_init; c++) { float a = cb[c]; for (int q = 0; q < G; q++) a += cw[(size_t)c * G + q] * g[q]; bias[c] = pb[c] + a; }
            dec_pre = prep_conv(b.w("sy.dec.pre.w"), bias.data(), up_init, inter, 7, 1);
        }
        int c = up_init;
        for (int i = 0; i < n_ups; i++) {
            int co = c / 2;
            ups.push_back(prep_convT1d(b.w(fmt("sy.dec.up%d.w", i)), b.w(fmt("sy.dec.up%d.b", i)), c, co, up_kernel[i], up_rate[i]));
            int sf = 1; for (int q = i + 1; q < n_ups; q++) sf *= up_rate[q];
            int nk = i + 1 < n_ups ? 2 * sf : 1;
            ncs.push_back(prep_conv(b.w(fmt("sy.dec.nc%d.w", i)), b.w(fmt("sy.dec.nc%d.b", i)), co, 1, nk, 1));
            std::vector<std::vector<std::pair<ConvW, ConvW>>> stage;
            for (int j = 0; j < n_rb; j++) {
                std::vector<std::pair<ConvW, ConvW>> chain;
                for (int m = 0; m < n_rbd; m++) {
                    ConvW c1 = prep_conv(b.w(fmt("sy.dec.rb%d_%d.c1_%d.w", i, j, m)), b.w(fmt("sy.dec.rb%d_%d.c1_%d.b", i, j, m)), co, co, rb_k[j], 1);
                    ConvW c2 = prep_conv(b.w(fmt("sy.dec.rb%d_%d.c2_%d.w", i, j, m)), b.w(fmt("sy.dec.rb%d_%d.c2_%d.b", i, j, m)), co, co, rb_k[j], 1);
                    chain.push_back({c1, c2});
                }
                stage.push_back(chain);
            }
            rbs.push_back(stage);
            // the n_rb chains' q-th convs run as phases of one launch: their weights share an allocation
            for (int m = 0; m < n_rbd && n_rb > 1; m++) {
                std::vector<ConvW *> a, bb;
                for (int j = 0; j < n_rb; j++) { a.push_back(&rbs.back()[j][m].first); bb.push_back(&rbs.back()[j][m].second); }
                merge_convs(a); merge_convs(bb);
            }
            c = co;
        }
        dec_post = prep_conv(b.w("sy.dec.post.w"), nullptr, 1, c, 7, 1);
        src_w = b.w("sy.src")[0]; src_b = b.w("sy.src")[1];
        weight_bytes = b.bytes();
        // the f0 / feature frame rate is 100 Hz (rvc.rs:153, 160 samples @16 kHz): a synthesizer whose hop is not sr / 100 would
        // return audio of the wrong length without any error (e.g. an import that guessed the first upsample rate)
        if (sr != 100 * upp()) throw std::runtime_error(fmt("synthesizer: sr %d", sr) + fmt(" != 100 * prod(upsample rates) = %d", 100 * upp()));
    }
    // One stream: every flow's WaveNet runs 4 x (gated k-tap in-layer, 1x1 res_skip layer) -- ten dependent launches of a 21-column window.  The
    // res_skip layers are linear, so they are composed into what follows them (exactly, in double, when the model is loaded):
    //   x_j = h0 + sum_{i<j} (R_i a_i + r_i)                      =>  in_j(x_j) = W_j * [1 | h0 | a_0 .. a_{j-1}]   with W_j(a_i) = W_j o R_i
    //   post(skip) = P (sum_j S_j a_j + s_j) + p                   =>  one 1x1 layer over [a_0 .. a_{n-1}]
    // (R_i / S_i: the residual / skip rows of res_skip layer i; the constant r_i rides on a row of ones -- zero in the halo, like the zero padding
    // the in-layer sees -- so the edges of the window stay exact.)  The latent z rides in the same tensor ([ones | h0 | a_0 .. | z]), and a flow's
    // post layer and the NEXT flow's pre layer become one 1x1 layer over [a_0 .. a_{n-1} | z] that writes h0_next and z_next into the other of two
    // such tensors (two phases of one launch: same input, two outputs): five launches per flow (+ one pre at the start) instead of ten.
    void compose_flows()
    {
        if (composed) return;
        const int H = hidden, I = inter, half = inter / 2, K5 = wn_k, nl = wn_layers;
        const int nfl = (int)flows.size();
        // per flow: x0 / x1 rows of the latent, the full-latent pre weights [H][I] (zero on the x1 half), post rows on the x1 half
        auto x1_row0 = [&](const Flow &F) { return F.flipped ? 0 : half; };
        auto x0_row0 = [&](const Flow &F) { return F.flipped ? half : 0; };
        std::vector<std::vector<std::vector<float>>> WJ(nfl), BJ(nfl);
        std::vector<std::vector<float>> WM(nfl), BM(nfl), WH(nfl), BH(nfl), WP1(nfl), BP1(nfl);
        // in-layer j of flow fi over [ones16 | h0 | a_0 .. a_{j-1}] (one task each: 1.7 GFLOP of double arithmetic in all, spread over the host's cores)
        auto in_layer = [&](int fi, int j) {
            Flow &F = flows[fi];
            {
                const int Cin = 16 + H * (j + 1), a0 = 16 + H;
                std::vector<double> w((size_t)2 * H * Cin * K5, 0.0);
                const float *W5 = F.h_in_w[j].data();                 // [2H][H][K5], model row order
                for (int o = 0; o < 2 * H; o++)
                    for (int mm = 0; mm < H; mm++)
                        for (int t = 0; t < K5; t++) w[((size_t)o * Cin + 16 + mm) * K5 + t] = W5[((size_t)o * H + mm) * K5 + t];
                std::vector<double> acc(H);
                for (int i = 0; i < j; i++) {
                    const float *Rr = F.h_rs_w[i].data(), *rb = F.h_rs_b[i].data();      // rows 0..H: the residual part
                    for (int o = 0; o < 2 * H; o++)
                        for (int t = 0; t < K5; t++) {
                            std::fill(acc.begin(), acc.end(), 0.0);
                            double one = 0.0;
                            for (int mm = 0; mm < H; mm++) {
                                const double v = W5[((size_t)o * H + mm) * K5 + t];
                                const float *Rm = Rr + (size_t)mm * H;
                                for (int c = 0; c < H; c++) acc[c] += v * Rm[c];
                                one += v * rb[mm];
                            }
                            for (int c = 0; c < H; c++) w[((size_t)o * Cin + a0 + H * i + c) * K5 + t] = acc[c];
                            w[((size_t)o * Cin) * K5 + t] += one;
                        }
                }
                std::vector<float> wp((size_t)2 * H * Cin * K5), pb((size_t)2 * H);
                for (int r = 0; r < 2 * H; r++) {                      // GLU row packing, as for the plain in-layers
                    const int f = r >> 4, kq = (r & 15) >> 2, rr = r & 3;
                    const int src = f * 8 + kq * 2 + (rr & 1) + (rr >= 2 ? H : 0);
                    for (size_t q = 0; q < (size_t)Cin * K5; q++) wp[(size_t)r * Cin * K5 + q] = (float)w[(size_t)src * Cin * K5 + q];
                    pb[r] = F.h_in_b[j][src];
                }
                WJ[fi][j] = std::move(wp); BJ[fi][j] = std::move(pb);
            }
        };
        auto one_flow = [&](int fi) {
            Flow &F = flows[fi];
            // first launch of the flow when it has no predecessor in processing order: h0 = pre(x0) from the full latent
            WP1[fi].assign((size_t)H * I, 0.f); BP1[fi] = F.h_pre_b;
            for (int r = 0; r < H; r++) for (int q = 0; q < half; q++) WP1[fi][(size_t)r * I + x0_row0(F) + q] = F.h_pre_w[(size_t)r * half + q];
            // composed post over [a_0 .. a_{n-1}]: P (sum_j S_j a_j + s_j) + p, rows = the x1 half in its physical order
            const int KA = nl * H, Kin = KA + I;                       // last launch's input: [a_0 .. a_{n-1} | z]
            std::vector<double> pc((size_t)half * KA, 0.0), pcb(half, 0.0);
            for (int r = 0; r < half; r++) {
                double bacc = F.h_post_b[r];
                for (int j = 0; j < nl; j++) {
                    const int row0 = j < nl - 1 ? H : 0;               // skip rows of res_skip layer j
                    const float *S = F.h_rs_w[j].data() + (size_t)row0 * H, *sb = F.h_rs_b[j].data() + row0;
                    for (int h = 0; h < H; h++) {
                        const double v = F.h_post_w[(size_t)r * H + h];
                        for (int c = 0; c < H; c++) pc[(size_t)r * KA + (size_t)j * H + c] += v * S[(size_t)h * H + c];
                        bacc += v * sb[h];
                    }
                }
                pcb[r] = bacc;
            }
            // last launch of the flow, input [A | z] (K = n H + I): z_next = z - [0 ; post(A)] on the x1 rows, and for the next flow in processing order
            //   h0_next = pre_next(z_next) = Wn z - Wn[:, x1 rows] post(A) + (bn - Wn[:, x1 rows] p)        (two phases of one launch: same input, two outputs)
            const bool has_next = fi > 0;
            const int r1 = x1_row0(F);
            std::vector<double> wz((size_t)I * Kin, 0.0), bz(I, 0.0);
            for (int c = 0; c < I; c++) wz[(size_t)c * Kin + KA + c] = 1.0;
            for (int r = 0; r < half; r++) {
                for (int q = 0; q < KA; q++) wz[(size_t)(r1 + r) * Kin + q] = -pc[(size_t)r * KA + q];
                bz[r1 + r] = -pcb[r];
            }
            WM[fi].resize(wz.size()); BM[fi].resize(I);
            for (size_t q = 0; q < wz.size(); q++) WM[fi][q] = (float)wz[q];
            for (int r = 0; r < I; r++) BM[fi][r] = (float)bz[r];
            if (has_next) {
                const Flow &N = flows[fi - 1];
                std::vector<double> wh((size_t)H * Kin, 0.0);
                WH[fi].resize(wh.size()); BH[fi].resize(H);
                for (int r = 0; r < H; r++) {
                    double bacc = N.h_pre_b[r];
                    for (int q = 0; q < half; q++) {
                        const double v = N.h_pre_w[(size_t)r * half + q];
                        const int zc = x0_row0(N) + q;                 // latent row this weight multiplies
                        wh[(size_t)r * Kin + KA + zc] += v;
                        if (zc >= r1 && zc < r1 + half) {
                            const int pr = zc - r1;
                            for (int c = 0; c < KA; c++) wh[(size_t)r * Kin + c] -= v * pc[(size_t)pr * KA + c];
                            bacc -= v * pcb[pr];
                        }
                    }
                    BH[fi][r] = (float)bacc;
                }
                for (size_t q = 0; q < wh.size(); q++) WH[fi][q] = (float)wh[q];
            }
        };
        for (int i = 0; i < nfl; i++) { WJ[i].resize(nl); BJ[i].resize(nl); }
        std::vector<std::thread> th;
        for (int i = 0; i < nfl; i++) {
            th.emplace_back([&, i]() { one_flow(i); });
            for (int j = 0; j < nl; j++) th.emplace_back([&, i, j]() { in_layer(i, j); });
        }
        for (auto &t : th) t.join();
        for (int i = 0; i < nfl; i++) {
            Flow &F = flows[i];
            F.pre1 = prep_conv(WP1[i].data(), BP1[i].data(), H, I, 1, 1);
            for (int j = 0; j < nl; j++) F.inc.push_back(prep_conv(WJ[i][j].data(), BJ[i][j].data(), 2 * H, 16 + H * (j + 1), K5, 1));
            F.postc = prep_conv(WM[i].data(), BM[i].data(), I, nl * H + I, 1, 1);
            if (i > 0) {
                F.posth = prep_conv(WH[i].data(), BH[i].data(), H, nl * H + I, 1, 1);
                std::vector<float> pb(BH[i]); pb.insert(pb.end(), BM[i].begin(), BM[i].end());
                F.pair_bias = upload_f(pb); owned.push_back(F.pair_bias);
            }
        }
        composed = true;
    }
    ~ModelSY()
    {
        free_conv(phone); free_conv(proj); free_conv(dec_pre); free_conv(dec_post);
        for (auto &L : layers) { free_conv(L.qkv); free_conv(L.o); free_conv(L.ff1); free_conv(L.ff2); free_conv(L.qkv_f); if (L.qkv_wsum) (void)hipFree(L.qkv_wsum); }
        free_conv(proj_f); if (proj_wsum) (void)hipFree(proj_wsum);
        for (auto &F : flows) { free_conv(F.pre); free_conv(F.post); for (auto &c : F.in) free_conv(c); for (auto &c : F.rs) free_conv(c); if (composed) { free_conv(F.pre1); free_conv(F.postc); free_conv(F.posth); for (auto &c : F.inc) free_conv(c); } }
        for (auto &c : ups) free_conv(c);
        for (auto &c : ncs) free_conv(c);
        for (auto &s : rbs) for (auto &ch : s) for (auto &pr : ch) { free_conv(pr.first); free_conv(pr.second); }
        for (float *p : owned) (void)hipFree(p);
    }
    int upp() const { int u = 1; for (int i = 0; i < n_ups; i++) u *= up_rate[i]; return u; }
};

// ---------------------------------------------------------------------------------------
// the engine
// ---------------------------------------------------------------------------------------
}  // namespace rvc

namespace rvc { struct StreamSet; }
using namespace rvc;

struct rvc_engine {
    std::string data_path, err;
    int device = 0;
    // aux streams: 1 = f0 branch, 2 = side work (NSF source), 3 = ContentVec branch when the CUs are partitioned.  Four streams
    // in total: the runtime multiplexes streams onto 4 hardware queues, a fifth stream would share (and serialise with) another.
    hipStream_t stream = nullptr, aux[3] = {nullptr, nullptr, nullptr};
    struct rvc::StreamSet *sset = nullptr;             // the engine's streams are borrowed from a per-device pool (never destroyed)
    bool partition_ok = false, partitioned = false;   // CU-masked streams available / currently in use (n_streams <= 4)
    hipEvent_t ev_fork[3] = {nullptr, nullptr, nullptr}, ev_join[3] = {nullptr, nullptr, nullptr};
    std::unique_ptr<ModelCV> cv;
    std::unique_ptr<ModelRM> rm;
    std::unique_ptr<ModelSY> sy;
    // constants for the mel front end
    float *d_window = nullptr, *d_twiddle = nullptr, *d_basis = nullptr; int *d_band = nullptr;
    // retrieval index
    float *d_index = nullptr, *d_indexT = nullptr, *d_indexF = nullptr, *d_ynorm = nullptr, *d_nhn = nullptr; size_t index_n = 0, index_dim = 0; bool index_owned = true;
    float index_rate = 0.f;
    float index_prep_ms = 0.f;                                  // device-side repack + norms of the last index load
    double bcast_ms[3] = {0, 0, 0}; int bcast_ranks = 0;         // last rvc_index_broadcast: communicator set-up, broadcast, repack (ms); ranks the communicator reports
    // streams
    int n_streams = 1;
    StreamState *d_state = nullptr;
    StreamState *d_state_bucket = nullptr; int *d_bucket_idx = nullptr;      // rvc_infer_batch_g: the states of one geometry bucket, gathered contiguously, and their stream numbers
    CallParams *d_cp = nullptr, *h_cp = nullptr;   // h_cp: ring of 64 pinned blocks, one per call (an async copy reads its block later)
    unsigned cp_slot = 0; hipEvent_t ev_cp = nullptr;
    uint32_t seed = 0, stream_id0 = 0;
    // plans (keyed by geometry)
    std::vector<std::unique_ptr<Plan>> plans;
    Plan *last_plan = nullptr;
    int taps_on = 0;               // 0 off, 1 taps on the explicit plan, 2 taps on the production plan (rvc_enable_taps)
    bool profile_on = false, use_graph = false;
    // offline throughput mode: consecutive unsynchronised infer_device calls overlap chunk i+1's two front branches with chunk i's
    // synthesizer (two plan slots; the branch streams are ordered by events instead of forking from the main stream)
    bool pipeline = false, pipe_now = false; int pipe_slot = 0; hipEvent_t ev_in = nullptr; const void *pipe_input = nullptr; size_t pipe_input_bytes = 0;
    std::vector<float> pushed_up; uint32_t pushed_seed = 0; bool pushed_valid = false;      // what the device holds: per-stream multipliers, seed
    float *h_up = nullptr; unsigned up_slot = 0;        // pinned ring of 8 blocks of 4096 per-stream multipliers (async strided copies read them later)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    size_t last_knn_rows = 0;
    int *h_status = nullptr;        // pinned, one word per stream (up to 4096)
    bool status_queued = false;     // an async copy of the status words is already in the stream in front of the caller's sync
};

namespace rvc {

static void set_device(rvc_engine *e) { HIPCHK(hipSetDevice(e->device)); }

static void init_kernel_attrs()
{
    // per device: function attributes belong to the device that is current when they are set (an engine on a second GPU of one process needs its own)
    static std::mutex mu; static bool done[64] = {};
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64) { if (done[dev]) return; done[dev] = true; }
    HIPCHK(hipFuncSetAttribute((const void *)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)layernorm_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));   // + 1.3 KB static
    HIPCHK(hipFuncSetAttribute((const void *)gru_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)knn_select_blend_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));   // + ~5 KB static
    HIPCHK(hipFuncSetAttribute((const void *)knn_dot_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    conv_tile_prepare_device();
    HIPCHK(hipFuncSetAttribute((const void *)relpos_attention_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)relpos_attention_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}

static void init_constants(rvc_engine *e)
{
    init_kernel_attrs();
    // periodic Hann, f64 cosine cast to f32 then 0.5*(1-c) in f32 (rmvpe.rs:33-37, Q9)
    std::vector<float> win(1024), tw(1024), basis((size_t)128 * 513);
    for (int i = 0; i < 1024; i++) { float c = (float)cos(2.0 * M_PI * (double)i / 1024.0); win[i] = 0.5f * (1.0f - c); }
    for (int j = 0; j < 512; j++) { double a = -2.0 * M_PI * (double)j / 1024.0; tw[2 * j] = (float)cos(a); tw[2 * j + 1] = (float)sin(a); }
    // mel_spec::mel::mel(16000, 1024, 128, 30, 8000, htk=true, norm=true) (rmvpe.rs:146-148,220): HTK mel scale,
    // triangular filters, Slaney area normalisation, computed in f64 then cast to f32
    const int nb = 513, nm = 128;
    const double sr = 16000.0, fmin = 30.0, fmax = 8000.0;
    std::vector<double> melf(nm + 2);
    double mlo = 2595.0 * log10(1.0 + fmin / 700.0), mhi = 2595.0 * log10(1.0 + fmax / 700.0);
    for (int i = 0; i < nm + 2; i++) { double m = mlo + (mhi - mlo) * (double)i / (double)(nm + 1); melf[i] = 700.0 * (pow(10.0, m / 2595.0) - 1.0); }
    for (int i = 0; i < nm; i++) {
        double fd0 = melf[i + 1] - melf[i], fd1 = melf[i + 2] - melf[i + 1], enorm = 2.0 / (melf[i + 2] - melf[i]);
        for (int j = 0; j < nb; j++) {
            double f = (sr / 2.0) * (double)j / (double)(nb - 1);
            double lower = -(melf[i] - f) / fd0, upper = (melf[i + 2] - f) / fd1;
            double v = std::max(0.0, std::min(lower, upper));
            basis[(size_t)i * nb + j] = (float)(v * enorm);
        }
    }
    e->d_window = upload_f(win); e->d_twiddle = upload_f(tw); e->d_basis = upload_f(basis);
    std::vector<int> band(256);
    for (int i = 0; i < nm; i++) {
        int lo = nb, hi = 0;
        for (int j = 0; j < nb; j++) if (basis[(size_t)i * nb + j] != 0.f) { lo = std::min(lo, j); hi = std::max(hi, j + 1); }
        if (hi <= lo) { lo = 0; hi = 0; }
        band[2 * i] = lo; band[2 * i + 1] = hi;
    }
    HIPCHK(hipMalloc(&e->d_band, band.size() * sizeof(int)));
    HIPCHK(hipMemcpy(e->d_band, band.data(), band.size() * sizeof(int), hipMemcpyHostToDevice));
}

static void reset_state(rvc_engine *e)
{
    std::vector<StreamState> st(e->n_streams);
    for (int b = 0; b < e->n_streams; b++) { memset(&st[b], 0, sizeof(StreamState)); st[b].stream_id = e->stream_id0 + (uint32_t)b; }
    HIPCHK(hipMemcpy(e->d_state, st.data(), sizeof(StreamState) * e->n_streams, hipMemcpyHostToDevice));
    e->pushed_valid = false;       // (the per-stream multipliers live in StreamState)
}

// CU partition for the two concurrent branches of a chunk at low stream counts: the f0 branch (RMVPE: ~140 short weight-streaming
// kernels) gets 1/8 of the CUs, ContentVec the rest.  Sharing CUs slows the f0 branch by ~35 % (measured).  With
// many streams every kernel fills the chip and the streams are plain.
//
// CU-masked streams are never destroyed: the runtime recycles the hardware queue of a destroyed stream, mask included, for the next
// stream it creates -- an engine created after another one had been destroyed then ran its MAIN stream on a partition (measured:
// 2.41 -> 2.79-3.1 ms per chunk for the second engine of a process; no effect without masks).  Masked pairs live in a per-device
// pool for the life of the process; engines borrow a pair and hand it back.
// Round 3: the same holds, less visibly, for plain streams: every stream an engine creates takes the next hardware queue, and after a
// few create / destroy cycles in one process (the bench's sub-configurations, a host that reloads engines) a new engine's main stream
// can share a queue with its own masked f0 stream -- the two then serialise (measured in one bench process: v1 2.15 -> 2.71 ms, two
// streams 3.35 -> 4.79 ms per chunk for engines created after several others had come and gone).  So ALL streams of an engine come
// from a per-device pool of complete sets (main + three plain auxiliaries + the masked pair), created together -- six consecutive
// hardware queues -- and never destroyed; an engine borrows a set and hands it back.
struct StreamSet { int device; hipStream_t main, plain[3], f0, cv; bool masked_ok, in_use; };
static std::mutex g_pool_mu;
static std::vector<StreamSet *> g_pool;

static StreamSet *acquire_stream_set(int device, int ncu, int nf0, bool want_masks)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (StreamSet *m : g_pool)
        if (m->device == device && !m->in_use) { m->in_use = true; return m; }
    StreamSet *m = new StreamSet{device, nullptr, {nullptr, nullptr, nullptr}, nullptr, nullptr, false, true};
    // creation order = the order a lone engine always used: main, the masked pair, the side stream (the first bench run of this
    // pool created all plain streams first: 2.17 -> 2.80 ms per chunk -- queue assignment follows creation order); the two plain
    // streams that stand in for the masked pair at more than 4 streams are created when such an engine first borrows the set
    // tuning aid: RVC_STREAM_ORDER = a permutation of "mfcs" (main, f0 masked, ContentVec masked, side), RVC_STREAM_PAD = dummy streams first
    const char *ord = tune_env("RVC_STREAM_ORDER"); if (!ord || strlen(ord) != 4) ord = "mfcs";
    if (const char *pd = tune_env("RVC_STREAM_PAD")) for (int i = 0; i < atoi(pd); i++) { hipStream_t d; HIPCHK(hipStreamCreateWithFlags(&d, hipStreamNonBlocking)); }
    m->masked_ok = want_masks;
    for (int k = 0; k < 4; k++) {
        const char w = ord[k];
        if (w == 'm') HIPCHK(hipStreamCreateWithFlags(&m->main, hipStreamNonBlocking));
        else if (w == 's') HIPCHK(hipStreamCreateWithFlags(&m->plain[1], hipStreamNonBlocking));
        else if (want_masks && m->masked_ok) {
            const int i = w == 'f' ? 0 : 1;
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int c = 0; c < ncu; c++) if ((c < nf0) == (i == 0)) mask[c / 32] |= 1u << (c % 32);
            hipStream_t *dst = i == 0 ? &m->f0 : &m->cv;
            if (hipExtStreamCreateWithCUMask(dst, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
                (void)hipGetLastError();
                m->masked_ok = false;        // (a first stream that did get created stays allocated: it must not be destroyed either)
            }
        }
    }
    g_pool.push_back(m);
    return m;
}
static void release_stream_set(StreamSet *m)
{
    if (!m) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    m->in_use = false;
}

static void configure_aux_streams(rvc_engine *e)
{
    hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, e->device));
    int ncu = prop.multiProcessorCount, nf0 = ncu / 8;
    g_ncu = ncu > 0 ? ncu : 256;
    if (const char *f = tune_env("RVC_F0_CUS")) { const int v = atoi(f); if (v >= 8 && v < ncu) nf0 = v; }   // tuning aid
    if (!e->sset) {
        e->sset = acquire_stream_set(e->device, ncu, nf0, e->partition_ok && ncu >= 64 && ncu <= 1024);
        e->stream = e->sset->main;
        if (!e->sset->masked_ok) e->partition_ok = false;
    }
    const bool want = e->partition_ok && e->sset->masked_ok && e->n_streams <= 4;
    if (!want) for (int i = 0; i < 3; i += 2) if (!e->sset->plain[i]) HIPCHK(hipStreamCreateWithFlags(&e->sset->plain[i], hipStreamNonBlocking));
    e->partitioned = want;
    e->aux[0] = want ? e->sset->f0 : e->sset->plain[0];
    e->aux[1] = e->sset->plain[1];
    e->aux[2] = want ? e->sset->cv : e->sset->plain[2];
}

static void alloc_state(rvc_engine *e)
{
    if (e->d_state) (void)hipFree(e->d_state);
    if (e->d_state_bucket) { (void)hipFree(e->d_state_bucket); e->d_state_bucket = nullptr; }
    if (e->d_bucket_idx) { (void)hipFree(e->d_bucket_idx); e->d_bucket_idx = nullptr; }
    HIPCHK(hipMalloc(&e->d_state, sizeof(StreamState) * e->n_streams));
    reset_state(e);
    e->plans.clear();
    e->last_plan = nullptr;
}

// ------------------------------- ContentVec ------------------------------------------
static T1 build_contentvec(rvc_engine *e, Plan &pl, int B, size_t L)
{
    ModelCV &m = *e->cv;
    Arena &A = pl.arena;
    T1 x; x.p = pl.d_in; x.B = B; x.C = 1; x.T = (int)L; x.ld = (int)L; x.halo = 0; x.bs = (long long)L;
    int T = (int)L;
    for (int i = 0; i < 7; i++) {
        int To = (T - m.conv_k[i]) / m.conv_s[i] + 1;
        T1 y = make_t1(A, B, m.conv_dim, To, 0);
        if (i == 0 && m.conv_k[0] <= 16 && To <= 32 * 256 && m.conv0_raw && !tune_env("RVC_NO_CONV0_FUSE")) {
            // first layer fused: conv (Cin = 1) + per-channel GroupNorm + GELU, outputs held in registers between the passes
            dim3 grid(m.conv_dim, B);
            const float *w0 = m.conv0_raw, *gg = m.gn_g, *bb = m.gn_b; const int kt = m.conv_k[0], st = m.conv_s[0];
            const float *ain = x.p; const long long abs_ = x.bs;
            const int nt = (To + 255) / 256;
            // 16 channels per workgroup share one register copy of the input samples at many streams; one stream: 2 (256 workgroups of
            // 1024 threads, half the strided gathers: 42.8 -> ~15 us, 25-30 us off the ContentVec branch; 4 and 8 measured the same / worse)
            int cpw = B >= 16 ? 16 : (B >= 4 ? 4 : 2);
            if (const char *f = tune_env("RVC_CONV0_CPW")) cpw = std::max(1, atoi(f));      // tuning aid
            while (cpw > 1 && m.conv_dim % cpw) cpw >>= 1;
            if (kt == 10 && To <= 8 * 1024 && cpw > 1 && !test_opt("RVC_NO_CONV0_MULTI")) {
                dim3 gridm(m.conv_dim / cpw, B);
                const int nt1k = (To + 1023) / 1024;
                Plan *plp = &pl;
                pl.ops.push_back([=](hipStream_t s) {
                    const float *in_ = plp->cur_in ? plp->cur_in : ain;      // a device-resident caller's buffer is read in place
                    if (nt1k <= 4) hipLaunchKernelGGL((conv0_gn_gelu_multi_kernel<4, 10>), gridm, dim3(1024), 0, s, in_, abs_, w0, st, gg, bb, y.p, To, y.ld, y.bs, cpw);
                    else hipLaunchKernelGGL((conv0_gn_gelu_multi_kernel<8, 10>), gridm, dim3(1024), 0, s, in_, abs_, w0, st, gg, bb, y.p, To, y.ld, y.bs, cpw);
                });
                add_tap(pl, "cv.conv0", y);
                x = y; T = To;
                continue;
            }
            Plan *plp = &pl;
            pl.ops.push_back([=](hipStream_t s) {
                const float *in_ = plp->cur_in ? plp->cur_in : ain;
                if (nt <= 8) hipLaunchKernelGGL((conv0_gn_gelu_kernel<8>), grid, dim3(256), 0, s, in_, abs_, w0, kt, st, gg, bb, y.p, To, y.ld, y.bs);
                else if (nt <= 16) hipLaunchKernelGGL((conv0_gn_gelu_kernel<16>), grid, dim3(256), 0, s, in_, abs_, w0, kt, st, gg, bb, y.p, To, y.ld, y.bs);
                else hipLaunchKernelGGL((conv0_gn_gelu_kernel<32>), grid, dim3(256), 0, s, in_, abs_, w0, kt, st, gg, bb, y.p, To, y.ld, y.bs);
            });
            add_tap(pl, "cv.conv0", y);
            x = y; T = To;
            continue;
        }
        ConvOpts o; o.act = i == 0 ? ACT_NONE : ACT_GELU;
        if (i == 0) pl.in_direct_ok = false;      // (the generic convolution bakes its input pointer: this plan keeps the staging copy)
        add_conv1d(pl, m.conv[i], x, y, m.conv_s[i], 0, 1, o);
        if (i == 0) {
            dim3 grid(m.conv_dim, B);
            float *g = m.gn_g, *bb = m.gn_b;
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(groupnorm_gelu_kernel, grid, dim3(256), 0, s, y.p, g, bb, y.T, y.ld, y.bs); });
            add_tap(pl, "cv.conv0", y);
        }
        x = y; T = To;
    }
    add_tap(pl, "cv.feat", x);
    const bool fuse_ln = B == 1 && m.has_folded && !pl.plain_plan && !test_opt("RVC_NO_LN_FUSE");
    const int E = m.embed;
    T1 h = make_t1(A, B, E, T, m.pos_k / 2);
    if (fuse_ln && m.proj_wsum) { ConvOpts o; o.ln_wsum = m.proj_wsum; o.ln_rows = m.conv_dim; add_conv1d(pl, m.proj_f, x, h, 1, 0, 1, o); }
    else {
    add_layernorm(pl, x, m.ln0_g, m.ln0_b);
    add_conv1d(pl, m.proj, x, h, 1, 0, 1);
    }
    add_tap(pl, "cv.proj", h);
    T1 h2 = make_t1(A, B, E, T, 0);
    { ConvOpts o; o.act = ACT_GELU; o.res = h.p; o.res_cs = h.ld; o.res_bs = h.bs; add_conv1d(pl, m.pos, h, h2, 1, m.pos_k / 2, 1, o); }
    if (!fuse_ln) add_layernorm(pl, h2, m.encln_g, m.encln_b);      // (folded: layer 0 consumes the not yet normalised sum, see below)
    add_tap(pl, fuse_ln ? "cv.pos.raw" : "cv.pos", h2);
    T1 qkv = make_t1(A, B, 3 * E, T, 0), att = make_t1(A, B, E, T, 0), ff = make_t1(A, B, m.ffn, T, 0);
    const int hd = E / m.heads, Tp = T | 1;
    const size_t attn_lds = ((size_t)((hd * Tp + 3) & ~3) + 16 * Tp + 16 * hd) * sizeof(float);
    if (attn_lds > 160 * 1024) throw ShapeError("ContentVec attention: window too long for the LDS-resident kernel (T <= ~490 at head size 64)");
    HIPCHK(hipFuncSetAttribute((const void *)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // One stream: the 2 LayerNorm launches of a layer are folded into the GEMMs around them (h2 then holds the NOT yet normalised sum;
    // `raw` says so, with the pending LayerNorm's scale / shift and the buffer its column statistics are published in)
    bool raw = fuse_ln; const float *raw_g = m.encln_g, *raw_b = m.encln_b; float *raw_st = nullptr;
    for (int l = 0; l < m.run_layers; l++) {
        ModelCV::Layer &Ly = m.layers[l];
        if (fuse_ln) {
            float *st_a = A.floats((size_t)2 * T + 16);
            if (raw) { ConvOpts o; o.ln_wsum = Ly.qkv_wsum; o.ln_stats_out = st_a; o.ln_rows = E; add_conv1d(pl, Ly.qkv_f, h2, qkv, 1, 0, 1, o); raw_st = st_a; }
            else add_conv1d(pl, Ly.qkv, h2, qkv, 1, 0, 1);
        } else
        add_conv1d(pl, Ly.qkv, h2, qkv, 1, 0, 1);
        AttnP ap{}; ap.qkv = qkv.p; ap.out = att.p; ap.E = E; ap.T = T; ap.heads = m.heads; ap.cs = qkv.ld; ap.bs = qkv.bs; ap.o_cs = att.ld; ap.o_bs = att.bs;
        ap.scale = 1.0f / sqrtf((float)hd); ap.rel_k = nullptr; ap.rel_v = nullptr; ap.window = 0;
        dim3 ag(m.heads * ((T + 15) / 16), B);
        if (B >= 16 && hd == 64 && T <= 256 && !tune_env("RVC_ATTN_VALU") && !tune_env("RVC_NO_QLOOP")) { ap.qloop = 1; ag = dim3(m.heads, B); }
        if (hd == 64 && T <= 128 && !tune_env("RVC_ATTN_VALU")) {
            const size_t mfma_lds = ((size_t)16 * (2 * 64 + 1) + 128) * sizeof(float);
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL((attention_mfma_kernel<64, 2>), ag, dim3(256), mfma_lds, s, ap); });
        } else if (hd == 64 && T <= 256 && !tune_env("RVC_ATTN_VALU")) {
            const size_t mfma_lds = ((size_t)16 * (4 * 64 + 1) + 128) * sizeof(float);
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL((attention_mfma_kernel<64, 4>), ag, dim3(256), mfma_lds, s, ap); });
        } else {
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(attention_kernel, ag, dim3(256), attn_lds, s, ap); });
        }
        if (fuse_ln) {
            float *st_1 = A.floats((size_t)2 * T + 16);
            {   // attention output projection + residual; the residual is LayerNorm2 of the previous layer when that one is still pending
                ConvOpts o; o.res = h2.p; o.res_cs = h2.ld; o.res_bs = h2.bs;
                if (raw) { o.ln_stats_in = raw_st; o.ln_g = raw_g; o.ln_b = raw_b; }
                add_conv1d(pl, Ly.o, att, h2, 1, 0, 1, o);
            }
            { ConvOpts o; o.act = ACT_GELU; o.ln_wsum = Ly.ff1_wsum; o.ln_stats_out = st_1; o.ln_rows = E; add_conv1d(pl, Ly.ff1_f, h2, ff, 1, 0, 1, o); }     // LayerNorm1 folded
            { ConvOpts o; o.res = h2.p; o.res_cs = h2.ld; o.res_bs = h2.bs; o.ln_stats_in = st_1; o.ln_g = Ly.ln1_g; o.ln_b = Ly.ln1_b; add_conv1d(pl, Ly.ff2, ff, h2, 1, 0, 1, o); }
            if (l + 1 < m.run_layers) { raw = true; raw_g = Ly.ln2_g; raw_b = Ly.ln2_b; }
            else { add_layernorm(pl, h2, Ly.ln2_g, Ly.ln2_b); raw = false; }
        } else {
        { ConvOpts o; o.res = h2.p; o.res_cs = h2.ld; o.res_bs = h2.bs; add_conv1d(pl, Ly.o, att, h2, 1, 0, 1, o); }
        add_layernorm(pl, h2, Ly.ln1_g, Ly.ln1_b);
        { ConvOpts o; o.act = ACT_GELU; add_conv1d(pl, Ly.ff1, h2, ff, 1, 0, 1, o); }
        { ConvOpts o; o.res = h2.p; o.res_cs = h2.ld; o.res_bs = h2.bs; add_conv1d(pl, Ly.ff2, ff, h2, 1, 0, 1, o); }
        add_layernorm(pl, h2, Ly.ln2_g, Ly.ln2_b);
        }
        if (pl.with_taps) { char nm[32]; snprintf(nm, sizeof nm, raw ? "cv.l%d.raw" : "cv.l%d", l); add_tap(pl, nm, h2); } else if (l % 4 == 3) add_stamp(pl, "cv.l4");
    }
    T1 out = h2;
    if (m.out_dim != E) { out = make_t1(A, B, m.out_dim, T, 0); add_conv1d(pl, m.final_proj, h2, out, 1, 0, 1); }
    add_tap(pl, "cv.out", out);
    pl.T = T; pl.C = m.out_dim;
    return out;
}

// ------------------------------- RMVPE ------------------------------------------------
// c1 (3x3, ReLU) -> y1 and the shortcut (1x1, no activation) -> out as the two phases of ONE launch over the shared input x
// Two 1x1 convolutions of ONE input with the same M and K into two output tensors, as two phases of one launch (the flows' merged post / next-pre
// layer).  pair_bias = [c0's bias | c1's bias].
static void add_conv1d_two(Plan &pl, const ConvW &c0, const ConvW &c1, const float *pair_bias, const T1 &x, const T1 &y0, const T1 &y1)
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

static void add_conv2d_with_shortcut(Plan &pl, const ResBlockW &w, const T2 &x, const T2 &y1, const T2 &out)
{
    if (x.H != y1.H || x.W != y1.W || out.H != x.H || out.W != x.W || y1.cs != out.cs || y1.ld != out.ld || (x.B > 1 && y1.bs != out.bs)) throw ShapeError("conv2d + shortcut: layouts differ");
    const ConvW &c1 = w.c1, &sc = w.sc;
    IgemmP p{};
    p.x = x.p; p.y = y1.p;
    p.M = c1.M; p.N = x.H * x.W;
    p.NW = x.W; p.x_hs = x.ld; p.x_ws = 1; p.y_hm = 1; p.y_ws = 1; p.OW = y1.W;
    p.x_bs = x.bs; p.y_bs = y1.bs; p.y_cs = y1.cs; p.y_rs = y1.ld;
    ConvOpts o; o.act = ACT_RELU;
    fill_epilogue(p, c1, o);
    p.bias = w.pair_bias;
    std::vector<int> koff;
    std::vector<PhaseD> ph(2);
    ph[0] = PhaseD{}; ph[1] = PhaseD{};
    // phase 0: the 3x3 convolution (one-row images: only the middle tap row can hit data, see add_conv2d)
    if (x.H == 1 && !c1.host_w.empty() && !tune_env("RVC_NO_TAP_PRUNE")) {
        const int K3 = c1.Cin * 3, Kp3 = round16(K3);
        std::vector<float> panel((size_t)c1.M * Kp3, 0.f);
        for (int mo = 0; mo < c1.M; mo++)
            for (int ci = 0; ci < c1.Cin; ci++)
                for (int kw = 0; kw < 3; kw++) panel[(size_t)mo * Kp3 + ci * 3 + kw] = c1.host_w[(size_t)mo * c1.K + ci * 9 + 3 + kw];
        float *dw = upload_fragments(panel, 1, c1.M, Kp3);
        pl.owned_dev.push_back(dw);
        p.w = dw; ph[0].nchunks = Kp3 / 16;
        koff.assign(Kp3, 0);
        for (int ci = 0; ci < c1.Cin; ci++) for (int kw = 0; kw < 3; kw++) koff[ci * 3 + kw] = ci * x.cs + (kw - 1);
    } else {
        p.w = c1.w; ph[0].nchunks = c1.Kp / 16;
        koff.assign(c1.Kp, 0);
        for (int ci = 0; ci < c1.Cin; ci++) for (int k = 0; k < 9; k++) koff[ci * 9 + k] = ci * x.cs + (k / 3 - 1) * x.ld + (k % 3 - 1);
    }
    // phase 1: the shortcut: its own weights (offset from phase 0's: both are device pointers of one flat address space), K, bias
    // slice, output tensor and (no) activation
    ph[1].w_off = sc.w - p.w;
    ph[1].nchunks = sc.Kp / 16;
    ph[1].koff_off = (int)koff.size();
    ph[1].bias_off = c1.M;
    ph[1].act_p1 = ACT_NONE + 1;
    ph[1].y_off = out.p - y1.p;
    const size_t base = koff.size();
    koff.resize(base + sc.Kp, 0);
    for (int ci = 0; ci < sc.Cin; ci++) koff[base + ci] = ci * x.cs;
    p.K = std::max(ph[0].nchunks, ph[1].nchunks) * 16;
    queue_igemm(pl, p, x.B, koff, ph);
}

static T2 res_block(Plan &pl, const ResBlockW &w, const T2 &x, const T2 &out)
{
    Arena &A = pl.arena;
    T2 y1 = make_t2(A, x.B, w.co, x.H, x.W);
    // few streams: the 3x3 convolution and the 1x1 shortcut read the same input -- one launch with two phases (own K, own output tensor,
    // own activation) instead of two dependent launches (11 blocks of RMVPE have a shortcut: 11 launches off the f0 branch)
    if (w.has_sc && w.pair_bias && x.B <= 4 && (x.B == 1 || y1.bs == out.bs) && !tune_env("RVC_NO_SC_MERGE")) {      // (one stream stride for both outputs)
        add_conv2d_with_shortcut(pl, w, x, y1, out);
        ConvOpts o; o.act = ACT_RELU; o.accumulate = true; add_conv2d(pl, w.c2, y1, out, o);
        return out;
    }
    { ConvOpts o; o.act = ACT_RELU; add_conv2d(pl, w.c1, x, y1, o); }
    if (w.has_sc) {
        add_conv2d(pl, w.sc, x, out);
        ConvOpts o; o.act = ACT_RELU; o.accumulate = true; add_conv2d(pl, w.c2, y1, out, o);
    } else {
        ConvOpts o; o.act = ACT_RELU; o.res = x.p; o.res_cs = x.cs; o.res_bs = x.bs; o.res_rs = x.ld; add_conv2d(pl, w.c2, y1, out, o);
    }
    return out;
}

static T1 build_rmvpe(rvc_engine *e, Plan &pl, int B, size_t L, size_t frame16k, bool update_cache)
{
    ModelRM &m = *e->rm;
    Arena &A = pl.arena;
    const size_t fr = 5120 * ((frame16k + 800 - 1) / 5120 + 1) - 160;     // rmvpe.rs:256
    if (fr > L) throw PanicError("input shorter than f0_extractor_frame");
    const int Tm = (int)(1 + fr / 160);
    if (Tm % 32 != 0) throw PanicError("mel frame count is not a multiple of 32 (rmvpe.rs:229-233 branch)");
    if (Tm > 1024) throw ShapeError("f0 window too long");
    pl.Tm = Tm;
    const int H0 = Tm, W0 = m.n_mels;
    if ((H0 >> m.levels) < 1 || (W0 >> m.levels) < 1) throw ShapeError("RMVPE: input too small for the U-Net depth");
    T2 img = make_t2(A, B, 1, H0, W0);
    float *d_mel = A.floats((size_t)B * 128 * Tm);
    {
        MelP mp{};
        mp.audio = pl.d_in; mp.audio_bs = (long long)L; mp.n = (int)L; mp.frame = (int)fr; mp.Tm = Tm;
        mp.window = e->d_window; mp.twiddle = e->d_twiddle; mp.basis = e->d_basis; mp.band = e->d_band;
        mp.mel = d_mel; mp.img = img.p; mp.img_bs = img.bs; mp.img_ld = img.ld; mp.bn_scale = m.bn_scale; mp.bn_shift = m.bn_shift;
        dim3 grid(Tm, B);
        Plan *plp = &pl;
        pl.ops.push_back([=](hipStream_t s) { MelP m2 = mp; if (plp->cur_in) m2.audio = plp->cur_in; hipLaunchKernelGGL(mel_frontend_kernel, grid, dim3(256), 0, s, m2); });
        add_stamp(pl, "rm.mel0");
        if (pl.with_taps) { T1 t; t.p = d_mel; t.B = B; t.C = 128; t.T = Tm; t.ld = Tm; t.halo = 0; t.bs = 128LL * Tm; add_tap(pl, "rm.mel", t); }
    }
    // encoder; every level's pre-pool output is written straight into the second half of the decoder's concat buffer
    std::vector<T2> cat(m.levels);
    {
        int H = H0, W = W0, co = m.en_out;
        for (int lv = 0; lv < m.levels; lv++) { cat[lv] = make_t2(A, B, 2 * co, H, W); H /= 2; W /= 2; co *= 2; }
    }
    T2 x = img;
    int H = H0, W = W0;
    for (int lv = 0; lv < m.levels; lv++) {
        const int co = m.enc[lv][0].co;
        for (int j = 0; j < m.n_blocks; j++) {
            T2 out = (j == m.n_blocks - 1) ? cat[lv].chans(co, co) : make_t2(A, B, co, H, W);
            x = res_block(pl, m.enc[lv][j], x, out);
        }
        if (pl.with_taps) { char nm[32]; snprintf(nm, sizeof nm, "rm.enc%d", lv); add_tap2(pl, nm, x); }
        T2 p = make_t2(A, B, co, H / 2, W / 2);
        {
            T2 xi = x;
            dim3 grid((co * (H / 2) * (W / 2) + 255) / 256, B);
            pl.ops.push_back([=](hipStream_t s) {
                hipLaunchKernelGGL(avgpool2_kernel, grid, dim3(256), 0, s, xi.p, xi.ld, xi.cs, xi.bs, p.p, p.ld, p.cs, p.bs, co, p.H, p.W);
            });
        }
        x = p; H /= 2; W /= 2;
    }
    for (int lv = 0; lv < m.inter_layers; lv++)
        for (int j = 0; j < m.n_blocks; j++) { T2 out = make_t2(A, B, m.inter[lv][j].co, H, W); x = res_block(pl, m.inter[lv][j], x, out); }
    add_tap2(pl, "rm.int", x);
    for (int lv = 0; lv < m.levels; lv++) {
        const int sl = m.levels - 1 - lv, co = m.up[lv].Cout;
        H *= 2; W *= 2;
        { ConvOpts o; o.act = ACT_RELU; add_convT2d(pl, m.up[lv], x, cat[sl].chans(0, co), o); }
        x = cat[sl];
        for (int j = 0; j < m.n_blocks; j++) { T2 out = make_t2(A, B, co, H, W); x = res_block(pl, m.dec[lv][j], x, out); }
        if (pl.with_taps) { char nm[32]; snprintf(nm, sizeof nm, "rm.dec%d", lv); add_tap2(pl, nm, x); }
    }
    T2 cn = make_t2(A, B, 3, H, W);
    add_conv2d(pl, m.cnn, x, cn);
    const int Hg = m.gru_hidden, I = 3 * m.n_mels;
    T1 feat = make_t1(A, B, I, Tm, 0), gi = make_t1(A, B, 6 * Hg, Tm, 0), gout = make_t1(A, B, 2 * Hg, Tm, 0), sal = make_t1(A, B, m.n_out, Tm, 0);
    {
        dim3 grid((3 * m.n_mels * Tm + 255) / 256, B); int nm = m.n_mels;
        pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(gru_input_kernel, grid, dim3(256), 0, s, cn.p, cn.ld, cn.cs, cn.bs, feat.p, feat.ld, feat.bs, Tm, nm); });
    }
    add_conv1d(pl, m.gru_ih, feat, gi, 1, 0, 1);
    {
        if (3 * Hg > 1024) throw ShapeError("GRU hidden size too large for the single-workgroup recurrence");
        int threads = (3 * Hg + 63) / 64 * 64;
        size_t lds = (size_t)4 * Hg * sizeof(float);
        float *wt = m.whhT, *bh = m.bhh;
        dim3 grid(2, B);
        if (Hg == 256 && B <= 8 && Tm <= 256 && !tune_env("RVC_GRU_GENERIC")) {
            // few streams: spread each direction over 8 CUs with W_hh resident in LDS (granule hand-off per step)
            GruMultiP gp{}; gp.gi = gi.p; gp.gi_cs = gi.ld; gp.gi_bs = gi.bs; gp.whh = m.whh; gp.bhh = m.bhh; gp.out = gout.p; gp.o_cs = gout.ld; gp.o_bs = gout.bs;
            gp.Tm = Tm; gp.status = &e->d_state[0].status; gp.status_stride = (int)(sizeof(StreamState) / sizeof(int));
            const size_t gbytes = (size_t)B * 2 * 2 * 256 * sizeof(unsigned long long);
            gp.gran = (unsigned long long *)pl.arena.alloc(gbytes);
            const size_t lds3 = (size_t)(256 + 96 + (size_t)Tm * 96) * sizeof(float);      // h, gate pre-activations, this slice's input gates for all steps
            const dim3 g3(8, 2, B);
            pl.ops.push_back([=](hipStream_t s) {
                HIPCHK(hipMemsetAsync(gp.gran, 0, gbytes, s));
                hipLaunchKernelGGL(gru_multi_kernel, g3, dim3(384), lds3, s, gp);
            });
        } else {
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(gru_kernel, grid, dim3(threads), lds, s, gi.p, gi.ld, gi.bs, wt, bh, gout.p, gout.ld, gout.bs, Hg, Tm); });
        }
    }
    { ConvOpts o; o.act = ACT_SIGMOID; add_conv1d(pl, m.fc, gout, sal, 1, 0, 1, o); }
    if (pl.with_taps) { add_tap(pl, "rm.sal_ct", sal); add_tap(pl, "rm.gru_ct", gout); add_tap(pl, "rm.cnn_ct", feat); } else add_stamp(pl, "rm.sal");
    return sal;
}

// decode + pitch shift + pitch cache + get_f0_post (rmvpe.rs:118-133,243-248; rvc.rs:121,167-180; f0/mod.rs:7-12)
static void build_pitch_post(rvc_engine *e, Plan &pl, int B, const T1 &sal, bool update_cache, size_t frame16k, size_t hubert_length,
                             float **pitchf_out, int **pitch_out)
{
    Arena &A = pl.arena;
    const int Tm = pl.Tm;
    pl.d_f0 = A.floats((size_t)B * Tm);
    PitchP pp{};
    pp.sal = sal.p; pp.sal_cs = sal.ld; pp.sal_bs = sal.bs; pp.Tm = Tm;
    pp.st = e->d_state; pp.cp = e->d_cp; pp.f0 = pl.d_f0; pp.threshold = 0.03f;   // rvc.rs:122
    if (update_cache) {
        const int R = (int)pl.R;
        const size_t shift = frame16k / 160;                                   // rvc.rs:168
        if (shift > 1024 || Tm < 5) throw PanicError("pitch cache shift out of range");
        const long long cache_start = 1024 + 4 - Tm;                            // rvc.rs:172
        const long long read_start = 1024 - (long long)hubert_length + pl.skip_head;   // rvc.rs:176
        if (cache_start < 0 || read_start < 0 || read_start + R > 1024) throw PanicError("pitch cache slice out of range");
        pp.pitchf = A.floats((size_t)B * R);
        pp.pitch = (int *)A.alloc((size_t)B * R * sizeof(int));
        pp.shift = (int)shift; pp.cache_start = (int)cache_start; pp.read_start = (int)read_start; pp.R = R;
        *pitchf_out = pp.pitchf; *pitch_out = pp.pitch;
    } else {
        pp.R = 0; pp.shift = 0; pp.cache_start = 1 << 30; pp.read_start = 0; pp.pitchf = nullptr; pp.pitch = nullptr;
    }
    pp.update = update_cache ? 1 : 0;
    pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(pitch_post_kernel, dim3(B), dim3(1024), 0, s, pp); });
}

// ------------------------------- synthesizer ------------------------------------------
// NSF harmonic source: depends only on the f0 branch, so it is queued on that branch's stream
static T1 build_nsf_source(rvc_engine *e, Plan &pl, int B, float *d_pitchf)
{
    ModelSY &m = *e->sy;
    Arena &A = pl.arena;
    const int R = (int)pl.R;
    const int upp = m.upp();
    const size_t N = (size_t)R * upp;
    if (R > 512) throw ShapeError("return_length too long for the NSF source kernel");
    int max_sf = 1; { int sf = 1; for (int i = m.n_ups - 1; i >= 1; i--) { sf *= m.up_rate[i]; max_sf = std::max(max_sf, sf); } }
    T1 src = make_t1(A, B, 1, (int)N, max_sf + 2);
    {
        SrcP sp{}; sp.pitchf = d_pitchf; sp.src = src.p; sp.src_bs = src.bs; sp.T = R; sp.upp = upp; sp.sr = (float)m.sr;
        sp.lin_w = m.src_w; sp.lin_b = m.src_b; sp.st = e->d_state; sp.cp = e->d_cp;
        pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(nsf_source_kernel, dim3(B), dim3(1024), 0, s, sp); });
    }
    add_tap(pl, "sy.src", src);
    return src;
}

// The decoder adds a strided convolution of the harmonic source to the output of every upsampling stage.  Those convolutions depend on
// the source only: they are queued right behind it on the side stream (next to the text encoder and the flow) and the upsampling
// convolution takes their result as its residual -- 4 launches off the serial chain; the sum has the same operands as before.
static std::vector<T1> build_noise_convs(rvc_engine *e, Plan &pl, int B, const T1 &src)
{
    ModelSY &m = *e->sy;
    std::vector<T1> nz;
    int c = m.up_init, Tc = (int)pl.R;
    for (int i = 0; i < m.n_ups; i++) {
        const int co = c / 2, Tn = Tc * m.up_rate[i];
        T1 t = make_t1(pl.arena, B, co, Tn, 0);
        int sf = 1; for (int q = i + 1; q < m.n_ups; q++) sf *= m.up_rate[q];
        if (i + 1 < m.n_ups) add_conv1d(pl, m.ncs[i], src, t, sf, sf / 2, 1); else add_conv1d(pl, m.ncs[i], src, t, 1, 0, 1);
        nz.push_back(t);
        c = co; Tc = Tn;
    }
    return nz;
}

static void build_synth(rvc_engine *e, Plan &pl, int B, const T1 &phone, const T1 &src, float *d_pitchf, int *d_pitch, int src_join_sid,
                        const std::vector<T1> *nz = nullptr)
{
    ModelSY &m = *e->sy;
    Arena &A = pl.arena;
    const int R = (int)pl.R, H = m.hidden, I = m.inter, F = m.filter, half = I / 2;
    const int HALO = 4;
    if (m.enc_k / 2 > HALO || m.wn_k / 2 > HALO) throw ShapeError("synth kernel sizes exceed the halo");
    T1 z = make_t1(A, B, I, R, HALO), zf = make_t1(A, B, I, R, HALO);
    {
        T1 x = make_t1(A, B, H, R, HALO);
        add_conv1d(pl, m.phone, phone, x, 1, 0, 1);
        {
            dim3 grid((H * R + 255) / 256, B); float *emb = m.pitch_emb; float sq = sqrtf((float)H);
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(embed_pitch_kernel, grid, dim3(256), 0, s, x.p, x.ld, x.bs, emb, d_pitch, H, R, sq); });
        }
        add_tap(pl, "sy.emb", x);
        T1 qkv = make_t1(A, B, 3 * H, R, 0), att = make_t1(A, B, H, R, 0), ff = make_t1(A, B, F, R, HALO);
        const int kc = H / m.heads, Tp = R | 1;
        const size_t attn_lds = ((size_t)((kc * Tp + 3) & ~3) + 16 * Tp + 16 * kc) * sizeof(float);
        if (attn_lds > 160 * 1024) throw ShapeError("synth attention: return_length too long for the LDS-resident kernel");
        // one stream: the second LayerNorm of every encoder layer is folded into the next projection (see build_contentvec)
        const bool fuse_ln = B == 1 && m.has_folded && !pl.plain_plan && !test_opt("RVC_NO_LN_FUSE");
        bool raw = false; const float *raw_g = nullptr, *raw_b = nullptr; float *raw_st = nullptr;
        for (int l = 0; l < m.enc_layers; l++) {
            ModelSY::Layer &Ly = m.layers[l];
            if (raw) { raw_st = A.floats((size_t)2 * R + 16); ConvOpts o; o.ln_wsum = Ly.qkv_wsum; o.ln_stats_out = raw_st; o.ln_rows = H; add_conv1d(pl, Ly.qkv_f, x, qkv, 1, 0, 1, o); }
            else add_conv1d(pl, Ly.qkv, x, qkv, 1, 0, 1);
            AttnP ap{}; ap.qkv = qkv.p; ap.out = att.p; ap.E = H; ap.T = R; ap.heads = m.heads; ap.cs = qkv.ld; ap.bs = qkv.bs; ap.o_cs = att.ld; ap.o_bs = att.bs;
            ap.scale = 1.0f / sqrtf((float)kc); ap.rel_k = Ly.rel_k; ap.rel_v = Ly.rel_v; ap.window = m.window;
            const size_t small_lds = ((size_t)2 * kc * Tp + 2 * (2 * m.window + 1) * kc + 4 * kc + 4 * 64) * sizeof(float);
            // one stream: the matrix-core form (VALU form: 12.4 us per layer of dependent LDS reads)
            const int a_tp = R | 1, a_nr = 2 * m.window + 1, a_jf = (R + 15) / 16, a_pw = (a_nr + 15) / 16 * 16, a_nrp = (a_nr + 3) / 4 * 4;
            const size_t mfma_lds = ((size_t)kc * 16 + 2 * (size_t)kc * a_tp + (size_t)a_pw * kc + (size_t)a_nrp * kc + 16 * a_jf * 16 + 2 * 16 * a_pw + 64) * sizeof(float);
            if (B <= 4 && R <= 64 && kc % 16 == 0 && mfma_lds <= 160 * 1024 && !tune_env("RVC_NO_SMALL_ATTN") && !tune_env("RVC_ATTN_VALU") && !tune_env("RVC_NO_SMALL_ATTN_MFMA")) {
                dim3 ag(m.heads * a_jf, B);
                pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(relpos_attention_mfma_kernel, ag, dim3(256), mfma_lds, s, ap); });
            } else if (R <= 64 && small_lds <= 160 * 1024 && !tune_env("RVC_NO_SMALL_ATTN")) {
                dim3 ag(m.heads * ((R + 3) / 4), B);
                pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(relpos_attention_small_kernel, ag, dim3(256), small_lds, s, ap); });
            } else {
                dim3 ag(m.heads * ((R + 15) / 16), B);
                pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(attention_kernel, ag, dim3(256), attn_lds, s, ap); });
            }
            {
                ConvOpts o; o.res = x.p; o.res_cs = x.ld; o.res_bs = x.bs;
                if (raw) { o.ln_stats_in = raw_st; o.ln_g = raw_g; o.ln_b = raw_b; }
                add_conv1d(pl, Ly.o, att, x, 1, 0, 1, o);
            }
            add_layernorm(pl, x, Ly.ln1_g, Ly.ln1_b);
            { ConvOpts o; o.act = ACT_RELU; add_conv1d(pl, Ly.ff1, x, ff, 1, m.enc_k / 2, 1, o); }
            { ConvOpts o; o.res = x.p; o.res_cs = x.ld; o.res_bs = x.bs; add_conv1d(pl, Ly.ff2, ff, x, 1, m.enc_k / 2, 1, o); }
            if (fuse_ln) { raw = true; raw_g = Ly.ln2_g; raw_b = Ly.ln2_b; }
            else add_layernorm(pl, x, Ly.ln2_g, Ly.ln2_b);
        }
        add_tap(pl, raw ? "sy.enc.raw" : "sy.enc", x);
        // one stream: WaveNets with their res_skip layers composed away and post + next pre merged (ModelSY::compose_flows): 21 launches for
        // four flows instead of 40.  U[k] = [ones16 | h0 (H) | a_0 .. a_{n-1} | z (I)]; flow k reads U[k & 1] and writes h0 and z of U[(k + 1) & 1]
        static const int wn_max_b = tune_env("RVC_WN_COMPOSE_MAX") ? atoi(tune_env("RVC_WN_COMPOSE_MAX")) : 8;
        const bool wn_composed = B <= wn_max_b && H % 16 == 0 && I == H && !pl.plain_plan && !test_opt("RVC_NO_WN_COMPOSE");      // (launch-bound up to a few streams)
        T1 U[2];
        const int u_z = 16 + H + H * m.wn_layers;                     // first latent row of U
        if (wn_composed) {
            m.compose_flows();
            std::vector<float> ones(R, 1.0f);
            for (int k = 0; k < 2; k++) {
                U[k] = make_t1(A, B, u_z + I, R, HALO);
                for (int bb = 0; bb < B; bb++) HIPCHK(hipMemcpy(U[k].p + (long long)bb * U[k].bs, ones.data(), (size_t)R * sizeof(float), hipMemcpyHostToDevice));
            }
            z = U[0].rows(u_z, I);                                     // the prior sample lands in U[0]'s latent rows
        }
        T1 stats = make_t1(A, B, 2 * I, R, 0);
        if (raw) { ConvOpts o; o.ln_wsum = m.proj_wsum; o.ln_rows = H; add_conv1d(pl, m.proj_f, x, stats, 1, 0, 1, o); }
        else add_conv1d(pl, m.proj, x, stats, 1, 0, 1);
        add_tap(pl, "sy.stats", stats);
        {
            dim3 grid(((I * R + 3) / 4 + 255) / 256, B); StreamState *st = e->d_state; CallParams *cp = e->d_cp;
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(prior_sample_kernel, grid, dim3(256), 0, s, stats.p, stats.ld, stats.bs, z.p, z.ld, z.bs, I, R, st, cp); });
        }
        add_tap(pl, "sy.zp", z);
        // hh (WaveNet state, rows 0..H) and skip (rows H..2H) share one tensor: the res_skip conv updates both in one launch
        T1 hs = make_t1(A, B, 2 * H, R, HALO), acts = make_t1(A, B, H, R, 0);
        T1 hh = hs.rows(0, H), skip = hs.rows(H, H);
        for (int fi = m.flow_n - 1; fi >= 0; fi--) {
            ModelSY::Flow &Fw = m.flows[fi];
            const T1 x0 = Fw.flipped ? z.rows(half, half) : z.rows(0, half), x1 = Fw.flipped ? z.rows(0, half) : z.rows(half, half);
            if (wn_composed) {
                const int k = m.flow_n - 1 - fi;
                const T1 &Uc = U[k & 1], &Un = U[(k + 1) & 1];
                if (k == 0) add_conv1d(pl, Fw.pre1, Uc.rows(u_z, I), Uc.rows(16, H), 1, 0, 1);
                for (int j = 0; j < m.wn_layers; j++) { ConvOpts o; o.glu = true; add_conv1d(pl, Fw.inc[j], Uc.rows(0, 16 + H * (j + 1)), Uc.rows(16 + H * (j + 1), H), 1, (m.wn_k - 1) / 2, 1, o); }
                if (fi > 0) add_conv1d_two(pl, Fw.posth, Fw.postc, Fw.pair_bias, Uc.rows(16 + H, H * m.wn_layers + I), Un.rows(16, H), Un.rows(u_z, I));
                else add_conv1d(pl, Fw.postc, Uc.rows(16 + H, H * m.wn_layers + I), Un.rows(u_z, I), 1, 0, 1);
                if (fi == 0) z = Un.rows(u_z, I);
                if (pl.with_taps) { char nm[32]; snprintf(nm, sizeof nm, "sy.flow%d", fi); add_tap(pl, nm, Un.rows(u_z, I)); } else add_stamp(pl, "sy.flow");
                continue;
            }
            add_conv1d(pl, Fw.pre, x0, hs, 1, 0, 1);                       // hh = pre(x0), skip = 0
            for (int j = 0; j < m.wn_layers; j++) {
                { ConvOpts o; o.glu = true; add_conv1d(pl, Fw.in[j], hh, acts, 1, (m.wn_k - 1) / 2, 1, o); }   // acts = tanh(.) * sigmoid(.)
                ConvOpts o; o.accumulate = true;
                if (j < m.wn_layers - 1) add_conv1d(pl, Fw.rs[j], acts, hs, 1, 0, 1, o);     // hh += res, skip += skip part
                else add_conv1d(pl, Fw.rs[j], acts, skip, 1, 0, 1, o);
            }
            { ConvOpts o; o.scale = -1.f; o.accumulate = true; add_conv1d(pl, Fw.post, skip, x1, 1, 0, 1, o); }
            if (pl.with_taps) { char nm[32]; snprintf(nm, sizeof nm, "sy.flow%d", fi); add_tap(pl, nm, z); } else add_stamp(pl, "sy.flow");
        }
        if (m.flow_n & 1) {
            // odd number of flips: materialise the last one
            T1 zi = z, zo = zf; dim3 grid((I * R + 255) / 256, B);
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(flip_channels_kernel, grid, dim3(256), 0, s, zi.p, zo.p, I, R, zi.ld, zi.bs); });
            std::swap(z, zf);
        }
    }
    add_tap(pl, "sy.z", z);
    const int upp = m.upp();
    const size_t N = (size_t)R * upp;
    (void)N;
    // decoder
    int max_pad = 3;
    for (int j = 0; j < m.n_rb; j++) for (int q = 0; q < m.n_rbd; q++) max_pad = std::max(max_pad, (m.rb_k[j] * m.rb_d[q] - m.rb_d[q]) / 2);
    const int DH = (max_pad + 3) / 4 * 4;
    int c = m.up_init, Tc = R;
    if (src_join_sid > 0) pl.ops.join(src_join_sid);     // the harmonic source was produced on a side stream
    T1 xd = make_t1(A, B, c, Tc, DH);
    add_conv1d(pl, m.dec_pre, z, xd, 1, 3, 1);
    add_tap(pl, "sy.pre", xd);
    for (int i = 0; i < m.n_ups; i++) {
        const int co = c / 2, K = m.up_kernel[i], S = m.up_rate[i], Tn = Tc * S;
        if ((K - S) % 2 != 0) throw ShapeError("upsample kernel/stride parity not supported");
        T1 u = make_t1(A, B, co, Tn, DH);
        if (nz) {
            const T1 &r = (*nz)[i];
            ConvOpts o; o.pre_act = ACT_LRELU; o.pre_slope = 0.1f; o.res = r.p; o.res_cs = r.ld; o.res_bs = r.bs;
            add_convT1d(pl, m.ups[i], xd, u, (K - S) / 2, o);
        } else {
        { ConvOpts o; o.pre_act = ACT_LRELU; o.pre_slope = 0.1f; add_convT1d(pl, m.ups[i], xd, u, (K - S) / 2, o); }
        int sf = 1; for (int q = i + 1; q < m.n_ups; q++) sf *= m.up_rate[q];
        { ConvOpts o; o.accumulate = true; if (i + 1 < m.n_ups) add_conv1d(pl, m.ncs[i], src, u, sf, sf / 2, 1, o); else add_conv1d(pl, m.ncs[i], src, u, 1, 0, 1, o); }
        }
        if (pl.with_taps) { char nm[32]; snprintf(nm, sizeof nm, "sy.up%d", i); add_tap(pl, nm, u); } else add_stamp(pl, "sy.up");
        // the n_rb ResBlock chains of a stage are independent until their average
        T1 xs = make_t1(A, B, co, Tn, DH);
        std::vector<T1> finals;
        const bool fused = m.n_rb > 1 && m.n_rb <= 3 && B < 16 && !tune_env("RVC_SERIAL_RESBLOCKS");   // many streams: every conv fills the chip by itself
        if (fused) {
            // one launch per (dilation, conv): phase j = chain j (kernel size rb_k[j]); 6 launches per stage instead of 6*n_rb
            const int nr = m.n_rb;
            T1 ra = make_t1(A, B, nr * co, Tn, DH), rb = make_t1(A, B, nr * co, Tn, DH), tt = make_t1(A, B, nr * co, Tn, DH), fin = make_t1(A, B, nr * co, Tn, 0);
            T1 cur = u; bool grouped = false;
            for (int q = 0; q < m.n_rbd; q++) {
                const int d = m.rb_d[q];
                std::vector<const ConvW *> c1, c2; std::vector<int> p1, d1, p2, d2;
                for (int j = 0; j < nr; j++) {
                    c1.push_back(&m.rbs[i][j][q].first); c2.push_back(&m.rbs[i][j][q].second);
                    p1.push_back((m.rb_k[j] * d - d) / 2); d1.push_back(d); p2.push_back((m.rb_k[j] - 1) / 2); d2.push_back(1);
                }
                { ConvOpts o; o.pre_act = ACT_LRELU; o.pre_slope = 0.1f; o.act = ACT_LRELU; o.slope = 0.1f; add_conv1d_multi(pl, c1, cur, grouped, tt, p1, d1, o); }
                const bool last = q == m.n_rbd - 1;
                T1 dst = last ? fin : (cur.p == ra.p ? rb : ra);
                ConvOpts o; o.res = cur.p; o.res_cs = cur.ld; o.res_bs = cur.bs;
                add_conv1d_multi(pl, c2, tt, true, dst, p2, d2, o, grouped);
                cur = dst; grouped = true;
            }
            for (int j = 0; j < nr; j++) finals.push_back(fin.rows(j * co, co));
        }
        for (int j = 0; j < m.n_rb && !fused; j++) {
            const int k = m.rb_k[j];
            T1 ra = make_t1(A, B, co, Tn, DH), rb = make_t1(A, B, co, Tn, DH), tt = make_t1(A, B, co, Tn, DH), fin = make_t1(A, B, co, Tn, 0);
            T1 cur = u;
            for (int q = 0; q < m.n_rbd; q++) {
                const int d = m.rb_d[q];
                { ConvOpts o; o.pre_act = ACT_LRELU; o.pre_slope = 0.1f; o.act = ACT_LRELU; o.slope = 0.1f; add_conv1d(pl, m.rbs[i][j][q].first, cur, tt, 1, (k * d - d) / 2, d, o); }
                const bool last = q == m.n_rbd - 1;
                T1 dst = last ? fin : (cur.p == ra.p ? rb : ra);
                ConvOpts o; o.res = cur.p; o.res_cs = cur.ld; o.res_bs = cur.bs;
                add_conv1d(pl, m.rbs[i][j][q].second, tt, dst, 1, (k - 1) / 2, 1, o);
                cur = dst;
            }
            finals.push_back(fin);
        }
        {
            // xs = (r0 + r1 + ...) / n_rb, summed in chain order as in the reference definition
            const int nrb = m.n_rb; const float inv = 1.0f / (float)m.n_rb;
            const float *f0 = finals[0].p, *f1 = nrb > 1 ? finals[1].p : nullptr, *f2 = nrb > 2 ? finals[2].p : nullptr;
            if (nrb > 3) throw ShapeError("more than 3 ResBlock kernels per stage");
            T1 fi = finals[0];
            dim3 grid((co * Tn + 255) / 256, B);
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(mean3_kernel, grid, dim3(256), 0, s, f0, f1, f2, fi.ld, fi.bs, xs.p, xs.ld, xs.bs, co, Tn, inv); });
        }
        if (pl.with_taps) { char nm[32]; snprintf(nm, sizeof nm, "sy.rb%d", i); add_tap(pl, nm, xs); } else add_stamp(pl, "sy.rb");
        xd = xs; c = co; Tc = Tn;
    }
    pl.audio = make_t1(A, B, 1, Tc, 0);
    { ConvOpts o; o.pre_act = ACT_LRELU; o.pre_slope = 0.01f; o.act = ACT_TANH; o.no_bias = true; o.final_out = true; add_conv1d(pl, m.dec_post, xd, pl.audio, 1, 3, 1, o); }
    pl.N = (size_t)Tc;
    pl.out_direct_ok = pl.audio.ld == Tc && !pl.with_taps;     // (the split-K fallback writes through a second kernel: ksplit > 1 never happens for this 7-tap layer)
    if (pl.audio.ld != Tc) {
        // make the output rows contiguous [B][N] for the device-pointer API
        T1 a2; a2.p = A.floats((size_t)B * Tc); a2.B = B; a2.C = 1; a2.T = Tc; a2.ld = Tc; a2.halo = 0; a2.bs = Tc;
        T1 a1 = pl.audio;
        pl.ops.push_back([=](hipStream_t s) { HIPCHK(hipMemcpy2DAsync(a2.p, (size_t)Tc * 4, a1.p, (size_t)a1.bs * 4, (size_t)Tc * 4, B, hipMemcpyDeviceToDevice, s)); });
        pl.audio = a2;
    }
    add_stamp(pl, "sy.audio");
}

// ------------------------------- plan -------------------------------------------------
static void ensure_index_transposed(rvc_engine *e);
static Plan *get_plan(rvc_engine *e, int mode, size_t L, size_t frame16k, uint32_t skip_head, uint32_t R, int slot = 0, int bucket_B = 0)
{
    // bucket_B > 0 (rvc_infer_batch_g): a plan for bucket_B of the engine's streams, whose states the caller gathers into d_state_bucket.
    // The builders read the stream count and the state block from the engine: both are swapped for the duration of the build.
    struct Swap {
        rvc_engine *e; int n0; StreamState *s0; bool on;
        Swap(rvc_engine *e_, int B, StreamState *st) : e(e_), n0(e_->n_streams), s0(e_->d_state), on(B > 0) { if (on) { e->n_streams = B; e->d_state = st; } }
        ~Swap() { if (on) { e->n_streams = n0; e->d_state = s0; } }
    } swap_guard(e, bucket_B, e->d_state_bucket);
    const int B = e->n_streams;
    const bool with_index = mode == 0 && e->d_index && e->index_rate > 0.f;
    for (auto &p : e->plans)
        if (p->mode == mode && p->L == L && p->frame16k == frame16k && p->skip_head == skip_head && p->R == R && p->B == B &&
            p->with_index == with_index && p->with_taps == (e->taps_on != 0) && p->plain_plan == (e->taps_on == 1) && p->slot == slot && p->bucket == (bucket_B > 0))
            return p.get();
    std::unique_ptr<Plan> up(new Plan());
    Plan &pl = *up;
    pl.mode = mode; pl.L = L; pl.frame16k = frame16k; pl.skip_head = skip_head; pl.R = R; pl.B = B; pl.with_index = with_index; pl.with_taps = e->taps_on != 0; pl.plain_plan = e->taps_on == 1; pl.bucket = bucket_B > 0;
    pl.slot = slot;
    pl.d_in = pl.arena.floats((size_t)B * L + 64);
    T1 sal0, src0; float *d_pitchf0 = nullptr; int *d_pitch0 = nullptr;
    size_t rm_begin = 0, rm_end = 0;
    const bool part = mode == 0 && e->partitioned;
    const int f0_sid = 1, cv_sid = part ? 3 : 0;
    if (mode == 0) {
        // f0 branch first (auxiliary stream): independent of ContentVec until the synthesizer
        const int Tcv = e->cv->out_frames(L);
        if (Tcv < 1) throw ShapeError("input too short for ContentVec");
        const size_t hubert_length0 = std::min(L / 160, 2 * (size_t)Tcv + 1);   // rvc.rs:153
        pl.ops.fork(f0_sid);
        rm_begin = pl.ops.v.size();
        pl.ops.cur = f0_sid;
        sal0 = build_rmvpe(e, pl, B, L, frame16k, true);
        build_pitch_post(e, pl, B, sal0, true, frame16k, hubert_length0, &d_pitchf0, &d_pitch0);
        pl.ops.cur = 0;
        rm_end = pl.ops.v.size();
    }
    if (mode == 0 || mode == 1) {
        if (!e->cv) throw std::logic_error("contentvec");
        if (e->cv->out_frames(L) < 1) throw ShapeError("input too short for ContentVec");
        if (cv_sid) { pl.ops.fork(cv_sid); pl.ops.cur = cv_sid; }
        pl.cv_out = build_contentvec(e, pl, B, L);
        if (mode == 1) {
            const int T = pl.T, C = pl.C;
            pl.d_feat = pl.arena.floats((size_t)(2 * T + 1) * C);
            T1 cvo = pl.cv_out; float *df = pl.d_feat;
            dim3 grid(((2 * T + 1) * C + 255) / 256);
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(extract_feature_kernel, grid, dim3(256), 0, s, cvo.p, cvo.ld, C, T, df); });
        }
    }
    if (mode == 2) {
        T1 sal = build_rmvpe(e, pl, B, L, frame16k, false);
        float *pf; int *pi;
        build_pitch_post(e, pl, B, sal, false, frame16k, 0, &pf, &pi);
    }
    if (mode == 0) {
        const int T = pl.T, C = pl.C;
        const size_t T2 = 2 * (size_t)T + 1;
        const size_t hubert_length = std::min(L / 160, T2);                 // rvc.rs:153
        if ((size_t)skip_head + R > T2) throw PanicError("feature slice out of range (rvc.rs:155)");
        if (R < 1) throw ShapeError("return_length must be >= 1");
        if (e->sy->phone_dim != C) throw std::runtime_error("synthesizer phone dimension does not match the ContentVec output");
        T1 phone = make_t1(pl.arena, B, C, (int)R, 0);
        {
            T1 cvo = pl.cv_out; dim3 grid((C * (int)R + 255) / 256, B);
            pl.ops.push_back([=](hipStream_t s) {
                hipLaunchKernelGGL(gather_phone_kernel, grid, dim3(256), 0, s, cvo.p, cvo.ld, cvo.bs, C, T, (int)skip_head, (int)R, phone.p, phone.ld, phone.bs);
            });
        }
        if (with_index) {
            if (e->index_dim != (size_t)C) throw std::runtime_error("index dimension does not match the feature dimension");
            // unique raw frames behind the sliced frames (Q2): first_raw .. last_raw
            const int first_raw = std::min((int)skip_head / 2, T - 1), last_raw = std::min((int)(skip_head + R - 1) / 2, T - 1);
            const int nq = last_raw - first_raw + 1;
            float *d_q = pl.arena.floats((size_t)B * nq * C);
            const int nblk = (int)((e->index_n + 255) / 256);
            float *cand_d = pl.arena.floats((size_t)B * nq * nblk * KNN_K);
            int *cand_i = (int *)pl.arena.alloc((size_t)B * nq * nblk * KNN_K * sizeof(int));
            pl.d_knn_idx = (int *)pl.arena.alloc((size_t)B * R * KNN_K * sizeof(int));
            pl.d_knn_dist = pl.arena.floats((size_t)B * R * KNN_K);
            T1 cvo = pl.cv_out;
            {
                dim3 grid((nq * C + 255) / 256, B);
                pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(knn_queries_kernel, grid, dim3(256), 0, s, cvo.p, cvo.ld, cvo.bs, C, first_raw, nq, d_q); });
            }
            // Stage A + B: approximate distances on the matrix cores in one pass over the index (HBM-bound), exact re-rank of a
            // provably sufficient candidate set; the exhaustive exact scan below only runs for streams whose candidate set overflowed.
            const bool fast = C % 16 == 0 && !test_opt("RVC_KNN_EXHAUSTIVE");
            int *d_overflow = (int *)pl.arena.alloc((size_t)B * sizeof(int));
            // many streams: all queries against the index as ONE implicit GEMM (queries = weight operand in fragment order, transposed
            // index = activation operand, -|y|^2 / 2 as a per-column residual, scale -2): one pass over the index instead of one per 16
            // queries (64 streams x 11 queries: 44 passes, 3.5 ms -> one ~1 ms MFMA-bound launch).  Same approximate distances up to
            // fp32 summation order; the exact re-rank behind it is unchanged.
            const int Q = B * nq, Qpad = (Q + 127) / 128 * 128;
            // (the GEMM path addresses its operands with 32-bit byte / element offsets: the knn_dot loop, whose strides are 64-bit, takes
            // indexes beyond that range)
            const bool gemm_fits = (size_t)C * e->index_n * sizeof(float) < ((size_t)1 << 31) && (size_t)Qpad * e->index_n < ((size_t)1 << 31);
            const bool gemm_scan = fast && Q >= 128 && gemm_fits && e->d_nhn && !test_opt("RVC_KNN_NO_GEMM");
            if (gemm_scan || !fast) ensure_index_transposed(e);
            if (fast) {
                float *d_approx = pl.arena.floats((size_t)(gemm_scan ? Qpad : Q) * e->index_n);
                if (gemm_scan) {
                    float *d_qf = pl.arena.floats((size_t)Qpad * C);
                    const int n_idx = (int)e->index_n;
                    {
                        dim3 grid(Qpad / 16, C / 16); int *ovf = d_overflow; const int nb = B;
                        pl.ops.push_back([=](hipStream_t s) {
                            HIPCHK(hipMemsetAsync(ovf, 0, (size_t)nb * sizeof(int), s));
                            hipLaunchKernelGGL(knn_pack_queries_kernel, grid, dim3(64), 0, s, d_q, Q, C, d_qf);
                        });
                    }
                    ConvW qw; qw.w = d_qf; qw.bias = nullptr; qw.M = Qpad; qw.K = C; qw.Kp = C; qw.Cin = C; qw.Cout = Qpad; qw.KW = 1; qw.groups = 1; qw.nphase = 1; qw.owns = false;
                    T1 xi; xi.p = e->d_indexT; xi.B = 1; xi.C = C; xi.T = n_idx; xi.ld = n_idx; xi.halo = 0; xi.bs = (long long)C * n_idx;
                    T1 ya; ya.p = d_approx; ya.B = 1; ya.C = Qpad; ya.T = n_idx; ya.ld = n_idx; ya.halo = 0; ya.bs = (long long)Qpad * n_idx;
                    ConvOpts o; o.no_bias = true; o.res = e->d_nhn; o.res_cs = 0; o.res_bs = 0; o.scale = -2.0f;
                    add_conv1d(pl, qw, xi, ya, 1, 0, 1, o);
                }
                // per-wave candidate lists of the one-pass scan (one stream / few streams: the select stage reads n / 4 entries per query)
                const long long nwaves = ((long long)e->index_n + 15) / 16;
                float *wl_d = nullptr; int *wl_i = nullptr;
                if (!gemm_scan && !tune_env("RVC_KNN_NO_WAVE_LISTS")) {
                    wl_d = pl.arena.floats((size_t)B * nq * nwaves * 4);
                    wl_i = (int *)pl.arena.alloc((size_t)B * nq * nwaves * 4 * sizeof(int));
                }
                for (int q0 = 0; q0 < nq && !gemm_scan; q0 += 16) {
                    KnnDotP dp{}; dp.indexF = e->d_indexF; dp.wl_d = wl_d; dp.wl_i = wl_i; dp.wl_bs = (long long)nq * nwaves * 4; dp.ynorm = e->d_ynorm; dp.n = (int)e->index_n; dp.dim = C;
                    dp.q = d_q; dp.q_bs = (long long)nq * C; dp.nq = nq; dp.q0 = q0; dp.approx = d_approx; dp.approx_bs = (long long)nq * e->index_n;
                    dp.overflow = d_overflow;
                    // persistent grid (the waves walk the index tiles; measured: 256 / 512 / 768 / 1024 / one tile per wave = 100 / 79 / 86 / 73 / 74 us per 307 MB)
                    static const unsigned knn_wgs = tune_env("RVC_KNN_WGS") ? (unsigned)atoi(tune_env("RVC_KNN_WGS")) : 1024u;
                    dim3 grid(std::min((unsigned)((e->index_n + 63) / 64), std::max(knn_wgs / (unsigned)B, 64u)), B);
                    const size_t qlds = (size_t)16 * (C + 4) * sizeof(float);
                    if (qlds > 160 * 1024) throw ShapeError("feature dimension too large for the retrieval kernel");
                    Plan *plp = &pl;
                    const double scan_bytes = (double)e->index_n * C * sizeof(float) * B;     // algorithmic bytes: the index, read once per query group
                    pl.ops.push_back([=](hipStream_t s) {
                        ProfEvent *pe = nullptr;
                        if (plp->profile) {
                            if (plp->prof_used == plp->prof.size()) { ProfEvent ev; HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b)); ev.flops = 0; ev.bytes = 0; plp->prof.push_back(ev); }
                            pe = &plp->prof[plp->prof_used++]; pe->flops = 0; pe->bytes = scan_bytes;
                        }
                        if (pe) hipExtLaunchKernelGGL(knn_dot_kernel, grid, dim3(256), (uint32_t)qlds, s, pe->a, pe->b, 0, dp);
                        else hipLaunchKernelGGL(knn_dot_kernel, grid, dim3(256), qlds, s, dp);
                    });
                }
                KnnSelP sp{}; sp.approx = d_approx; sp.approx_bs = (long long)nq * e->index_n; sp.n = (int)e->index_n; sp.dim = C; sp.nq = nq;
                sp.index = e->d_index; sp.q = d_q; sp.q_bs = (long long)nq * C; sp.skip_head = (int)skip_head; sp.T = T; sp.R = (int)R; sp.first_raw = first_raw;
                sp.rate = e->index_rate; sp.phone = phone.p; sp.ph_cs = phone.ld; sp.ph_bs = phone.bs; sp.out_idx = pl.d_knn_idx; sp.out_dist = pl.d_knn_dist;
                sp.overflow = d_overflow; sp.wl_d = wl_d; sp.wl_i = wl_i; sp.wl_bs = (long long)nq * nwaves * 4;
                dim3 sgrid(nq, B);
                const size_t slds = (size_t)33 * (C + 4) * sizeof(float);
                if (slds > 128 * 1024) throw ShapeError("feature dimension too large for the retrieval kernel");
                pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(knn_select_blend_kernel, sgrid, dim3(1024), slds, s, sp); });
            }
            for (int q0 = 0; q0 < nq; q0 += KNN_MAXQ) {
                const int qn = std::min(KNN_MAXQ, nq - q0);
                KnnP kp{}; kp.indexT = e->d_indexT; kp.index = e->d_index; kp.n = (int)e->index_n; kp.dim = C; kp.nblk = nblk;
                kp.v_stride = e->d_indexT ? 1 : C; kp.d_stride = e->d_indexT ? (long long)e->index_n : 1;
                // query sub-range: pointers offset so that [B][nq] strides stay those of the full arrays
                kp.q = d_q + (size_t)q0 * C; kp.nq = qn; kp.cand_d = cand_d + (size_t)q0 * nblk * KNN_K; kp.cand_i = cand_i + (size_t)q0 * nblk * KNN_K;
                kp.overflow = fast ? d_overflow : nullptr;
                const int nq_total = nq;
                dim3 grid(nblk, B);
                pl.ops.push_back([=](hipStream_t s) {
                    KnnP k2 = kp; k2.q_bs = (long long)nq_total * C; k2.cand_bs = (long long)nq_total * nblk * KNN_K;
                    hipLaunchKernelGGL(knn_scan_kernel, grid, dim3(256), 0, s, k2);
                });
            }
            KnnBlendP bp{}; bp.cand_d = cand_d; bp.cand_i = cand_i; bp.nblk = nblk; bp.nq = nq; bp.index = e->d_index; bp.dim = C; bp.q = d_q;
            bp.skip_head = (int)skip_head; bp.T = T; bp.R = (int)R; bp.first_raw = first_raw; bp.rate = e->index_rate;
            bp.phone = phone.p; bp.ph_cs = phone.ld; bp.ph_bs = phone.bs; bp.out_idx = pl.d_knn_idx; bp.out_dist = pl.d_knn_dist;
            bp.overflow = fast ? d_overflow : nullptr;
            dim3 grid(nq, B);
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(knn_merge_blend_kernel, grid, dim3(256), 0, s, bp); });
        }
        add_tap(pl, "phone_ct", phone);
        float *d_pitchf = d_pitchf0; int *d_pitch = d_pitch0;
        (void)hubert_length;
        pl.ops.cur = 0;
        if (rm_end > rm_begin) {
            // Two concurrent branches: f0 (ops [rm_begin, rm_end), aux stream 1) and ContentVec (ops [rm_end, here), main stream).
            // Launches reach the hardware queues in host order (~3 us each).  Eager: interleave the branches in proportion to
            // their lengths so both queues are fed from the start (the f0 branch is the longer one once they share the chip).
            // Graph replay submits a whole branch at a time: put ContentVec first, the f0 branch then starts ~0.2 ms late.
            OpList &ol = pl.ops;
            const size_t cv_end = ol.v.size(), na = rm_end - rm_begin, nb = cv_end - rm_end;
            for (size_t i = 0; i < rm_begin; i++) { ol.order_eager.push_back((int)i); ol.order_graph.push_back((int)i); }
            size_t ia = 0, ib = 0;
            while (ia < na || ib < nb) {
                const bool take_a = ib >= nb || (ia < na && (2 * ia + 1) * nb <= (2 * ib + 1) * na);
                ol.order_eager.push_back((int)(take_a ? rm_begin + ia++ : rm_end + ib++));
            }
            for (size_t i = rm_end; i < cv_end; i++) ol.order_graph.push_back((int)i);
            for (size_t i = rm_begin; i < rm_end; i++) ol.order_graph.push_back((int)i);
        }
        if (cv_sid) pl.ops.join(cv_sid);
        pl.ops.join(f0_sid);
        // the NSF harmonic source is first needed by the decoder: it runs on a side stream next to the text encoder and the flow
        pl.ops.fork(2); pl.ops.cur = 2;
        src0 = build_nsf_source(e, pl, B, d_pitchf0);
        std::vector<T1> nz;
        const bool side_nz = !tune_env("RVC_NO_SIDE_NOISE_CONVS");
        if (side_nz) nz = build_noise_convs(e, pl, B, src0);
        pl.ops.cur = 0;
        build_synth(e, pl, B, phone, src0, d_pitchf, d_pitch, 2, side_nz ? &nz : nullptr);
        StreamState *st = e->d_state;
        int *hst = e->h_status;      // (pinned host memory, mapped: the kernel writes the status words where the host reads them)
        pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(advance_chunk_kernel, dim3((B + 63) / 64), dim3(64), 0, s, st, B, hst); });
    }
    HIPCHK(hipDeviceSynchronize());
    // bounded plan cache (each geometry owns its activation arena and graph): evict the oldest
    while (e->plans.size() >= 8) { if (e->last_plan == e->plans.front().get()) e->last_plan = nullptr; e->plans.erase(e->plans.begin()); }
    e->plans.push_back(std::move(up));
    return e->plans.back().get();
}

static void issue_ops(rvc_engine *e, Plan &pl, bool capturing)
{
    // fork/join through events; under stream capture the auxiliary streams become parallel branches of the hipGraph
    const std::vector<int> &ord = capturing ? pl.ops.order_graph : pl.ops.order_eager;
    const size_t n = pl.ops.v.size();
    // Every op on the main stream: as a tuning aid (environment), and for the per-launch profile when the branches share the CUs
    // (more than 4 streams, no partition).  There a short f0 kernel that is co-scheduled with a 3 ms ContentVec GEMM is stretched to
    // the GEMM's length by the workgroup dispatcher (19 us alone, 3158 us measured); its HIP-event duration then says nothing about
    // the kernel.  With a CU partition the branches own disjoint CUs and are profiled as they run.
    const bool serial_env = test_opt("RVC_SERIAL_BRANCHES") != nullptr;
    const bool serial = serial_env || (pl.profile && !capturing && !e->partitioned);
    for (size_t k = 0; k < n; k++) {
        const size_t i = (k < ord.size() && !serial) ? (size_t)ord[k] : k;      // ops queued after the reordered prefix keep their position
        const int sid = serial ? 0 : pl.ops.sid[i];
        if (serial && pl.ops.kind[i] != 0) continue;
        hipStream_t st = sid == 0 ? e->stream : e->aux[sid - 1];
        if (pl.ops.kind[i] == 1) {
            if (e->pipe_now && (sid == 1 || sid == 3)) continue;      // pipelined: the front branches are ordered by events, not by the main stream
            HIPCHK(hipEventRecord(e->ev_fork[sid - 1], e->stream));
            HIPCHK(hipStreamWaitEvent(st, e->ev_fork[sid - 1], 0));
        } else if (pl.ops.kind[i] == 2) {
            HIPCHK(hipEventRecord(e->ev_join[sid - 1], st));
            HIPCHK(hipStreamWaitEvent(e->stream, e->ev_join[sid - 1], 0));
        } else {
            pl.ops.v[i](st);
        }
    }
}

static void run_plan(rvc_engine *e, Plan &pl)
{
    pl.profile = e->profile_on;
    pl.prof_used = 0;
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    if (e->use_graph && !e->profile_on) {
        if (!pl.graph_exec) {
            hipGraph_t g;
            HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
            issue_ops(e, pl, true);
            HIPCHK(hipStreamEndCapture(e->stream, &g));
            HIPCHK(hipGraphInstantiate(&pl.graph_exec, g, nullptr, nullptr, 0));
            HIPCHK(hipGraphDestroy(g));
        }
        HIPCHK(hipGraphLaunch(pl.graph_exec, e->stream));
    } else {
        issue_ops(e, pl, false);
    }
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    HIPCHK(hipGetLastError());
    e->last_plan = &pl;
}

static float uppower(int32_t pitch_shift) { return ldexpf(1.0f, pitch_shift / 12); }   // rvc.rs:121, truncating division (Q1)

// Per-call parameters: the seed (CallParams) and one pitch-shift multiplier PER STREAM (StreamState::uppower: every stream of a batch
// is its own caller with its own settings, obs-rvc/src/lib.rs:701-707).  shifts == nullptr: `pitch_shift` for every stream.
static void push_call_params(rvc_engine *e, int32_t pitch_shift, const int32_t *shifts = nullptr)
{
    const int B = e->n_streams;
    // The device already holds these values (every write to them is ordered on the main stream, and the last one wrote exactly this):
    // nothing to copy -- a small H2D copy is a 4-5 us blit kernel in front of both branches of every chunk otherwise.
    bool same = e->pushed_valid && e->pushed_seed == e->seed && (int)e->pushed_up.size() == B;
    for (int b = 0; b < B && same; b++) same = e->pushed_up[b] == uppower(shifts ? shifts[b] : pitch_shift);
    if (same) return;
    if (e->pipeline) HIPCHK(hipDeviceSynchronize());   // the f0 branch of the next chunk may already be running: drain before the values change
    // every call writes its own pinned block: an unsynchronised call's copy may still be pending when the next call arrives
    if (!e->pushed_valid || e->pushed_seed != e->seed) {
        CallParams *h = e->h_cp + (e->cp_slot++ & 63u);
        h->uppower = uppower(shifts ? shifts[0] : pitch_shift);
        h->seed = e->seed;
        h->chunk_base = 0;
        HIPCHK(hipMemcpyAsync(e->d_cp, h, sizeof(CallParams), hipMemcpyHostToDevice, e->stream));
    }
    float *hu = e->h_up + (size_t)(e->up_slot++ & 7u) * 4096;
    e->pushed_up.resize(B);
    for (int b = 0; b < B; b++) hu[b] = e->pushed_up[b] = uppower(shifts ? shifts[b] : pitch_shift);
    HIPCHK(hipMemcpy2DAsync((char *)e->d_state + offsetof(StreamState, uppower), sizeof(StreamState), hu, sizeof(float), sizeof(float), (size_t)B, hipMemcpyHostToDevice, e->stream));
    e->pushed_seed = e->seed; e->pushed_valid = true;
    if (e->pipeline) {
        // pipelined calls do not fork the front branches from the main stream: order them behind the parameter copy explicitly
        HIPCHK(hipEventRecord(e->ev_cp, e->stream));
        HIPCHK(hipStreamWaitEvent(e->aux[0], e->ev_cp, 0));
        HIPCHK(hipStreamWaitEvent(e->aux[2], e->ev_cp, 0));
    }
}

// Status words of all streams (0 ok, 6 = the reference would have panicked at rmvpe.rs:124, 7 = GRU hand-off time-out): one strided
// copy into pinned memory, queued in front of the call's final synchronisation so that reading them costs no extra round trip.
static void queue_status(rvc_engine *e)
{
    HIPCHK(hipMemcpy2DAsync(e->h_status, sizeof(int), (char *)e->d_state + offsetof(StreamState, status), sizeof(StreamState), sizeof(int), (size_t)e->n_streams,
                            hipMemcpyDeviceToHost, e->stream));
    e->status_queued = true;
}

static rvc_status check_status(rvc_engine *e)
{
    if (!e->status_queued) { queue_status(e); HIPCHK(hipStreamSynchronize(e->stream)); }
    e->status_queued = false;
    for (int b = 0; b < e->n_streams; b++)
        if (e->h_status[b] != 0) {
            const int code = e->h_status[b];
            int zero = 0;
            for (int c = b; c < e->n_streams; c++)
                if (e->h_status[c] != 0) HIPCHK(hipMemcpy((char *)(e->d_state + c) + offsetof(StreamState, status), &zero, sizeof(int), hipMemcpyHostToDevice));
            if (code == 7) { e->err = "a cross-workgroup hand-off timed out (GRU recurrence)"; return RVC_BACKEND; }
            e->err = "to_local_average_cents: index out of bounds (argmax bin >= 348), the reference panics here";
            return RVC_PANIC;
        }
    return RVC_OK;
}

template <typename Fn> static rvc_status guarded(rvc_engine *e, Fn fn)
{
    if (!e) return RVC_BACKEND;
    try {
        set_device(e);
        return fn();
    } catch (const ShapeError &x) { e->err = x.what(); return RVC_SHAPE; }
    catch (const PanicError &x) { e->err = x.what(); return RVC_PANIC; }
    catch (const std::exception &x) { e->err = x.what(); return RVC_BACKEND; }
}

static std::string native_path(const std::string &p)
{
    if (p.size() > 5 && p.substr(p.size() - 5) == ".onnx") return p.substr(0, p.size() - 5) + ".rvcw";
    return p;
}

}  // namespace rvc

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

// rvc_version(): version.cpp

rvc_status rvc_create(const char *data_path, int device, rvc_engine **out)
{
    if (!out) return RVC_BACKEND;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return RVC_BACKEND;   // no CPU fallback, by design
    rvc_engine *e = new rvc_engine();
    e->data_path = data_path ? data_path : "";
    if (device < 0) { const char *lr = getenv("LOCAL_RANK"); device = lr ? atoi(lr) % ndev : 0; }
    e->device = device;
    try {
        set_device(e);
        for (int i = 0; i < 3; i++) { HIPCHK(hipEventCreateWithFlags(&e->ev_fork[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming)); }
        e->partition_ok = !tune_env("RVC_NO_CUMASK");
        configure_aux_streams(e);
        HIPCHK(hipEventCreate(&e->ev0)); HIPCHK(hipEventCreate(&e->ev1));
        HIPCHK(hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming));
        HIPCHK(hipMalloc(&e->d_cp, sizeof(CallParams)));
        HIPCHK(hipHostMalloc((void **)&e->h_cp, 64 * sizeof(CallParams)));
        HIPCHK(hipEventCreateWithFlags(&e->ev_cp, hipEventDisableTiming));
        HIPCHK(hipHostMalloc((void **)&e->h_status, 4096 * sizeof(int)));
        memset(e->h_status, 0, 4096 * sizeof(int));
        HIPCHK(hipHostMalloc((void **)&e->h_up, (size_t)8 * 4096 * sizeof(float)));
        init_constants(e);
        alloc_state(e);
    } catch (const std::exception &x) {
        fprintf(stderr, "rvc_create: %s\n", x.what());
        release_stream_set(e->sset);
        delete e;
        return RVC_BACKEND;
    }
    *out = e;
    return RVC_OK;
}

void rvc_destroy(rvc_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    e->plans.clear();
    e->cv.reset(); e->rm.reset(); e->sy.reset();
    if (e->d_window) (void)hipFree(e->d_window);
    if (e->d_twiddle) (void)hipFree(e->d_twiddle);
    if (e->d_basis) (void)hipFree(e->d_basis);
    if (e->d_band) (void)hipFree(e->d_band);
    if (e->d_index && e->index_owned) (void)hipFree(e->d_index);
    if (e->d_indexT) (void)hipFree(e->d_indexT);
    if (e->d_ynorm) (void)hipFree(e->d_ynorm);
    if (e->d_nhn) (void)hipFree(e->d_nhn);
    if (e->d_indexF) (void)hipFree(e->d_indexF);
    if (e->d_state) (void)hipFree(e->d_state);
    if (e->d_state_bucket) (void)hipFree(e->d_state_bucket);
    if (e->d_bucket_idx) (void)hipFree(e->d_bucket_idx);
    if (e->d_cp) (void)hipFree(e->d_cp);
    if (e->h_cp) (void)hipHostFree(e->h_cp);
    if (e->h_status) (void)hipHostFree(e->h_status);
    if (e->h_up) (void)hipHostFree(e->h_up);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_cp) (void)hipEventDestroy(e->ev_cp);
    for (int i = 0; i < 3; i++) {
        if (e->ev_fork[i]) (void)hipEventDestroy(e->ev_fork[i]);
        if (e->ev_join[i]) (void)hipEventDestroy(e->ev_join[i]);
    }
    release_stream_set(e->sset);      // (the streams stay in the per-device pool)
    delete e;
}

const char *rvc_last_error_message(rvc_engine *e) { return e ? e->err.c_str() : "null engine"; }
int rvc_device(rvc_engine *e) { return e ? e->device : -1; }

// rvc.rs:46-54 + models.rs:52-64
rvc_status rvc_load_contentvec(rvc_engine *e, int model_version)
{
    return guarded(e, [&]() {
        const int dim = model_version == RVC_VERSION_V1 ? 256 : 768, layer = model_version == RVC_VERSION_V1 ? 9 : 12;   // enums.rs:10-23
        char name[64]; snprintf(name, sizeof name, "vec-%d-layer-%d.rvcw", dim, layer);
        Blob b(e->data_path + "/contentvec/" + name);
        HIPCHK(hipDeviceSynchronize());      // unsynchronised calls may still be running on the plans freed below
        e->plans.clear(); e->last_plan = nullptr;
        e->cv.reset(new ModelCV(b));
        return RVC_OK;
    });
}
// rvc.rs:56-60
rvc_status rvc_load_model(rvc_engine *e, const char *model_path)
{
    return guarded(e, [&]() {
        Blob b(native_path(model_path ? model_path : ""));
        HIPCHK(hipDeviceSynchronize());      // unsynchronised calls may still be running on the plans freed below
        e->plans.clear(); e->last_plan = nullptr;
        e->sy.reset(new ModelSY(b));
        return RVC_OK;
    });
}
// rvc.rs:62-75 + models.rs:66-76
rvc_status rvc_load_f0(rvc_engine *e, int pitch_algorithm)
{
    (void)pitch_algorithm;   // only Rmvpe exists (enums.rs:26-28); unknown values map to it (enums.rs:96-103)
    return guarded(e, [&]() {
        Blob b(e->data_path + "/f0/rmvpe.rvcw");
        HIPCHK(hipDeviceSynchronize());      // unsynchronised calls may still be running on the plans freed below
        e->plans.clear(); e->last_plan = nullptr;
        e->rm.reset(new ModelRM(b));
        return RVC_OK;
    });
}
// rvc.rs:77-79
void rvc_unload_model(rvc_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    e->plans.clear(); e->last_plan = nullptr;
    e->sy.reset();
}

static rvc_status run_single_input(rvc_engine *e, Plan *pl, const float *input, size_t n, int32_t pitch_shift)
{
    HIPCHK(hipMemcpyAsync(pl->d_in, input, n * sizeof(float) * (size_t)pl->B, hipMemcpyHostToDevice, e->stream));
    push_call_params(e, pitch_shift);
    run_plan(e, *pl);
    return RVC_OK;
}

// rvc.rs:81-97
rvc_status rvc_hubert(rvc_engine *e, const float *input, size_t n, float *out, size_t cap, size_t dims[3])
{
    return guarded(e, [&]() {
        if (!e->cv) return RVC_CONTENTVEC_NOT_LOADED;
        if (e->n_streams != 1) throw ShapeError("hubert() is a single-stream call");
        Plan *pl = get_plan(e, 1, n, 0, 0, 0);
        dims[0] = 1; dims[1] = (size_t)pl->C; dims[2] = (size_t)pl->T;
        if (cap < (size_t)pl->C * pl->T) return RVC_SHAPE;
        run_single_input(e, pl, input, n, 0);
        HIPCHK(hipMemcpy2DAsync(out, (size_t)pl->T * 4, pl->cv_out.p, (size_t)pl->cv_out.ld * 4, (size_t)pl->T * 4, pl->C, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        return RVC_OK;
    });
}
// rvc.rs:99-109
rvc_status rvc_extract_feature(rvc_engine *e, const float *input, size_t n, float *out, size_t cap, size_t dims[3])
{
    return guarded(e, [&]() {
        if (!e->cv) return RVC_CONTENTVEC_NOT_LOADED;
        if (e->n_streams != 1) throw ShapeError("extract_feature() is a single-stream call");
        Plan *pl = get_plan(e, 1, n, 0, 0, 0);
        const size_t T2 = 2 * (size_t)pl->T + 1;
        dims[0] = 1; dims[1] = T2; dims[2] = (size_t)pl->C;
        if (cap < T2 * pl->C) return RVC_SHAPE;
        run_single_input(e, pl, input, n, 0);
        HIPCHK(hipMemcpyAsync(out, pl->d_feat, T2 * pl->C * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        return RVC_OK;
    });
}
// rvc.rs:111-131
rvc_status rvc_pitch(rvc_engine *e, const float *input, size_t n, int32_t pitch_shift, size_t sample_frame_16k_size, float *out, size_t cap, size_t *out_len)
{
    return guarded(e, [&]() {
        if (!e->rm) return RVC_F0_NOT_LOADED;   // the reference hits unreachable!() here (rvc.rs:125)
        if (e->n_streams != 1) throw ShapeError("pitch() is a single-stream call");
        Plan *pl = get_plan(e, 2, n, sample_frame_16k_size, 0, 0);
        *out_len = (size_t)pl->Tm;
        if (cap < (size_t)pl->Tm) return RVC_SHAPE;
        run_single_input(e, pl, input, n, pitch_shift);
        HIPCHK(hipMemcpyAsync(out, pl->d_f0, (size_t)pl->Tm * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        queue_status(e);
        HIPCHK(hipStreamSynchronize(e->stream));
        return check_status(e);
    });
}

static rvc_status infer_common(rvc_engine *e, const void *input, bool input_on_device, size_t n, size_t frame16k, int32_t pitch_shift,
                               uint32_t skip_head, uint32_t return_length, void *out, bool out_on_device, size_t cap, size_t *out_len, bool sync,
                               const int32_t *shifts = nullptr)
{
    if (!e->sy) return RVC_MODEL_NOT_LOADED;             // rvc.rs:141-143
    if (!e->cv) return RVC_CONTENTVEC_NOT_LOADED;        // rvc.rs:85-88 (via extract_feature at rvc.rs:151)
    if (!e->rm) return RVC_F0_NOT_LOADED;                // reference: unreachable!() at rvc.rs:125
    const bool pipe = e->pipeline && !sync && input_on_device && out_on_device && e->partitioned && !e->use_graph && !e->profile_on && !e->taps_on;
    Plan *pl = get_plan(e, 0, n, frame16k, skip_head, return_length, pipe ? (e->pipe_slot ^= 1) : 0);
    if (out_len) *out_len = pl->N;
    if (cap < pl->N) return RVC_SHAPE;
    const int B = pl->B;
    push_call_params(e, pitch_shift, shifts);
    // Device-resident callers: the first kernels read the caller's buffer and the last one writes the caller's buffer (eager launches
    // take the pointers at launch time) -- no staging copy in front of the chunk, no copy behind it.  A captured graph bakes its
    // pointers and keeps both copies; pipelined calls keep the input copy (it decouples the caller's buffer from the chunk in flight).
    static const bool no_direct = tune_env("RVC_NO_DIRECT_IO") != nullptr;
    const bool direct_in = input_on_device && !pipe && !e->use_graph && pl->in_direct_ok && !no_direct;
    const bool direct_out = out_on_device && !e->use_graph && pl->out_direct_ok && !no_direct;
    pl->cur_in = direct_in ? (const float *)input : nullptr;
    pl->cur_out = direct_out ? (float *)out : nullptr; pl->cur_out_bs = (long long)cap;
    if (pipe) {
        // chunk pipelining: the two front branches of this chunk start as soon as THEIR previous work and this plan slot's previous
        // chunk are done -- not after the previous chunk's synthesizer on the main stream
        hipStream_t cvs = e->aux[2], rms = e->aux[0];
        if (pl->ev_done_valid) { HIPCHK(hipStreamWaitEvent(cvs, pl->ev_done, 0)); HIPCHK(hipStreamWaitEvent(rms, pl->ev_done, 0)); }
        HIPCHK(hipMemcpyAsync(pl->d_in, input, (size_t)B * n * sizeof(float), hipMemcpyDeviceToDevice, cvs));
        HIPCHK(hipEventRecord(e->ev_in, cvs));
        HIPCHK(hipStreamWaitEvent(rms, e->ev_in, 0));
    } else if (!direct_in) {
        HIPCHK(hipMemcpyAsync(pl->d_in, input, (size_t)B * n * sizeof(float), input_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->stream));
    }
    e->pipe_now = pipe;
    run_plan(e, *pl);
    e->pipe_now = false;
    if (!direct_out)
        HIPCHK(hipMemcpy2DAsync(out, cap * sizeof(float), pl->audio.p, pl->N * sizeof(float), pl->N * sizeof(float), B,
                                out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e->stream));
    if (!pl->ev_done) HIPCHK(hipEventCreateWithFlags(&pl->ev_done, hipEventDisableTiming));
    HIPCHK(hipEventRecord(pl->ev_done, e->stream));
    pl->ev_done_valid = true;
    e->last_knn_rows = pl->with_index ? return_length : 0;
    e->status_queued = true;        // the chunk's last kernel writes the status words into the host-mapped block (advance_chunk_kernel)
    if (!sync) return RVC_OK;
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipEventElapsedTime(&e->last_ms, e->ev0, e->ev1));
    return check_status(e);
}

// rvc.rs:133-220
rvc_status rvc_infer(rvc_engine *e, const float *input, size_t n, size_t sample_frame_16k_size, int has_pitch_shift, int32_t pitch_shift,
                     uint32_t skip_head, uint32_t return_length, float *out, size_t cap, size_t *out_len)
{
    return guarded(e, [&]() {
        if (e->n_streams != 1) throw ShapeError("infer() is a single-stream call; use rvc_infer_batch");
        return infer_common(e, input, false, n, sample_frame_16k_size, has_pitch_shift ? pitch_shift : 0, skip_head, return_length, out, false, cap, out_len, true);
    });
}

rvc_status rvc_infer_batch(rvc_engine *e, const float *input, size_t n, size_t sample_frame_16k_size, int32_t pitch_shift, uint32_t skip_head,
                           uint32_t return_length, float *out, size_t cap_per_stream, size_t *out_len)
{
    return guarded(e, [&]() { return infer_common(e, input, false, n, sample_frame_16k_size, pitch_shift, skip_head, return_length, out, false, cap_per_stream, out_len, true); });
}

rvc_status rvc_infer_device(rvc_engine *e, const void *d_input, size_t n, size_t sample_frame_16k_size, int32_t pitch_shift, uint32_t skip_head,
                            uint32_t return_length, void *d_out, size_t cap_per_stream, size_t *out_len, int sync)
{
    return guarded(e, [&]() { return infer_common(e, d_input, true, n, sample_frame_16k_size, pitch_shift, skip_head, return_length, d_out, true, cap_per_stream, out_len, sync != 0); });
}

// many streams, every stream with its own pitch shift (each stream of a batch is a caller of its own: obs-rvc/src/lib.rs:701-707)
rvc_status rvc_infer_batch_v(rvc_engine *e, const float *input, size_t n, size_t sample_frame_16k_size, const int32_t *pitch_shift, uint32_t skip_head,
                             uint32_t return_length, float *out, size_t cap_per_stream, size_t *out_len)
{
    return guarded(e, [&]() {
        if (!pitch_shift) throw ShapeError("infer_batch_v: pitch_shift[n_streams] is required");
        return infer_common(e, input, false, n, sample_frame_16k_size, 0, skip_head, return_length, out, false, cap_per_stream, out_len, true, pitch_shift);
    });
}

rvc_status rvc_infer_device_v(rvc_engine *e, const void *d_input, size_t n, size_t sample_frame_16k_size, const int32_t *pitch_shift, uint32_t skip_head,
                              uint32_t return_length, void *d_out, size_t cap_per_stream, size_t *out_len, int sync)
{
    return guarded(e, [&]() {
        if (!pitch_shift) throw ShapeError("infer_device_v: pitch_shift[n_streams] is required");
        return infer_common(e, d_input, true, n, sample_frame_16k_size, 0, skip_head, return_length, d_out, true, cap_per_stream, out_len, sync != 0, pitch_shift);
    });
}

// Many streams, every stream with ITS OWN geometry.  In the reference every stream is a process of its own with its own chunk length,
// crossfade and extra context (obs-rvc/src/lib.rs:200-227, 694, 701-707); a server that batches such callers cannot ask them to agree on
// (n, sample_frame_16k_size, skip_head, return_length).  Streams with equal geometry form a bucket; every bucket runs as one batch through
// its own plan on a gathered copy of its streams' states (pitch cache, counters, status), which is scattered back afterwards.
rvc_status rvc_infer_batch_g(rvc_engine *e, const float *const *inputs, const size_t *n, const size_t *sample_frame_16k_size, const int32_t *pitch_shift,
                             const uint32_t *skip_head, const uint32_t *return_length, float *const *outs, const size_t *caps, size_t *out_lens)
{
    return guarded(e, [&]() {
        if (!inputs || !n || !sample_frame_16k_size || !skip_head || !return_length || !outs || !caps) throw ShapeError("infer_batch_g: null argument array");
        if (!e->sy) return RVC_MODEL_NOT_LOADED;
        if (!e->cv) return RVC_CONTENTVEC_NOT_LOADED;
        if (!e->rm) return RVC_F0_NOT_LOADED;
        const int S = e->n_streams;
        if (e->pipeline || e->use_graph) throw ShapeError("infer_batch_g: not with chunk pipelining / graph replay");
        struct Key { size_t n, f; uint32_t sh, rl; bool operator<(const Key &o) const { return std::tie(n, f, sh, rl) < std::tie(o.n, o.f, o.sh, o.rl); } };
        std::map<Key, std::vector<int>> buckets;
        for (int s = 0; s < S; s++) {
            if (!inputs[s] || !outs[s]) throw ShapeError("infer_batch_g: null stream buffer");
            buckets[Key{n[s], sample_frame_16k_size[s], skip_head[s], return_length[s]}].push_back(s);
        }
        if (!e->d_state_bucket) { HIPCHK(hipMalloc(&e->d_state_bucket, sizeof(StreamState) * S)); HIPCHK(hipMalloc(&e->d_bucket_idx, sizeof(int) * S)); }
        // plans first (a geometry the engine rejects must not leave some buckets already advanced), then the work
        std::vector<std::pair<Plan *, const std::vector<int> *>> work;
        for (auto &kv : buckets) {
            Plan *pl = get_plan(e, 0, kv.first.n, kv.first.f, kv.first.sh, kv.first.rl, 0, (int)kv.second.size());
            for (int s : kv.second) { if (out_lens) out_lens[s] = pl->N; if (caps[s] < pl->N) return RVC_SHAPE; }
            work.push_back({pl, &kv.second});
        }
        if (e->plans.size() < work.size()) throw ShapeError("infer_batch_g: more than 8 different geometries in one call");      // (the plan cache holds 8)
        for (auto &w : work)           // (a later get_plan may have evicted an earlier bucket's plan: every plan of this call must still be cached)
            { bool ok = false; for (auto &p : e->plans) ok = ok || p.get() == w.first; if (!ok) throw ShapeError("infer_batch_g: more than 8 different geometries in one call"); }
        push_call_params(e, 0, pitch_shift);            // per-stream multipliers into the streams' own states (a null array: no shift)
        for (auto &w : work) {
            Plan *pl = w.first; const std::vector<int> &ids = *w.second; const int Bk = (int)ids.size();
            HIPCHK(hipMemcpyAsync(e->d_bucket_idx, ids.data(), sizeof(int) * Bk, hipMemcpyHostToDevice, e->stream));
            hipLaunchKernelGGL(state_gather_kernel, dim3(Bk), dim3(256), 0, e->stream, e->d_state, e->d_state_bucket, e->d_bucket_idx, 0);
            for (int j = 0; j < Bk; j++) HIPCHK(hipMemcpyAsync(pl->d_in + (size_t)j * pl->L, inputs[ids[j]], pl->L * sizeof(float), hipMemcpyHostToDevice, e->stream));
            pl->cur_in = nullptr; pl->cur_out = nullptr;
            run_plan(e, *pl);
            for (int j = 0; j < Bk; j++) HIPCHK(hipMemcpyAsync(outs[ids[j]], pl->audio.p + (size_t)j * pl->N, pl->N * sizeof(float), hipMemcpyDeviceToHost, e->stream));
            hipLaunchKernelGGL(state_gather_kernel, dim3(Bk), dim3(256), 0, e->stream, e->d_state, e->d_state_bucket, e->d_bucket_idx, 1);
            HIPCHK(hipStreamSynchronize(e->stream));        // (ids / the pinned-less host buffers of this bucket are free again; the next bucket reuses the state block)
        }
        e->last_knn_rows = 0;
        queue_status(e);                                     // the streams' own status words (the plans wrote bucket-local ones)
        HIPCHK(hipStreamSynchronize(e->stream));
        return check_status(e);
    });
}

rvc_status rvc_synchronize(rvc_engine *e)
{
    return guarded(e, [&]() {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->ev0 && e->last_plan) (void)hipEventElapsedTime(&e->last_ms, e->ev0, e->ev1);
        return check_status(e);
    });
}

rvc_status rvc_set_streams(rvc_engine *e, int n_streams)
{
    return guarded(e, [&]() {
        if (n_streams < 1 || n_streams > 4096) throw ShapeError("n_streams out of range");
        HIPCHK(hipDeviceSynchronize());
        e->n_streams = n_streams;
        alloc_state(e);
        configure_aux_streams(e);
        return RVC_OK;
    });
}

void rvc_set_use_graph(rvc_engine *e, int on) { if (e) e->use_graph = on != 0; }
void rvc_set_pipeline(rvc_engine *e, int on)
{
    if (!e) return;
    (void)guarded(e, [&]() { HIPCHK(hipDeviceSynchronize()); e->pipeline = on != 0; e->pushed_valid = false; return RVC_OK; });
}
void rvc_set_profile(rvc_engine *e, int on) { if (e) e->profile_on = on != 0; }
void rvc_enable_taps(rvc_engine *e, int on) { if (e) e->taps_on = on == 2 ? 2 : (on != 0 ? 1 : 0); }
float rvc_last_gpu_ms(rvc_engine *e) { return e ? e->last_ms : 0.f; }
void rvc_set_index_rate(rvc_engine *e, float rate) { if (e) e->index_rate = rate; }

void rvc_set_noise_seed(rvc_engine *e, uint32_t seed, uint32_t stream_id)
{
    if (!e) return;
    (void)guarded(e, [&]() {
        HIPCHK(hipDeviceSynchronize());
        e->seed = seed; e->stream_id0 = stream_id;
        // keep pitch caches / chunk counters, only re-stamp the stream ids
        std::vector<StreamState> st(e->n_streams);
        HIPCHK(hipMemcpy(st.data(), e->d_state, sizeof(StreamState) * e->n_streams, hipMemcpyDeviceToHost));
        for (int b = 0; b < e->n_streams; b++) st[b].stream_id = stream_id + (uint32_t)b;
        HIPCHK(hipMemcpy(e->d_state, st.data(), sizeof(StreamState) * e->n_streams, hipMemcpyHostToDevice));
        return RVC_OK;
    });
}

void rvc_reset_state(rvc_engine *e)
{
    if (!e) return;
    (void)guarded(e, [&]() { HIPCHK(hipDeviceSynchronize()); reset_state(e); return RVC_OK; });
}

void rvc_get_pitch_cache(rvc_engine *e, int stream, float *out1024)
{
    if (!e || stream < 0 || stream >= e->n_streams) return;
    (void)guarded(e, [&]() {
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(out1024, e->d_state[stream].cache_pitchf, 1024 * sizeof(float), hipMemcpyDeviceToHost));
        return RVC_OK;
    });
}

}  // extern "C"

namespace rvc {

// Everything the retrieval kernels need besides the row-major matrix, built ON THE DEVICE from the copy that is already in HBM
// (uploaded once, or delivered by the RCCL broadcast): the MFMA-fragment-order copy for the one-pass approximate scan and the vector
// norms.  No host round trip (round 2 copied the 307 MB matrix back to the host, repacked it in a single-threaded loop and uploaded two
// more copies: seconds per rank behind a 2 ms broadcast).  The transposed copy is NOT built here: see ensure_index_transposed.
static void build_index_aux(rvc_engine *e)
{
    if (e->d_indexT) { (void)hipFree(e->d_indexT); e->d_indexT = nullptr; }
    if (e->d_indexF) { (void)hipFree(e->d_indexF); e->d_indexF = nullptr; }
    if (e->d_ynorm) (void)hipFree(e->d_ynorm);
    if (e->d_nhn) (void)hipFree(e->d_nhn);
    e->d_ynorm = e->d_nhn = nullptr;
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    HIPCHK(hipEventRecord(a, e->stream));
    if (e->index_dim % 16 == 0) {
        const long long nt = ((long long)e->index_n + 15) / 16, nc = (long long)e->index_dim / 16, total4 = nt * nc * 64;
        HIPCHK(hipMalloc(&e->d_indexF, (size_t)total4 * 4 * sizeof(float)));
        hipLaunchKernelGGL(knn_pack_index_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, e->stream, e->d_index, (long long)e->index_n, (int)e->index_dim, e->d_indexF, total4);
    }
    HIPCHK(hipMalloc(&e->d_ynorm, e->index_n * sizeof(float)));
    HIPCHK(hipMalloc(&e->d_nhn, e->index_n * sizeof(float)));
    hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((e->index_n + 255) / 256)), dim3(256), 0, e->stream, e->d_index, (int)e->index_n, (int)e->index_dim, e->d_ynorm, e->d_nhn);
    HIPCHK(hipEventRecord(b, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
    float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, a, b));
    e->index_prep_ms = ms;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
}

// [dim][n] copy of the index, built by a device transpose the first time a plan needs it: the many-stream distance GEMM (the index is
// its activation operand) and the forced / non-MFMA exhaustive scan.  A single stream never builds it (HBM then holds the index twice:
// row-major for the exact re-rank and the blend, fragment order for the scan); its degenerate-data fallback walks the row-major copy.
static void ensure_index_transposed(rvc_engine *e)
{
    if (e->d_indexT || !e->d_index) return;
    HIPCHK(hipMalloc(&e->d_indexT, e->index_n * e->index_dim * sizeof(float)));
    dim3 grid((unsigned)((e->index_n + 31) / 32), (unsigned)((e->index_dim + 31) / 32));
    hipLaunchKernelGGL(knn_transpose_kernel, grid, dim3(256), 0, e->stream, e->d_index, (long long)e->index_n, (int)e->index_dim, e->d_indexT);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipGetLastError());
}

}  // namespace rvc

extern "C" {

rvc_status rvc_load_index(rvc_engine *e, const float *vectors, size_t n, size_t dim)
{
    return guarded(e, [&]() {
        if (n < KNN_K || dim < 1) throw ShapeError("index needs at least 4 vectors");
        HIPCHK(hipDeviceSynchronize());
        if (e->d_index && e->index_owned) (void)hipFree(e->d_index);
        HIPCHK(hipMalloc(&e->d_index, n * dim * sizeof(float)));
        e->index_owned = true;
        HIPCHK(hipMemcpy(e->d_index, vectors, n * dim * sizeof(float), hipMemcpyHostToDevice));
        e->index_n = n; e->index_dim = dim;
        build_index_aux(e);
        e->plans.clear(); e->last_plan = nullptr;
        return RVC_OK;
    });
}

rvc_status rvc_load_index_device(rvc_engine *e, const void *d_vectors, size_t n, size_t dim)
{
    return guarded(e, [&]() {
        if (n < KNN_K || dim < 1) throw ShapeError("index needs at least 4 vectors");
        HIPCHK(hipDeviceSynchronize());
        if (e->d_index && e->index_owned) (void)hipFree(e->d_index);
        HIPCHK(hipMalloc(&e->d_index, n * dim * sizeof(float)));
        e->index_owned = true;
        HIPCHK(hipMemcpy(e->d_index, d_vectors, n * dim * sizeof(float), hipMemcpyDeviceToDevice));
        e->index_n = n; e->index_dim = dim;
        build_index_aux(e);
        e->plans.clear(); e->last_plan = nullptr;
        return RVC_OK;
    });
}

void *rvc_index_device_ptr(rvc_engine *e, size_t *bytes)
{
    if (!e || !e->d_index) { if (bytes) *bytes = 0; return nullptr; }
    if (bytes) *bytes = e->index_n * e->index_dim * sizeof(float);
    return e->d_index;
}

rvc_status rvc_get_knn(rvc_engine *e, int32_t *idx, float *dist, size_t cap_rows, size_t *rows)
{
    return guarded(e, [&]() {
        Plan *pl = e->last_plan;
        if (!pl || !pl->with_index) { if (rows) *rows = 0; return RVC_OK; }
        const size_t r = pl->R;
        if (rows) *rows = r;
        if (cap_rows < r) return RVC_SHAPE;
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(idx, pl->d_knn_idx, r * KNN_K * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(dist, pl->d_knn_dist, r * KNN_K * sizeof(float), hipMemcpyDeviceToHost));
        return RVC_OK;
    });
}

rvc_status rvc_profile_last(rvc_engine *e, int *launches, double *kernel_ms, double *flops)
{
    return guarded(e, [&]() {
        Plan *pl = e->last_plan;
        if (!pl) return RVC_SHAPE;
        HIPCHK(hipDeviceSynchronize());
        double ms = 0, fl = 0;
        int nl = 0;
        for (size_t i = 0; i < pl->prof_used; i++) { if (pl->prof[i].bytes > 0) continue; float t; HIPCHK(hipEventElapsedTime(&t, pl->prof[i].a, pl->prof[i].b)); ms += t; fl += pl->prof[i].flops; nl++; }
        if (launches) *launches = nl;
        if (kernel_ms) *kernel_ms = ms;
        if (flops) *flops = pl->prof_used ? fl : pl->igemm_flops;
        return RVC_OK;
    });
}

rvc_status rvc_envelop_mixing(rvc_engine *e, const float *input, float *output, size_t output_len, size_t sample_rate, double mix_rate)
{
    return guarded(e, [&]() {
        const size_t zc = sample_rate / 100;
        if (zc == 0 || output_len < zc) throw ShapeError("envelop_mixing: output shorter than one 10 ms hop");
        const int n = (int)output_len, frame = (int)(4 * zc), hop = (int)zc;
        const int nf = (n + 2 * (frame / 2) - frame) / hop + 1;
        float *d_in, *d_out, *d_r;
        HIPCHK(hipMalloc(&d_in, output_len * 4)); HIPCHK(hipMalloc(&d_out, output_len * 4)); HIPCHK(hipMalloc(&d_r, (size_t)2 * nf * 4));
        HIPCHK(hipMemcpyAsync(d_in, input, output_len * 4, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(d_out, output, output_len * 4, hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(post_rms_kernel, dim3(nf), dim3(256), 0, e->stream, d_in, n, frame, hop, d_r, 0LL, 0LL);
        hipLaunchKernelGGL(post_rms_kernel, dim3(nf), dim3(256), 0, e->stream, d_out, n, frame, hop, d_r + nf, 0LL, 0LL);
        hipLaunchKernelGGL(post_mix_kernel, dim3((n + 255) / 256), dim3(256), 0, e->stream, d_out, n, d_r, nf, d_r + nf, nf, (float)(1.0 - mix_rate), 0LL, 0LL, (const float *)nullptr);
        HIPCHK(hipMemcpyAsync(output, d_out, output_len * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_r);
        return RVC_OK;
    });
}

rvc_status rvc_sola_step(rvc_engine *e, float *output, size_t output_len, float *sola_buffer, size_t sola_len, size_t search,
                         size_t frame, float *frame_out, size_t *sola_offset)
{
    return guarded(e, [&]() {
        if (search + 1 > 1024) throw ShapeError("sola search range too long");
        if (output_len < sola_len + search || output_len < search + frame + sola_len) throw PanicError("sola: output shorter than offset + frame + tail (the reference slices out of range)");
        float *d_out, *d_sola, *d_frame, *d_cor; int *d_off;
        HIPCHK(hipMalloc(&d_out, output_len * 4)); HIPCHK(hipMalloc(&d_sola, sola_len * 4)); HIPCHK(hipMalloc(&d_frame, frame * 4)); HIPCHK(hipMalloc(&d_off, 4));
        HIPCHK(hipMalloc(&d_cor, (search + 1) * 4));
        HIPCHK(hipMemcpyAsync(d_out, output, output_len * 4, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(d_sola, sola_buffer, sola_len * 4, hipMemcpyHostToDevice, e->stream));
        hipLaunchKernelGGL(post_sola_corr_kernel, dim3((unsigned)(search + 4) / 4), dim3(256), 0, e->stream, d_out, d_sola, (int)sola_len, (int)search, d_cor, 0LL, 0LL, 0LL);
        hipLaunchKernelGGL(post_sola_kernel, dim3(1), dim3(1024), 0, e->stream, d_out, d_sola, (int)sola_len, (int)search, (int)frame, d_frame, d_off, d_cor, 0LL, 0LL, 0LL, 0LL);
        int off = 0;
        HIPCHK(hipMemcpyAsync(output, d_out, output_len * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(sola_buffer, d_sola, sola_len * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(frame_out, d_frame, frame * 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(&off, d_off, 4, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        if (sola_offset) *sola_offset = (size_t)off;
        (void)hipFree(d_out); (void)hipFree(d_sola); (void)hipFree(d_frame); (void)hipFree(d_off); (void)hipFree(d_cor);
        return RVC_OK;
    });
}

// test hook (see "switches" at the top of this file): set (value != NULL) or clear one of the named hooks; 0 = done, -1 = unknown name.
// Hooks are read when a model is loaded (RVC_NO_LN_FUSE) or a plan is built -- set them before.
int rvc_debug_option(const char *name, const char *value)
{
    if (!name) return -1;
#ifndef RVC_TUNING
    if (!is_test_hook(name)) return -1;
#endif
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (value) g_opts[name] = value; else g_opts.erase(name);
    return 0;
}

// timeline of the last call (RVC_STAMPS=1): "name us-since-first-stamp" lines
extern "C" int rvc_debug_stamps(rvc_engine *e, char *buf, size_t cap)
{
    if (!e || !e->last_plan || !e->last_plan->d_stamps) return 0;
    Plan &pl = *e->last_plan;
    std::vector<unsigned long long> h(pl.stamp_names.size());
    if (hipMemcpy(h.data(), pl.d_stamps, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    unsigned long long t0 = ~0ull; for (auto v : h) t0 = std::min(t0, v);
    std::string out;
    for (size_t i = 0; i < h.size(); i++) { char ln[96]; snprintf(ln, sizeof ln, "%s %.2f\n", pl.stamp_names[i].c_str(), (double)(h[i] - t0) / 100.0); out += ln; }
    if (out.size() + 1 > cap) return -1;
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)h.size();
}

#ifdef RVC_TUNING
// tuning build only: time one Conv1d(Cin -> M, KW taps, dilation dil, stride 1, "same" padding) over N positions and `streams` streams,
// through whatever kernel the planner (or RVC_FORCE_CFG) picks; returns microseconds per launch
double rvc_debug_conv_bench(rvc_engine *e, int M, int Cin, int KW, int dil, int N, int iters, int pre_act, int streams, int act)
{
    double us = -1.0;
    (void)guarded(e, [&]() {
        std::vector<float> w((size_t)M * Cin * KW), bias(M, 0.1f);
        for (size_t i = 0; i < w.size(); i++) w[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
        ConvW cw = prep_conv(w.data(), bias.data(), M, Cin, KW, 1);
        const int Bb = streams > 0 ? streams : 1;
        Plan pl; pl.B = Bb;
        const int pad = (KW - 1) * dil / 2;
        T1 x = make_t1(pl.arena, Bb, Cin, N, (pad + 3) / 4 * 4), y = make_t1(pl.arena, Bb, M, N, 0);
        std::vector<float> hx((size_t)Cin * x.ld, 0.25f);
        for (int b = 0; b < Bb; b++) HIPCHK(hipMemcpy(x.p + (long long)b * x.bs - x.halo, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        ConvOpts o; if (pre_act) { o.pre_act = ACT_LRELU; o.pre_slope = 0.1f; }
        o.act = act;
        add_conv1d(pl, cw, x, y, 1, pad, dil, o);
        HIPCHK(hipDeviceSynchronize());
        for (int i = 0; i < 3; i++) for (auto &op : pl.ops.v) op(e->stream);
        hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
        HIPCHK(hipEventRecord(a, e->stream));
        for (int i = 0; i < iters; i++) for (auto &op : pl.ops.v) op(e->stream);
        HIPCHK(hipEventRecord(b, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        float ms; HIPCHK(hipEventElapsedTime(&ms, a, b));
        us = ms * 1e3 / iters;
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        free_conv(cw);
        return RVC_OK;
    });
    return us;
}
#endif

// test aid for the folded LayerNorm (IgemmP::ln_*): two launches on deterministic data against a double-precision host evaluation --
//   (1) y1 = W1 . LN(x) + b1 through the folded weights on the RAW x, publishing the column statistics;
//   (2) y2 = W2 . z + b2 + LN(x), the residual normalised on the fly from the statistics launch (1) published (needs M2 = K rows).
// Tile shape / K split are whatever the planner (or RVC_FORCE_CFG) picks; the planner forces an in-workgroup K split.  Returns the
// largest |gpu - host| / rms(host) over both outputs and the published (mean, rstd); negative on failure.  `offset` = mean of the
// tensor being normalised (spread ~1.15): large values probe the cancellation in the one-pass statistics.
double rvc_debug_ln_fold_check(rvc_engine *e, int M, int K, int N, float offset)
{
    double worst = -1.0;
    (void)guarded(e, [&]() {
        if (K % 16 != 0) throw ShapeError("K must be a multiple of 16");
        auto rnd = [](size_t i, unsigned salt) { return (float)((((i + 1) * 2654435761u) ^ (salt * 40503u) ^ (i >> 5)) % 2001) / 1000.0f - 1.0f; };
        std::vector<float> w1((size_t)M * K), b1(M), w2((size_t)K * M), b2(K), g(K), beta(K), hx((size_t)K * N), hz((size_t)M * N);
        for (size_t i = 0; i < w1.size(); i++) w1[i] = 0.5f * rnd(i, 1);
        for (size_t i = 0; i < w2.size(); i++) w2[i] = 0.5f * rnd(i, 2);
        for (int m = 0; m < M; m++) b1[m] = 0.1f * rnd(m, 3);
        for (int k = 0; k < K; k++) { b2[k] = 0.1f * rnd(k, 4); g[k] = 1.0f + 0.3f * rnd(k, 5); beta[k] = 0.2f * rnd(k, 6); }
        for (size_t i = 0; i < hx.size(); i++) hx[i] = 2.0f * rnd(i, 7) + offset;        // column mean = offset, spread ~1.15
        for (size_t i = 0; i < hz.size(); i++) hz[i] = rnd(i, 8);
        float *wsum = nullptr;
        ConvW c1 = ModelCV::fold_ln(w1.data(), b1.data(), M, K, g.data(), beta.data(), &wsum);
        ConvW c2 = prep_conv(w2.data(), b2.data(), K, M, 1, 1);
        float *dg = upload_f(g), *dbeta = upload_f(beta);
        Plan pl; pl.B = 1;
        T1 x = make_t1(pl.arena, 1, K, N, 0), y1 = make_t1(pl.arena, 1, M, N, 0), z = make_t1(pl.arena, 1, M, N, 0), y2 = make_t1(pl.arena, 1, K, N, 0);
        float *st = pl.arena.floats((size_t)2 * N + 16);
        for (int k = 0; k < K; k++) HIPCHK(hipMemcpy(x.p + (long long)k * x.ld, &hx[(size_t)k * N], (size_t)N * 4, hipMemcpyHostToDevice));
        for (int m = 0; m < M; m++) HIPCHK(hipMemcpy(z.p + (long long)m * z.ld, &hz[(size_t)m * N], (size_t)N * 4, hipMemcpyHostToDevice));
        { ConvOpts o; o.ln_wsum = wsum; o.ln_stats_out = st; o.ln_rows = K; add_conv1d(pl, c1, x, y1, 1, 0, 1, o); }
        { ConvOpts o; o.res = x.p; o.res_cs = x.ld; o.res_bs = x.bs; o.ln_stats_in = st; o.ln_g = dg; o.ln_b = dbeta; add_conv1d(pl, c2, z, y2, 1, 0, 1, o); }
        HIPCHK(hipDeviceSynchronize());
        for (auto &op : pl.ops.v) op(e->stream);
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipGetLastError());
        // host: LayerNorm over the K rows of every column (two-pass, double), then the two layers
        std::vector<double> ln((size_t)K * N), mean(N), rstd(N);
        for (int n = 0; n < N; n++) {
            double s = 0; for (int k = 0; k < K; k++) s += hx[(size_t)k * N + n];
            const double mu = s / K; double q = 0;
            for (int k = 0; k < K; k++) { const double d = hx[(size_t)k * N + n] - mu; q += d * d; }
            mean[n] = mu; rstd[n] = 1.0 / std::sqrt(q / K + 1e-5);
            for (int k = 0; k < K; k++) ln[(size_t)k * N + n] = (hx[(size_t)k * N + n] - mu) * rstd[n] * g[k] + beta[k];
        }
        double err = 0.0;
        std::vector<float> row(N), hst((size_t)2 * N);
        {
            double ss = 0; std::vector<double> ref((size_t)M * N);
            for (int m = 0; m < M; m++) for (int n = 0; n < N; n++) {
                double a = b1[m]; for (int k = 0; k < K; k++) a += (double)w1[(size_t)m * K + k] * ln[(size_t)k * N + n];
                ref[(size_t)m * N + n] = a; ss += a * a;
            }
            const double rms1 = std::sqrt(ss / ((double)M * N)) + 1e-12;
            for (int m = 0; m < M; m++) {
                HIPCHK(hipMemcpy(row.data(), y1.p + (long long)m * y1.ld, (size_t)N * 4, hipMemcpyDeviceToHost));
                for (int n = 0; n < N; n++) err = std::max(err, std::fabs((double)row[n] - ref[(size_t)m * N + n]) / rms1);
            }
        }
        {
            double ss = 0; std::vector<double> ref((size_t)K * N);
            for (int k = 0; k < K; k++) for (int n = 0; n < N; n++) {
                double a = b2[k]; for (int m = 0; m < M; m++) a += (double)w2[(size_t)k * M + m] * hz[(size_t)m * N + n];
                a += ln[(size_t)k * N + n];
                ref[(size_t)k * N + n] = a; ss += a * a;
            }
            const double rms2 = std::sqrt(ss / ((double)K * N)) + 1e-12;
            for (int k = 0; k < K; k++) {
                HIPCHK(hipMemcpy(row.data(), y2.p + (long long)k * y2.ld, (size_t)N * 4, hipMemcpyDeviceToHost));
                for (int n = 0; n < N; n++) err = std::max(err, std::fabs((double)row[n] - ref[(size_t)k * N + n]) / rms2);
            }
        }
        HIPCHK(hipMemcpy(hst.data(), st, (size_t)2 * N * 4, hipMemcpyDeviceToHost));
        for (int n = 0; n < N; n++) {
            err = std::max(err, std::fabs((double)hst[2 * n] - mean[n]) / (std::fabs(mean[n]) + 1.0));
            err = std::max(err, std::fabs((double)hst[2 * n + 1] - rstd[n]) / rstd[n]);
        }
        worst = err;
        free_conv(c1); free_conv(c2);
        (void)hipFree(wsum); (void)hipFree(dg); (void)hipFree(dbeta);
        return RVC_OK;
    });
    return worst;
}

#ifdef RVC_TUNING
// tuning build only (-DRVC_KPROBE): one launch of a Conv1d with per-wave phase stamps (device wall clock, 10 ns ticks):
// out[wave][16]; returns the number of waves (workgroups * waves per workgroup), *event_us = the dispatch's own begin..end time
int rvc_debug_conv_probe(rvc_engine *e, int M, int Cin, int KW, int dil, int N, unsigned long long *out, size_t cap_waves, double *event_us, int *waves_per_wg)
{
    int nw = -1;
    (void)guarded(e, [&]() {
        std::vector<float> w((size_t)M * Cin * KW), bias(M, 0.1f);
        for (size_t i = 0; i < w.size(); i++) w[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
        ConvW cw = prep_conv(w.data(), bias.data(), M, Cin, KW, 1);
        Plan pl; pl.B = 1;
        const int pad = (KW - 1) * dil / 2;
        T1 x = make_t1(pl.arena, 1, Cin, N, (pad + 3) / 4 * 4), y = make_t1(pl.arena, 1, M, N, 0);
        std::vector<float> hx((size_t)Cin * x.ld, 0.25f);
        HIPCHK(hipMemcpy(x.p - x.halo, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        unsigned long long *d_probe; const size_t pbytes = (size_t)1 << 24;
        HIPCHK(hipMalloc(&d_probe, pbytes)); HIPCHK(hipMemset(d_probe, 0, pbytes));
        g_kprobe = d_probe;
        ConvOpts o; if (const char *a = tune_env("RVC_BENCH_ACT")) o.act = atoi(a);
        add_conv1d(pl, cw, x, y, 1, pad, dil, o);
        g_kprobe = nullptr;
        HIPCHK(hipDeviceSynchronize());
        for (int i = 0; i < 5; i++) for (auto &op : pl.ops.v) op(e->stream);
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemset(d_probe, 0, pbytes));
        HIPCHK(hipDeviceSynchronize());
        pl.profile = true; pl.prof_used = 0;
        for (auto &op : pl.ops.v) op(e->stream);
        HIPCHK(hipDeviceSynchronize());
        float t = 0.f; HIPCHK(hipEventElapsedTime(&t, pl.prof[0].a, pl.prof[0].b));
        if (event_us) *event_us = t * 1e3;
        nw = g_last_wgs * g_last_waves;
        if (waves_per_wg) *waves_per_wg = g_last_waves;
        if ((size_t)nw <= cap_waves && (size_t)nw * 128 <= pbytes) HIPCHK(hipMemcpy(out, d_probe, (size_t)nw * 128, hipMemcpyDeviceToHost));
        else nw = -2;
        (void)hipFree(d_probe);
        free_conv(cw);
        return RVC_OK;
    });
    return nw;
}
#endif

// test aid: one Conv1d(Cin -> M, KW taps, dilation dil, "same" padding, bias, optional input LeakyReLU) over N positions and `streams`
// streams on deterministic data, through whatever tile configuration the planner (or RVC_FORCE_CFG) picks, against a double-precision
// host evaluation.  Returns the largest |gpu - host| / (rms(host) + 1e-12); negative on failure.
double rvc_debug_conv_check(rvc_engine *e, int M, int Cin, int KW, int dil, int N, int streams, int pre_act)
{
    double worst = -1.0;
    (void)guarded(e, [&]() {
        std::vector<float> w((size_t)M * Cin * KW), bias(M);
        for (size_t i = 0; i < w.size(); i++) w[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
        for (int m = 0; m < M; m++) bias[m] = 0.01f * (float)(m % 7) - 0.02f;
        ConvW cw = prep_conv(w.data(), bias.data(), M, Cin, KW, 1);
        Plan pl; pl.B = streams;
        const int pad = (KW - 1) * dil / 2, halo = (pad + 3) / 4 * 4;
        T1 x = make_t1(pl.arena, streams, Cin, N, halo), y = make_t1(pl.arena, streams, M, N, 0);
        std::vector<float> hx((size_t)streams * Cin * N);
        for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)(((i * 40503u) ^ (i >> 3)) % 2001) / 1000.0f - 1.0f;
        for (int b = 0; b < streams; b++)
            for (int c = 0; c < Cin; c++)
                HIPCHK(hipMemcpy(x.p + (long long)b * x.bs + (long long)c * x.ld, &hx[((size_t)b * Cin + c) * N], (size_t)N * 4, hipMemcpyHostToDevice));
        ConvOpts o; if (pre_act) { o.pre_act = ACT_LRELU; o.pre_slope = 0.1f; }
        add_conv1d(pl, cw, x, y, 1, pad, dil, o);
        HIPCHK(hipDeviceSynchronize());
        for (auto &op : pl.ops.v) op(e->stream);
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipGetLastError());
        std::vector<float> hy((size_t)N);
        double err = 0.0, ss = 0.0; size_t cnt = 0;
        std::vector<double> ref((size_t)N);
        for (int b = 0; b < streams; b++)
            for (int m = 0; m < M; m++) {
                HIPCHK(hipMemcpy(hy.data(), y.p + (long long)b * y.bs + (long long)m * y.ld, (size_t)N * 4, hipMemcpyDeviceToHost));
                for (int n = 0; n < N; n++) {
                    double a = bias[m];
                    for (int c = 0; c < Cin; c++)
                        for (int k = 0; k < KW; k++) {
                            const int t = n + k * dil - pad;
                            if (t < 0 || t >= N) continue;
                            double v = hx[((size_t)b * Cin + c) * N + t];
                            if (pre_act && v < 0) v *= 0.1f;
                            a += (double)w[((size_t)m * Cin + c) * KW + k] * v;
                        }
                    ref[n] = a; ss += a * a; cnt++;
                }
                for (int n = 0; n < N; n++) err = std::max(err, std::fabs((double)hy[n] - ref[n]));
            }
        worst = err / (std::sqrt(ss / (double)std::max<size_t>(cnt, 1)) + 1e-12);
        free_conv(cw);
        return RVC_OK;
    });
    return worst;
}

rvc_status rvc_profile_last_knn(rvc_engine *e, int *launches, double *kernel_ms, double *bytes)
{
    return guarded(e, [&]() {
        Plan *pl = e->last_plan;
        if (!pl) return RVC_SHAPE;
        HIPCHK(hipDeviceSynchronize());
        double ms = 0, by = 0; int nl = 0;
        for (size_t i = 0; i < pl->prof_used; i++) { if (!(pl->prof[i].bytes > 0)) continue; float t; HIPCHK(hipEventElapsedTime(&t, pl->prof[i].a, pl->prof[i].b)); ms += t; by += pl->prof[i].bytes; nl++; }
        if (launches) *launches = nl;
        if (kernel_ms) *kernel_ms = ms;
        if (bytes) *bytes = by;
        return RVC_OK;
    });
}

// test aid: launches (ops) of the last call's plan
int rvc_debug_last_plan(rvc_engine *e, int *n_ops)
{
    if (!e || !e->last_plan) return 0;
    int n = 0;
    for (size_t i = 0; i < e->last_plan->ops.v.size(); i++) if (e->last_plan->ops.kind[i] == 0) n++;
    if (n_ops) *n_ops = n;
    return 1;
}

// tuning aid: one line per profiled launch of the last call: "<us> <gflop> <description>"
int rvc_debug_profile_dump(rvc_engine *e, char *buf, size_t cap)
{
    if (!e || !e->last_plan) return 0;
    Plan &pl = *e->last_plan;
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    std::string out;
    for (size_t i = 0; i < pl.prof_used; i++) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, pl.prof[i].a, pl.prof[i].b) != hipSuccess) continue;
        char ln[320];
        snprintf(ln, sizeof ln, "%.2f %.4f %s\n", t * 1e3, pl.prof[i].flops * 1e-9, pl.prof[i].desc >= 0 ? pl.descs[pl.prof[i].desc].c_str() : (pl.prof[i].bytes > 0 ? "knn_dot" : "?"));
        out += ln;
    }
    if (out.size() + 1 > cap) return -1;
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)pl.prof_used;
}

rvc_status rvc_get_tap(rvc_engine *e, const char *name, float *out, size_t cap, size_t *n)
{
    return guarded(e, [&]() {
        Plan *pl = e->last_plan;
        if (!pl) return RVC_SHAPE;
        HIPCHK(hipDeviceSynchronize());
        for (auto &t : pl->taps) {
            if (t.name != name) continue;
            if (t.rank == 1) {
                size_t need = (size_t)t.t1.C * t.t1.T; if (n) *n = need;
                if (cap < need) return RVC_SHAPE;
                HIPCHK(hipMemcpy2D(out, (size_t)t.t1.T * 4, t.t1.p, (size_t)t.t1.ld * 4, (size_t)t.t1.T * 4, t.t1.C, hipMemcpyDeviceToHost));
            } else {
                size_t need = (size_t)t.t2.C * t.t2.H * t.t2.W; if (n) *n = need;
                if (cap < need) return RVC_SHAPE;
                for (int c = 0; c < t.t2.C; c++)
                    HIPCHK(hipMemcpy2D(out + (size_t)c * t.t2.H * t.t2.W, (size_t)t.t2.W * 4, t.t2.p + (size_t)c * t.t2.cs, (size_t)t.t2.ld * 4, (size_t)t.t2.W * 4, t.t2.H, hipMemcpyDeviceToHost));
            }
            return RVC_OK;
        }
        if (!strcmp(name, "f0") && pl->d_f0) {
            if (n) *n = (size_t)pl->Tm;
            if (cap < (size_t)pl->Tm) return RVC_SHAPE;
            HIPCHK(hipMemcpy(out, pl->d_f0, (size_t)pl->Tm * 4, hipMemcpyDeviceToHost));
            return RVC_OK;
        }
        e->err = std::string("unknown tap ") + name;
        return RVC_SHAPE;
    });
}

}  // extern "C"

#include "resample.hip.h"
#include "session.hip.h"
#include "rccl_bcast.hip.h"

// single-translation-unit build (tuning tools: `hipcc -DRVC_UNITY engine.hip`): pull the instantiation units in
#ifdef RVC_UNITY
#define RVC_IGEMM2_CFG 0
#include "igemm2_inst.hip"
#undef RVC_IGEMM2_CFG
#undef RVC_FN
#undef RVC_MF
#undef RVC_NF
#undef RVC_D
#define RVC_IGEMM2_CFG 1
#include "igemm2_inst.hip"
#undef RVC_IGEMM2_CFG
#undef RVC_FN
#undef RVC_MF
#undef RVC_NF
#undef RVC_D
#define RVC_IGEMM2_CFG 2
#include "igemm2_inst.hip"
#undef RVC_IGEMM2_CFG
#undef RVC_FN
#undef RVC_MF
#undef RVC_NF
#undef RVC_D
#define RVC_IGEMM2_CFG 3
#include "igemm2_inst.hip"
#undef RVC_IGEMM2_CFG
#undef RVC_FN
#undef RVC_MF
#undef RVC_NF
#undef RVC_D
#define RVC_IGEMM2_CFG 4
#include "igemm2_inst.hip"
#include "igemm_tiled_inst.hip"
#include "conv_tile_inst.hip"
#endif
