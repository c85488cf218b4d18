// engine_int.h -- internal declarations shared by the translation units of librvc_mi355x.so:
//   plan.hip         the planner: device memory, prepared convolution weights, the op list, implicit-GEMM tile selection, test-hook table
//   model_cv.hip     ContentVec (build_contentvec)          model_rmvpe.hip   RMVPE + decode (build_rmvpe, build_pitch_post)
//   model_synth.hip  the synthesizer (build_synth, ...)      retrieval.hip     flat-L2 index: load, device-side layouts, the plan's search section
//   engine.hip       the engine object, plans, the C ABI (+ session.hip.h, resample.hip.h, rccl_bcast.hip.h)
// The model structs (weights as prepared at load) are defined here with their loaders inline; kernels are `static` / templates in
// kernels.hip.h, so every unit emits only the kernels it launches.
#pragma once
#include "../../include/rvc_mi355x.h"
#include "blob.h"
#include "kernels.hip.h"
#include "igemm_launch.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>
#include <thread>

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
const char *test_opt(const char *name);
extern std::atomic<unsigned> g_opt_gen;      // generation of the test-hook table (plan.hip)
extern std::atomic<int> g_knn_test_lose;     // RVC_KNN_LOSE_TICKET (retrieval.hip)
int test_opt_int(const char *name, int dflt);
#ifdef RVC_TUNING
const char *tune_env(const char *name);
#else
static inline const char *tune_env(const char *) { return nullptr; }
#endif

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
            size_t sz = std::max(bytes, chunk_min_);
            void *c;
            HIPCHK(hipMalloc(&c, sz));
            HIPCHK(hipMemset(c, 0, sz));
            if (chunks_.empty()) first_size_ = sz;
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
    void set_chunk_min(size_t b) { chunk_min_ = b; }          // (trial plans of the autotuner hold one layer's tables: small chunks)
    // start over in the first chunk, keeping the memory (the autotuner's scratch plan: hundreds of trials per plan build, no hipMalloc / hipFree per trial);
    // later chunks are returned.  The memory is NOT zeroed again: only for arenas that hold tables, never activation tensors with halos.
    void rewind()
    {
        while (chunks_.size() > 1) { (void)hipFree(chunks_.back()); chunks_.pop_back(); }
        if (!chunks_.empty()) { cur_ = (char *)chunks_[0]; left_ = first_size_; total_ = first_size_; }
    }

private:
    size_t chunk_min_ = (size_t)64 << 20, first_size_ = 0;
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

static inline T1 make_t1(Arena &a, int B, int C, int T, int halo)
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
static inline T2 make_t2(Arena &a, int B, int C, int H, int W)
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

static inline int round16(int k) { return (k + 15) / 16 * 16; }


void *wmalloc(size_t bytes); void wfree(void *p);      // weight memory: slab-allocated per (device, class) (plan.hip)
void *wmalloc_plan(size_t bytes);                      // ... class 1: plan-lifetime copies, in slabs of their own
void wslab_info(int dev, int *count, size_t *bytes);
float *upload_f(const std::vector<float> &v, int cls = 0);
float *upload_f(const float *p, size_t n);
float *upload_fragments(const std::vector<float> &panel, int nphase, int M, int Kp, int cls = 0);
static inline long long phase_stride(const ConvW &c) { return (long long)((c.M + 15) / 16 * 16) * c.Kp; }
ConvW prep_conv(const float *w, const float *bias, int Cout, int Cin, int KW, int groups);
ConvW prep_convT1d(const float *w, const float *bias, int Cin, int Cout, int K, int S);
ConvW prep_convT2d(const float *w, const float *bias, int Cin, int Cout);
void free_conv(ConvW &c);
// conv32s_kernel's K order of a phase's fragment panel ((channel block, tap, channel group)-major): built ONCE per source panel by a device kernel and
// shared by every plan (and every trial of the autotuner); forgotten when the source conv is freed (plan.hip)
const float *c32s_panel(const float *src_frag, int M, int nchunks, int cin, int KW);
void merge_convs(const std::vector<ConvW *> &cs);

// ---------------------------------------------------------------------------------------
// op list ("plan") construction
// ---------------------------------------------------------------------------------------
// LayerNorm folded into the neighbouring GEMMs (ModelCV::fold_ln): one stream only.  The folded form works on streams folded into N as well
// (parity-tested in round 4), but the LayerNorm-consumer GEMM (two waves per SIMD, statistics in the operand stream) loses more on the wider
// N than the 31 launches save: 2 / 4 / 8 streams measured 3.33 / 5.09 / 8.30 ms with the fold against 3.30 / 4.83 / 7.85 without.
constexpr int LN_FOLD_MAX_STREAMS = 1;

struct ConvOpts {
    int act = ACT_NONE; float slope = 0.f; float scale = 1.f; bool accumulate = false;
    int pre_act = ACT_NONE; float pre_slope = 0.f;
    const float *res = nullptr; int res_cs = 0; long long res_bs = 0; int res_rs = 0;
    int m_off = 0, m_cnt = -1;   // output-row sub-range of the weight panel
    bool no_bias = false;
    bool glu = false;            // GLU-packed weight rows, gate fused into the epilogue (ModelSY flows)
    bool bf3 = false;            // exploratory: this 1x1 layer may run its products as three bf16 MFMAs (rvc_set_gemm_precision; igemm_bf3_kernel)
    int y_ws = 0;                // add_conv2d: output column stride in floats (0 = 1); with y.ld = 1 the layer writes its image transposed (RMVPE's head -> GRU input layout)
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
    // gather tables by content (round 6): the 3x3 layers of one RMVPE level have identical k -> offset tables; with one device copy per distinct table the second
    // and later layers of a level find it in the L2 instead of fetching a cold 6 KB table from HBM in front of their first operand gather
    std::map<std::vector<int>, const int *> koff_tabs;
    bool rm_fuse = false;         // RMVPE's shallow ConvBlockRes as one launch each (model_rmvpe.hip; decided per plan from the engine's f0 partition)
    bool bf3 = false;             // built under rvc_set_gemm_precision(e, 1): ContentVec's 1x1 GEMMs on the split-bf16 kernel (exploratory)
    bool autotune = false;        // rvc_set_plan_autotune: layers with several eligible kernels / tiles are chosen by timing them at plan build (plan.hip queue_igemm)
    double tune_ms = 0; int tuned_layers = 0, tune_changed = 0, tune_hits = 0;      // time spent in trials, layers tuned here / changed against the rules / taken from the process cache
    bool plain_plan = false;      // taps level 1: the explicit plan (LayerNorm launches, WaveNets layer by layer); level 2 taps the production plan
    int mode = 0;   // 0 infer, 1 hubert only, 2 pitch only
    // I/O tensors
    float *d_in = nullptr;  // [B][L]
    T1 cv_out, audio;
    float *d_f0 = nullptr;  // [B][Tm]
    float *d_feat = nullptr; // extract_feature output (1,2T+1,C)
    int *d_knn_idx = nullptr; float *d_knn_dist = nullptr;
    // one-launch retrieval: its ticket counters, and what runs instead when a selector gave up (engine.hip recover_retrieval)
    unsigned *knn_ticket = nullptr; size_t knn_ticket_bytes = 0;
    std::vector<Op> knn_fallback;
    size_t op_phone = 0, op_ret_begin = 0, op_ret_end = 0;      // ops [op_phone, op_ret_begin): phone gather; [op_ret_begin, op_ret_end): the retrieval section
    unsigned opt_gen = 0;         // test-hook generation the plan was built under
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
    bool final_out_honoured = false;     // the chunk's last convolution was queued on a launch path that writes Plan::cur_out (queue_igemm reports it)
    // graph
    hipGraphExec_t graph_exec = nullptr;
    ~Plan()
    {
        for (float *p : owned_dev) wfree(p);
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (ev_done) (void)hipEventDestroy(ev_done);
        for (auto &e : prof) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    }
};


// ---- planner entry points (plan.hip) ----
extern unsigned long long *g_kprobe;      // tuning build (-DRVC_KPROBE): destination of the per-wave phase stamps
extern int g_last_waves, g_last_wgs, g_ncu;
extern char g_last_kernel[16];          // family of the most recently queued implicit-GEMM launch ("reg", "g32", "c32s", ...)
void queue_igemm(Plan &pl, IgemmP p, int B, const std::vector<int> &koff, const std::vector<PhaseD> &phases, bool final_out = false);
void fill_epilogue(IgemmP &p, const ConvW &cw, const ConvOpts &o);
void add_conv1d(Plan &pl, const ConvW &cw, const T1 &x, const T1 &y, int stride, int pad, int dil, ConvOpts o = ConvOpts());
void add_conv1d_multi(Plan &pl, const std::vector<const ConvW *> &cws, const T1 &x, bool x_grouped, const T1 &y,
                      const std::vector<int> &pads, const std::vector<int> &dils, ConvOpts o = ConvOpts(), bool res_grouped = true);
void add_conv1d_two(Plan &pl, const ConvW &c0, const ConvW &c1, const float *pair_bias, const T1 &x, const T1 &y0, const T1 &y1);
void add_convT1d(Plan &pl, const ConvW &cw, const T1 &x, const T1 &y, int pad, ConvOpts o = ConvOpts());
void add_conv2d(Plan &pl, const ConvW &cw, const T2 &x, const T2 &y, ConvOpts o = ConvOpts());
void add_convT2d(Plan &pl, const ConvW &cw, const T2 &x, const T2 &y, ConvOpts o = ConvOpts());
void add_layernorm(Plan &pl, const T1 &x, const float *g, const float *b);
void add_stamp(Plan &pl, const char *name);
void add_tap(Plan &pl, const char *name, const T1 &t);
void add_tap2(Plan &pl, const char *name, const T2 &t);
void plan_kernel_attrs();                 // per-device function attributes of the kernels each unit launches
void cv_kernel_attrs(); void rmvpe_kernel_attrs(); void synth_kernel_attrs(); void retrieval_kernel_attrs();

// ---------------------------------------------------------------------------------------
// models
// ---------------------------------------------------------------------------------------
struct DevVec { float *p = nullptr; };
static inline float *dv(const Blob &b, const std::string &name) { const BlobTensor &t = b.t(name); return upload_f(t.data, t.nelem); }

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
        if (proj_wsum) wfree(proj_wsum);
        for (auto &L : layers) {
            free_conv(L.qkv); free_conv(L.o); free_conv(L.ff1); free_conv(L.ff2); free_conv(L.qkv_f); free_conv(L.ff1_f);
            if (L.qkv_wsum) wfree(L.qkv_wsum);
            if (L.ff1_wsum) wfree(L.ff1_wsum);
        }
        for (float *p : owned) wfree(p);
    }
    int out_frames(size_t L) const
    {
        long long T = (long long)L;
        for (int i = 0; i < 7; i++) { if (T < conv_k[i]) return 0; T = (T - conv_k[i]) / conv_s[i] + 1; }
        return (int)T;
    }
};

struct ResBlockW {
    ConvW c1, c2, sc; bool has_sc = false; int ci = 0, co = 0; float *pair_bias = nullptr;      // pair_bias: [c1.bias; sc.bias] for the fused c1 + shortcut launch
    // rm_block_kernel (rmblock.hip.h; blocks with 16 / 32 / 64 output channels): the three weight panels in its fragment order
    // [tap][Cin16 / 4][Cout / 16][64 lanes] (lane (li, kq) = W[16 mt + li][4 c4 + kq][tap], zero beyond Cin), packed once at model load
    // (one allocation [w1 | w2 | wsc]: the kernel touches every 128-byte line of it once at its start -- the panels are cold in HBM at every chunk)
    float *f_w1 = nullptr, *f_w2 = nullptr, *f_sc = nullptr; int f_lines = 0;
};
// -> fragment-order panel of a [co][ci * taps] convolution weight for rm_block_kernel
static inline std::vector<float> rm_block_panel(const float *w, int co, int ci, int taps)
{
    const int c16 = (ci + 15) / 16 * 16, steps = c16 / 4, MT = co / 16;
    std::vector<float> pk((size_t)taps * steps * MT * 64, 0.f);
    for (int t = 0; t < taps; t++)
        for (int c4 = 0; c4 < steps; c4++)
            for (int mt = 0; mt < MT; mt++)
                for (int l = 0; l < 64; l++) {
                    const int m = mt * 16 + (l & 15), c = c4 * 4 + (l >> 4);
                    if (c < ci) pk[(((size_t)t * steps + c4) * MT + mt) * 64 + l] = w[(size_t)m * ci * taps + (size_t)c * taps + t];
                }
    return pk;
}
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
        if ((co == 16 || co == 32) && ci <= 64) {
            std::vector<float> all = rm_block_panel(b.w(pre + "c1.w"), co, ci, 9);
            const size_t o2 = all.size();
            { std::vector<float> t = rm_block_panel(b.w(pre + "c2.w"), co, co, 9); all.insert(all.end(), t.begin(), t.end()); }
            const size_t o3 = all.size();
            if (r.has_sc) { std::vector<float> t = rm_block_panel(b.w(pre + "sc.w"), co, ci, 1); all.insert(all.end(), t.begin(), t.end()); }
            all.resize((all.size() + 31) / 32 * 32, 0.f);
            r.f_w1 = upload_f(all); r.f_w2 = r.f_w1 + o2; r.f_sc = r.has_sc ? r.f_w1 + o3 : nullptr; r.f_lines = (int)(all.size() / 32);
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
        auto fb = [](std::vector<std::vector<ResBlockW>> &vv) { for (auto &v : vv) for (auto &r : v) { free_conv(r.c1); free_conv(r.c2); free_conv(r.sc); if (r.pair_bias) wfree(r.pair_bias); if (r.f_w1) wfree(r.f_w1); } };
        fb(enc); fb(inter); fb(dec);
        for (auto &u : up) free_conv(u);
        free_conv(cnn); free_conv(gru_ih); free_conv(fc);
        if (whhT) wfree(whhT);
        if (bhh) wfree(bhh);
        if (whh) wfree(whh);
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
        for (auto &L : layers) { free_conv(L.qkv); free_conv(L.o); free_conv(L.ff1); free_conv(L.ff2); free_conv(L.qkv_f); if (L.qkv_wsum) wfree(L.qkv_wsum); }
        free_conv(proj_f); if (proj_wsum) wfree(proj_wsum);
        for (auto &F : flows) { free_conv(F.pre); free_conv(F.post); for (auto &c : F.in) free_conv(c); for (auto &c : F.rs) free_conv(c); if (composed) { free_conv(F.pre1); free_conv(F.postc); free_conv(F.posth); for (auto &c : F.inc) free_conv(c); } }
        for (auto &c : ups) free_conv(c);
        for (auto &c : ncs) free_conv(c);
        for (auto &s : rbs) for (auto &ch : s) for (auto &pr : ch) { free_conv(pr.first); free_conv(pr.second); }
        for (float *p : owned) wfree(p);
    }
    int upp() const { int u = 1; for (int i = 0; i < n_ups; i++) u *= up_rate[i]; return u; }
};

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
    int cv_cus = 256;                                 // CUs the ContentVec branch (and the retrieval behind it) runs on: all, or all minus the f0 partition
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
    int gemm_precision = 0;           // 0 = fp32 everywhere (the product), 1 = split-bf16 for ContentVec's 1x1 GEMMs at many streams (exploratory)
    int plan_cap = 8;                 // rvc_set_plan_cache
    int autotune = 1;                 // rvc_set_plan_autotune: 1 = plans of more than 4 streams pick among eligible kernels by timing them at plan build, 0 = the rules only
    double last_build_ms = 0;         // wall time of the last plan build (rvc_plan_autotune_info)
    int last_tuned = 0, last_changed = 0, last_hits = 0; double last_tune_ms = 0;
    long long knn_recoveries = 0;     // chunks whose retrieval was recomputed after a hand-off time-out (rvc_retrieval_info)
    long long plan_builds = 0;        // plans built since rvc_create (a miss = arena allocation + composed weights + a device synchronisation)
    int taps_on = 0;               // 0 off, 1 taps on the explicit plan, 2 taps on the production plan (rvc_enable_taps)
    bool profile_on = false, use_graph = false;
    // offline throughput mode: consecutive unsynchronised infer_device calls overlap chunk i+1's two front branches with chunk i's
    // synthesizer (two plan slots; the branch streams are ordered by events instead of forking from the main stream)
    bool pipeline = false, pipe_now = false; int pipe_slot = 0; hipEvent_t ev_in = nullptr; const void *pipe_input = nullptr; size_t pipe_input_bytes = 0;
    std::vector<float> pushed_up; uint32_t pushed_seed = 0; bool pushed_valid = false;      // what the device holds: per-stream multipliers, seed
    float *h_up = nullptr; unsigned up_slot = 0;        // pinned ring of 8 blocks of 4096 per-stream multipliers (async strided copies read them later)
    hipEvent_t ev_up[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; bool ev_up_used[8] = {false, false, false, false, false, false, false, false};      // completion of the copy that last read a block
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    size_t last_knn_rows = 0;
    int *h_status = nullptr;        // pinned, one word per stream (up to 4096)
    bool status_queued = false;     // an async copy of the status words is already in the stream in front of the caller's sync
};


namespace rvc {
static inline void set_device(rvc_engine *e) { HIPCHK(hipSetDevice(e->device)); }
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
// ---- model builders (model_*.hip) and the retrieval unit (retrieval.hip) ----
T1 build_contentvec(rvc_engine *e, Plan &pl, int B, size_t L);
T1 build_rmvpe(rvc_engine *e, Plan &pl, int B, size_t L, size_t frame16k, bool update_cache);
void build_pitch_post(rvc_engine *e, Plan &pl, int B, const T1 &sal, bool update_cache, size_t frame16k, size_t hubert_length, float **pitchf_out, int **pitch_out);
T1 build_nsf_source(rvc_engine *e, Plan &pl, int B, float *d_pitchf);
std::vector<T1> build_noise_convs(rvc_engine *e, Plan &pl, int B, const T1 &src);
void build_synth(rvc_engine *e, Plan &pl, int B, const T1 &phone, const T1 &src, float *d_pitchf, int *d_pitch, int src_join_sid, const std::vector<T1> *nz = nullptr);
void build_retrieval(rvc_engine *e, Plan &pl, int B, int T, int C, uint32_t skip_head, uint32_t R, const T1 &phone);
void build_index_aux(rvc_engine *e);
void ensure_index_transposed(rvc_engine *e);
}  // namespace rvc
