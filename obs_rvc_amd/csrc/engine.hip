// engine.hip -- the engine object and the C ABI of the MI355X-native RVC streaming inference engine.
//
// Mirrors rvc::RvcInfer (reference: rvc/src/rvc.rs:18-220): model handles, the 1024-entry
// pitch cache, and the per-chunk pipeline hubert -> (retrieval) -> pitch -> synthesizer.
// All compute is launched as hand-written gfx950 kernels (kernels.hip.h); nothing here falls
// back to a CPU path: without a HIP device every entry point returns RVC_BACKEND.
#include "engine_int.h"
#include <chrono>

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

static void init_kernel_attrs()
{
    // per device: function attributes belong to the device that is current when they are set (an engine on a second GPU of one process needs its own)
    static std::mutex mu; static bool done[64] = {};
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64) { if (done[dev]) return; done[dev] = true; }
    plan_kernel_attrs(); cv_kernel_attrs(); rmvpe_kernel_attrs(); synth_kernel_attrs(); retrieval_kernel_attrs();
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
// kernels) gets 1/8 of the CUs (one XCD), ContentVec the rest.  Sharing CUs slows the f0 branch by ~35 % (measured).  With
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
struct StreamSet { int device; hipStream_t main, plain[3], f0, cv; bool masked_ok, in_use; int nf0; };      // nf0: CUs of the masked f0 stream (sets differ only in this)
static std::mutex g_pool_mu;
static std::vector<StreamSet *> g_pool;

static StreamSet *acquire_stream_set(int device, int ncu, int nf0, bool want_masks)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (StreamSet *m : g_pool)
        if (m->device == device && !m->in_use && m->nf0 == nf0) { m->in_use = true; return m; }
    StreamSet *m = new StreamSet{device, nullptr, {nullptr, nullptr, nullptr}, nullptr, nullptr, false, true, nf0};
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
    // The f0 branch gets ONE whole XCD (an eighth of the CUs), ContentVec the other seven.  Round 5 tried two XCDs after ContentVec's GEMMs had become
    // faster.  The tuning build (per-wave probe stamps compiled in) said 2.253 -> 2.167 ms at one stream and 4.99 -> 4.61 at four; the PRODUCT build on
    // the same pool says the opposite: 2.044-2.048 -> 2.06-2.08 at one stream, 2.87 -> 3.03 at two, 4.21 -> 4.46 at four (only the v1 model, whose
    // ContentVec stops at layer 9, gains: 2.075 -> 1.994).  Partitions that cut an XCD in two lose badly either way (56: 2.244, 72: 2.469, 80: 2.484).
    // Lesson kept here: partition sizes are tuned on the product build only.
    // The one exception is measured on the product build too (tests/tools/f0_xcds.py, same box, two alternating runs): the v1 model (ContentVec stops at
    // layer 9: the chunk waits for the f0 branch) with two XCDs for f0 runs 1.98-2.02 -> 1.92-1.93 ms at one stream, 2.72-2.77 -> 2.68 at two,
    // 3.45 -> 3.43 at three (and 3.87 -> 3.95 at four: one XCD there; v2 loses at every count: 1.98-2.02 -> 2.04-2.06, 2.77 -> 2.96, 3.65 -> 3.79, 4.09 -> 4.36).
    // Sets with that partition are separate members of the pool (masks are fixed at creation).
    int ncu = prop.multiProcessorCount;
    int xcds = e->cv && e->cv->run_layers <= 9 && e->n_streams <= 3 ? 2 : 1;
    if (const char *f = test_opt("RVC_F0_XCDS")) { const int v = atoi(f); if (v >= 1 && v <= 4) xcds = v; }     // test hook: both partitions are covered by the parity tests
    int nf0 = xcds * (ncu / 8);
    g_ncu = ncu > 0 ? ncu : 256;
    if (const char *f = tune_env("RVC_F0_CUS")) { const int v = atoi(f); if (v >= 8 && v < ncu) nf0 = v; }   // tuning aid
    if (e->sset && e->sset->nf0 != nf0 && e->sset->masked_ok) {      // (callers have synchronised the device and dropped the plans)
        release_stream_set(e->sset); e->sset = nullptr;
    }
    if (!e->sset) {
        e->sset = acquire_stream_set(e->device, ncu, nf0, e->partition_ok && ncu >= 64 && ncu <= 1024);
        e->stream = e->sset->main;
        if (!e->sset->masked_ok) e->partition_ok = false;
    }
    const bool want = e->partition_ok && e->sset->masked_ok && e->n_streams <= 4;
    e->partitioned = want;
    e->cv_cus = want ? g_ncu - nf0 : g_ncu;
    // Without the partition (more than 4 streams) ContentVec runs on the main stream and the f0 branch on the SIDE stream: the side work (NSF source,
    // noise convolutions) is forked behind the join of the f0 branch, so the two never need the stream at the same time, and an engine needs no
    // stream beyond the set's four.  (Round 3 created two more plain streams for such engines.  A one-stream engine that runs late in a long
    // process -- the bench's last leg -- sometimes has its f0 branch at 1 150 us instead of 1 085 (chunk 2.06 -> 2.15-2.25 ms); with two streams
    // fewer in the process that happened in 3 of 6 bench runs instead of 4 of 6: not the cause, or not the only one.  DESIGN.md section 7.)
    e->aux[0] = want ? e->sset->f0 : e->sset->plain[1];
    e->aux[1] = e->sset->plain[1];
    e->aux[2] = want ? e->sset->cv : e->sset->plain[1];
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

// ------------------------------- plan -------------------------------------------------
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
    // plans built before a test hook changed are stale (the hooks are process-global and not part of the key)
    const unsigned gen = g_opt_gen.load();
    for (size_t i = 0; i < e->plans.size();)
        if (e->plans[i]->opt_gen != gen) {
            HIPCHK(hipDeviceSynchronize());
            if (e->last_plan == e->plans[i].get()) e->last_plan = nullptr;
            e->plans.erase(e->plans.begin() + i);
        } else i++;
    for (auto &p : e->plans)
        if (p->mode == mode && p->L == L && p->frame16k == frame16k && p->skip_head == skip_head && p->R == R && p->B == B &&
            p->with_index == with_index && p->bf3 == (e->gemm_precision == 1) && p->with_taps == (e->taps_on != 0) && p->plain_plan == (e->taps_on == 1) && p->slot == slot && p->bucket == (bucket_B > 0)) {
            // least recently used first: a hit moves to the back, so eviction (front) never takes a plan the current call has just fetched
            Plan *hit = p.get();
            std::rotate(&p, &p + 1, e->plans.data() + e->plans.size());
            return hit;
        }
    e->plan_builds++;
    const auto build_t0 = std::chrono::steady_clock::now();
    std::unique_ptr<Plan> up(new Plan());
    Plan &pl = *up;
    pl.autotune = e->autotune != 0 && B > 4;          // (queue_igemm: trials on this device while the plan is built; plans of <= 4 streams keep the latency-tuned rules)
    pl.mode = mode; pl.L = L; pl.frame16k = frame16k; pl.skip_head = skip_head; pl.R = R; pl.B = B; pl.with_index = with_index; pl.with_taps = e->taps_on != 0; pl.plain_plan = e->taps_on == 1; pl.bucket = bucket_B > 0;
    pl.slot = slot; pl.opt_gen = gen; pl.bf3 = e->gemm_precision == 1;
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
        pl.op_phone = pl.ops.v.size();
        {
            T1 cvo = pl.cv_out; dim3 grid((C * (int)R + 255) / 256, B);
            pl.ops.push_back([=](hipStream_t s) {
                hipLaunchKernelGGL(gather_phone_kernel, grid, dim3(256), 0, s, cvo.p, cvo.ld, cvo.bs, C, T, (int)skip_head, (int)R, phone.p, phone.ld, phone.bs);
            });
        }
        pl.op_ret_begin = pl.ops.v.size();
        if (with_index) build_retrieval(e, pl, B, T, C, skip_head, R, phone);
        pl.op_ret_end = pl.ops.v.size();
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
    e->last_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - build_t0).count();
    e->last_tuned = pl.tuned_layers; e->last_changed = pl.tune_changed; e->last_hits = pl.tune_hits; e->last_tune_ms = pl.tune_ms;
    // bounded plan cache (each geometry owns its activation arena and graph; rvc_set_plan_cache): evict the least recently used
    while ((int)e->plans.size() >= e->plan_cap) { if (e->last_plan == e->plans.front().get()) e->last_plan = nullptr; e->plans.erase(e->plans.begin()); }
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
    // a block is rewritten only after the strided copy that last read it has completed (nine unsynchronised calls with changing shifts
    // would otherwise overwrite a block whose copy is still pending: ADVICE r3)
    const unsigned us = e->up_slot++ & 7u;
    if (e->ev_up_used[us]) HIPCHK(hipEventSynchronize(e->ev_up[us]));
    float *hu = e->h_up + (size_t)us * 4096;
    e->pushed_up.resize(B);
    for (int b = 0; b < B; b++) hu[b] = e->pushed_up[b] = uppower(shifts ? shifts[b] : pitch_shift);
    HIPCHK(hipMemcpy2DAsync((char *)e->d_state + offsetof(StreamState, uppower), sizeof(StreamState), hu, sizeof(float), sizeof(float), (size_t)B, hipMemcpyHostToDevice, e->stream));
    if (!e->ev_up[us]) HIPCHK(hipEventCreateWithFlags(&e->ev_up[us], hipEventDisableTiming));
    HIPCHK(hipEventRecord(e->ev_up[us], e->stream)); e->ev_up_used[us] = true;
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

// -> RVC_OK / RVC_PANIC / RVC_BACKEND, or RVC_KNN_RETRY (internal): only the one-launch retrieval reported (ST_KNN_TIMEOUT) -- its counters are
// re-armed here and the caller may recompute the chunk from the retrieval on (recover_retrieval).  Every stream's word is looked at before anything
// is decided: a lost GRU hand-off on ANY stream means the plans are rebuilt, whatever a lower-numbered stream reported.
static const rvc_status RVC_KNN_RETRY = (rvc_status)100;
static rvc_status check_status(rvc_engine *e)
{
    if (!e->status_queued) { queue_status(e); HIPCHK(hipStreamSynchronize(e->stream)); }
    e->status_queued = false;
    int any = 0;
    for (int b = 0; b < e->n_streams; b++) any |= e->h_status[b];
    if (!any) return RVC_OK;
    const int zero = 0;
    for (int c = 0; c < e->n_streams; c++)
        if (e->h_status[c] != 0) { HIPCHK(hipMemcpy((char *)(e->d_state + c) + offsetof(StreamState, status), &zero, sizeof(int), hipMemcpyHostToDevice)); e->h_status[c] = 0; }
    if (any & ST_HANDOFF) {
        // a GRU step that gave up waiting left its hand-off granules behind: the plans are rebuilt before the next call
        HIPCHK(hipDeviceSynchronize());
        e->plans.clear(); e->last_plan = nullptr;
        e->err = "a cross-workgroup hand-off timed out (GRU recurrence)";
        return RVC_BACKEND;
    }
    if (any & ST_KNN_TIMEOUT) {
        // the retrieval's counters were poisoned by the selector that gave up (kernels.hip.h): re-arm them, whatever else happens
        HIPCHK(hipDeviceSynchronize());
        for (auto &p : e->plans) if (p->knn_ticket) HIPCHK(hipMemset(p->knn_ticket, 0, p->knn_ticket_bytes));
    }
    if (any & ST_PANIC) {
        e->err = "to_local_average_cents: index out of bounds (argmax bin >= 348), the reference panics here";
        return RVC_PANIC;
    }
    e->err = "the one-launch retrieval's hand-off timed out";
    return RVC_KNN_RETRY;
}

// A selector of knn_scan_select_kernel gave up (a workgroup of the launch did not arrive in time -- e.g. a GPU shared with another process): some
// queries of the chunk were not blended.  Nothing is lost: the inputs of the retrieval (ContentVec output, pitch) are still in the plan's arena.
// Rewind the chunk counters, then run [phone gather] + [exhaustive exact scan, merge, blend: the definition] + [everything behind the retrieval]
// in list order on the main stream.  Same hits (the one-launch form is tested bit-exact against this scan), same PCM.
static rvc_status final_status(rvc_engine *e)      // callers that cannot recompute the chunk: the internal retry code becomes an error
{
    const rvc_status st = check_status(e);
    return st == RVC_KNN_RETRY ? RVC_BACKEND : st;
}
static rvc_status recover_retrieval(rvc_engine *e, Plan &pl)
{
    if (pl.knn_fallback.empty() || pl.op_ret_end <= pl.op_ret_begin) { e->err = "the one-launch retrieval's hand-off timed out (no fallback on this plan)"; return RVC_BACKEND; }
    hipLaunchKernelGGL(rewind_chunk_kernel, dim3((pl.B + 63) / 64), dim3(64), 0, e->stream, e->d_state, pl.B);
    const size_t n = pl.ops.v.size();
    for (size_t i = pl.op_phone; i < pl.op_ret_begin; i++) if (pl.ops.kind[i] == 0) pl.ops.v[i](e->stream);
    for (auto &op : pl.knn_fallback) op(e->stream);
    for (size_t i = pl.op_ret_end; i < n; i++) if (pl.ops.kind[i] == 0) pl.ops.v[i](e->stream);
    HIPCHK(hipGetLastError());
    e->knn_recoveries++;
    return RVC_OK;
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

// rvc_version(): version.cpp; rvc_debug_option: plan.hip; the index entry points: retrieval.hip

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
    wfree(e->d_window); wfree(e->d_twiddle); wfree(e->d_basis);      // (upload_f: slab memory)
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
    for (int i = 0; i < 8; i++) if (e->ev_up[i]) (void)hipEventDestroy(e->ev_up[i]);
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
        configure_aux_streams(e);       // (the f0 partition depends on the ContentVec depth)
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
        return final_status(e);
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
    // (folded streams address the output with 32-bit offsets stream * cap: a caller-chosen cap beyond that range keeps the copy)
    const bool direct_out = out_on_device && !e->use_graph && pl->out_direct_ok && !no_direct && (long long)pl->B * (long long)cap < (1LL << 29);
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
    rvc_status st = check_status(e);
    if (st == RVC_KNN_RETRY) {
        // degrade, don't fail: the chunk's retrieval through the exhaustive launches, the rest of the chunk again, the output copy again
        st = recover_retrieval(e, *pl);
        if (st != RVC_OK) return st;
        if (!direct_out)
            HIPCHK(hipMemcpy2DAsync(out, cap * sizeof(float), pl->audio.p, pl->N * sizeof(float), pl->N * sizeof(float), B,
                                    out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, e->stream));
        e->status_queued = true;
        HIPCHK(hipStreamSynchronize(e->stream));
        st = check_status(e);
        if (st == RVC_KNN_RETRY) st = RVC_BACKEND;
    }
    return st;
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
        // every bucket's plan must stay cached until the call has run it: the cache is LRU and the call's plans are its most recent entries, so the
        // only way to lose one is more buckets than slots
        if ((int)buckets.size() > e->plan_cap) throw ShapeError("infer_batch_g: more different geometries in one call than the plan cache holds (rvc_set_plan_cache)");
        if (!e->d_state_bucket) { HIPCHK(hipMalloc(&e->d_state_bucket, sizeof(StreamState) * S)); HIPCHK(hipMalloc(&e->d_bucket_idx, sizeof(int) * S)); }
        // plans first (a geometry the engine rejects must not leave some buckets already advanced), then the work
        std::vector<std::pair<Plan *, const std::vector<int> *>> work;
        for (auto &kv : buckets) {
            Plan *pl = get_plan(e, 0, kv.first.n, kv.first.f, kv.first.sh, kv.first.rl, 0, (int)kv.second.size());
            for (int s : kv.second) { if (out_lens) out_lens[s] = pl->N; if (caps[s] < pl->N) return RVC_SHAPE; }
            work.push_back({pl, &kv.second});
        }
        for (auto &w : work)
            { bool ok = false; for (auto &p : e->plans) ok = ok || p.get() == w.first; if (!ok) throw std::logic_error("infer_batch_g: a plan of this call left the cache"); }
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
        return final_status(e);
    });
}

rvc_status rvc_synchronize(rvc_engine *e)
{
    return guarded(e, [&]() {
        HIPCHK(hipStreamSynchronize(e->stream));
        if (e->ev0 && e->last_plan) (void)hipEventElapsedTime(&e->last_ms, e->ev0, e->ev1);
        return final_status(e);
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

// EXPLORATORY, off by default and never part of the headline figure: mode 1 runs the 1-D layers with >= 128 output rows (ContentVec's projections and convolution stem, the decoder's
// 128- / 256-channel stages) as three bf16 MFMAs per fp32 product (a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulation; igemm_bf3_kernel) wherever a launch has >= 250 workgroups of 128 x 128
// (at 64 streams 2.5 of the 3.6 TFLOP of a step).  gfx950's fp32 MFMA runs at 1/16 of its bf16 rate and there is no xf32.  Results differ from the fp32 path by ~2^-16 relative
// per product (tests/test_gpu_round5.py measures it against the oracle); mode 0 is the reference's arithmetic.
rvc_status rvc_set_gemm_precision(rvc_engine *e, int mode)
{
    return guarded(e, [&]() {
        if (mode != 0 && mode != 1) throw ShapeError("gemm precision: 0 (fp32) or 1 (split bf16, exploratory)");
        e->gemm_precision = mode;
        return RVC_OK;
    });
}

// Plan cache: one plan (activation arena, composed weights, launch list) per geometry (mode, n, frame, skip_head, return_length, streams,
// retrieval on/off, pipeline slot).  Every plugin instance has its own geometry (obs-rvc/src/lib.rs:200-227): a server that serves more
// geometries than the cache holds rebuilds a plan on every miss.
// Plan-time selection by measurement (plan.hip, queue_igemm): on = 1 (default) lets plans of more than 4 streams time the eligible kernels / tiles of every
// layer on this device while the plan is built and keep the fastest; on = 0: the planner's rules only.  Cached plans are dropped when the setting changes.
rvc_status rvc_set_plan_autotune(rvc_engine *e, int on)
{
    return guarded(e, [&]() {
        const int v = on ? 1 : 0;
        if (v != e->autotune) {
            HIPCHK(hipDeviceSynchronize());
            e->plans.clear(); e->last_plan = nullptr;
            e->autotune = v;
        }
        return RVC_OK;
    });
}
// the engine's LAST plan build: layers tuned by trials in it, layers whose choice differs from the rules, layers taken from the process-wide cache of earlier
// trials, milliseconds spent in trials, milliseconds of the whole build
rvc_status rvc_plan_autotune_info(rvc_engine *e, int *tuned, int *changed, int *cache_hits, double *tune_ms, double *build_ms)
{
    return guarded(e, [&]() {
        if (tuned) *tuned = e->last_tuned;
        if (changed) *changed = e->last_changed;
        if (cache_hits) *cache_hits = e->last_hits;
        if (tune_ms) *tune_ms = e->last_tune_ms;
        if (build_ms) *build_ms = e->last_build_ms;
        return RVC_OK;
    });
}

rvc_status rvc_set_plan_cache(rvc_engine *e, int n_plans)
{
    return guarded(e, [&]() {
        if (n_plans < 2 || n_plans > 256) throw ShapeError("plan cache size out of range (2..256)");
        HIPCHK(hipDeviceSynchronize());
        e->plan_cap = n_plans;
        while ((int)e->plans.size() > e->plan_cap) { if (e->last_plan == e->plans.front().get()) e->last_plan = nullptr; e->plans.erase(e->plans.begin()); }
        return RVC_OK;
    });
}
int rvc_plan_cache_info(rvc_engine *e, int *capacity, int *cached, long long *builds)
{
    if (!e) return 0;
    if (capacity) *capacity = e->plan_cap;
    if (cached) *cached = (int)e->plans.size();
    if (builds) *builds = e->plan_builds;
    return 1;
}
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
        wfree(wsum); wfree(dg); wfree(dbeta);
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
        // host evaluation in double, the (stream, output row) pairs dealt to the host's threads (a test aid: the full-size shapes are 1e9 MACs each)
        std::vector<float> hy((size_t)streams * M * N);
        for (int b = 0; b < streams; b++)
            HIPCHK(hipMemcpy2D(&hy[(size_t)b * M * N], (size_t)N * 4, y.p + (long long)b * y.bs, (size_t)y.ld * 4, (size_t)N * 4, M, hipMemcpyDeviceToHost));
        const int nthr = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<double> errs(nthr, 0.0), sss(nthr, 0.0);
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; t++)
            pool.emplace_back([&, t]() {
                std::vector<double> ref((size_t)N);
                for (long long bm = t; bm < (long long)streams * M; bm += nthr) {
                    const int b = (int)(bm / M), m = (int)(bm % M);
                    for (int n = 0; n < N; n++) {
                        double a = bias[m];
                        for (int c = 0; c < Cin; c++)
                            for (int k = 0; k < KW; k++) {
                                const int tt = n + k * dil - pad;
                                if (tt < 0 || tt >= N) continue;
                                double v = hx[((size_t)b * Cin + c) * N + tt];
                                if (pre_act && v < 0) v *= 0.1f;
                                a += (double)w[((size_t)m * Cin + c) * KW + k] * v;
                            }
                        ref[n] = a; sss[t] += a * a;
                    }
                    const float *row = &hy[((size_t)b * M + m) * N];
                    for (int n = 0; n < N; n++) errs[t] = std::max(errs[t], std::fabs((double)row[n] - ref[n]));
                }
            });
        for (auto &th : pool) th.join();
        double err = 0.0, ss = 0.0; const size_t cnt = (size_t)streams * M * N;
        for (int t = 0; t < nthr; t++) { err = std::max(err, errs[t]); ss += sss[t]; }
        worst = err / (std::sqrt(ss / (double)std::max<size_t>(cnt, 1)) + 1e-12);
        free_conv(cw);
        return RVC_OK;
    });
    return worst;
}

// test aid: one Conv2d(Cin -> M, 3x3, pad 1, bias, ReLU) [kind 0] or ConvTranspose2d(Cin -> M, 3x3, stride 2, pad 1, output_pad 1, bias, ReLU) [kind 1] on
// `streams` images of H x W (RMVPE's layers, rvc/src/f0/rmvpe.rs:235-238), residual 0 = none, 1 = + a residual tensor, 2 = accumulate into the
// output; deterministic data, whatever kernel the planner (or a hook) picks, against a double-precision host evaluation.
// Returns the largest |gpu - host| / (rms(host) + 1e-12); negative on failure.
double rvc_debug_conv2d_check(rvc_engine *e, int M, int Cin, int H, int W, int streams, int kind, int residual)
{
    double worst = -1.0;
    (void)guarded(e, [&]() {
        if (kind == 1 && residual) throw ShapeError("transposed test layer takes no residual");
        std::vector<float> w((size_t)M * Cin * 9), bias(M);
        for (size_t i = 0; i < w.size(); i++) w[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;
        for (int m = 0; m < M; m++) bias[m] = 0.01f * (float)(m % 7) - 0.02f;
        // Conv2d: w [M][Cin][3][3]; ConvTranspose2d: w [Cin][M][3][3]
        ConvW cw = kind == 0 ? prep_conv(w.data(), bias.data(), M, Cin, 9, 1) : prep_convT2d(w.data(), bias.data(), Cin, M);
        Plan pl; pl.B = streams;
        const int OH = kind ? 2 * H : H, OW = kind ? 2 * W : W;
        T2 x = make_t2(pl.arena, streams, Cin, H, W), y = make_t2(pl.arena, streams, M, OH, OW), r = make_t2(pl.arena, streams, M, OH, OW);
        std::vector<float> hx((size_t)streams * Cin * H * W), hr((size_t)streams * M * OH * OW);
        for (size_t i = 0; i < hx.size(); i++) hx[i] = (float)(((i * 40503u) ^ (i >> 3)) % 2001) / 1000.0f - 1.0f;
        for (size_t i = 0; i < hr.size(); i++) hr[i] = (float)(((i * 9973u) ^ (i >> 2)) % 1001) / 1000.0f - 0.5f;
        for (int b = 0; b < streams; b++)
            for (int c = 0; c < Cin; c++)
                HIPCHK(hipMemcpy2D(x.p + (long long)b * x.bs + (long long)c * x.cs, (size_t)x.ld * 4, &hx[(((size_t)b * Cin + c) * H) * W], (size_t)W * 4, (size_t)W * 4, H, hipMemcpyHostToDevice));
        for (int b = 0; b < streams; b++)
            for (int c = 0; c < M; c++) {
                T2 &dst = residual == 2 ? y : r;
                HIPCHK(hipMemcpy2D(dst.p + (long long)b * dst.bs + (long long)c * dst.cs, (size_t)dst.ld * 4, &hr[(((size_t)b * M + c) * OH) * OW], (size_t)OW * 4, (size_t)OW * 4, OH, hipMemcpyHostToDevice));
            }
        ConvOpts o; o.act = ACT_RELU;
        if (residual == 1) { o.res = r.p; o.res_cs = r.cs; o.res_bs = r.bs; o.res_rs = r.ld; }
        if (residual == 2) o.accumulate = true;
        if (kind == 0) add_conv2d(pl, cw, x, y, o); else add_convT2d(pl, cw, x, y, o);
        HIPCHK(hipDeviceSynchronize());
        for (auto &op : pl.ops.v) op(e->stream);
        HIPCHK(hipStreamSynchronize(e->stream));
        HIPCHK(hipGetLastError());
        std::vector<float> hy((size_t)OH * OW);
        std::vector<double> ref((size_t)OH * OW);
        double err = 0.0, ss = 0.0; size_t cnt = 0;
        for (int b = 0; b < streams; b++)
            for (int m = 0; m < M; m++) {
                HIPCHK(hipMemcpy2D(hy.data(), (size_t)OW * 4, y.p + (long long)b * y.bs + (long long)m * y.cs, (size_t)y.ld * 4, (size_t)OW * 4, OH, hipMemcpyDeviceToHost));
                for (int oh = 0; oh < OH; oh++)
                    for (int ow = 0; ow < OW; ow++) {
                        double a = bias[m];
                        for (int c = 0; c < Cin; c++) {
                            const float *xc = &hx[(((size_t)b * Cin + c) * H) * W];
                            for (int kh = 0; kh < 3; kh++)
                                for (int kw = 0; kw < 3; kw++) {
                                    if (kind == 0) {
                                        const int ih = oh + kh - 1, iw = ow + kw - 1;
                                        if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
                                        a += (double)w[(((size_t)m * Cin + c) * 3 + kh) * 3 + kw] * xc[(size_t)ih * W + iw];
                                    } else {
                                        // out[oh] += w[kh] * in[ih] with oh = 2 ih - 1 + kh
                                        const int th = oh + 1 - kh, tw = ow + 1 - kw;
                                        if (th < 0 || tw < 0 || (th & 1) || (tw & 1)) continue;
                                        const int ih = th / 2, iw = tw / 2;
                                        if (ih >= H || iw >= W) continue;
                                        a += (double)w[(((size_t)c * M + m) * 3 + kh) * 3 + kw] * xc[(size_t)ih * W + iw];
                                    }
                                }
                        }
                        a = a > 0 ? a : 0;
                        if (residual) a += hr[(((size_t)b * M + m) * OH + oh) * OW + ow];
                        ref[(size_t)oh * OW + ow] = a; ss += a * a; cnt++;
                    }
                for (size_t i = 0; i < ref.size(); i++) err = std::max(err, std::fabs((double)hy[i] - ref[i]));
            }
        worst = err / (std::sqrt(ss / (double)std::max<size_t>(cnt, 1)) + 1e-12);
        free_conv(cw);
        return RVC_OK;
    });
    return worst;
}

// the kernel family of the most recently queued implicit-GEMM launch (the first word of its description): tests assert which path they exercised
const char *rvc_debug_last_kernel(void)
{
    static thread_local std::string buf;
    buf = g_last_kernel;
    return buf.c_str();
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
        snprintf(ln, sizeof ln, "%.2f %.4f %s\n", t * 1e3, pl.prof[i].flops * 1e-9, pl.prof[i].desc >= 0 ? pl.descs[pl.prof[i].desc].c_str() : (pl.prof[i].bytes > 0 ? "knn_scan_select" : "?"));
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
