// model_rmvpe.hip -- RMVPE f0 estimator (mel front end, U-Net, BiGRU, salience) and the decode / pitch-cache step as a plan (reference: rvc/src/f0/rmvpe.rs:118-133, 225-248; rvc/src/rvc.rs:111-131, 167-180)
#include "engine_int.h"
#include "rmblock.hip.h"

namespace rvc {

// One launch per ConvBlockRes on the shallow levels (rm_block_kernel, rmblock.hip.h): few streams only -- there the f0 branch is a chain of dependent
// 5-8 us launches on its own CU partition and the block's halo recomputation costs nothing that matters; with many streams folded into a launch the two
// convolutions fill the chip by themselves and keep the implicit-GEMM kernels.  Test hook RVC_RM_FUSE = 0: never, 2: at any stream count.  false = not taken.
// pool_src != nullptr: the block's input is AvgPool2d(2, 2) of that tensor, averaged while the tile is staged (the pooling launch in front of the block disappears);
// dry: eligibility only, nothing queued
static bool add_rm_block_fused(Plan &pl, const ResBlockW &w, const T2 &x, const T2 &out, const ResBlockW *next, const T2 *pool_src = nullptr, bool dry = false, const T2 *pool_dst = nullptr)
{
    const int mode = test_opt_int("RVC_RM_FUSE", 1);
    if (!w.f_w1 || mode == 0 || (!pl.rm_fuse && mode != 2)) return false;
    if (x.C != w.ci || out.C != w.co || out.H != x.H || out.W != x.W || (!w.has_sc && w.ci != w.co)) return false;
    const int MT = w.co / 16;
    RmBlockP q{};
    q.x = x.p; q.y = out.p; q.Cin = w.ci; q.Cin4 = (w.ci + 15) / 16 * 4; q.Cout = w.co;
    q.H = x.H; q.W = x.W;
    // output tile per workgroup: 128 / 32 / 8 pixels for 16 / 32 / 64 channels (32 workgroups per stream at the 32 x 128 mel image: one per CU of the
    // f0 partition); the y1 tile must fit the waves' n-tiles: ceil((TH + 2)(TW + 2) / 16) <= 4 * (4 / MT)
    q.TH = MT == 1 ? 8 : (MT == 2 ? 4 : 2); q.TW = MT == 1 ? 16 : (MT == 2 ? 8 : 4);
    q.TH = std::min(q.TH, x.H); q.TW = std::min(q.TW, x.W);
    // n-tiles per wave of the two convolutions (template parameters of the kernel): 3 / 2 for 16 channels (12 and 8 tiles over four waves), 2 / 1 for 32 and 64
    const int NT1 = MT == 1 ? 3 : 2, NT2 = MT == 1 ? 2 : 1, NWN = 4 / MT;
    if (MT > 2) return false;          // (64 channels: every workgroup would stream 2 x 147 KB of weights for 8 output pixels -- measured 19 us against 11.6 for the two launches)
    if (((q.TH + 2) * (q.TW + 2) + 15) / 16 > NT1 * NWN || (q.TH * q.TW + 15) / 16 > NT2 * NWN || (q.TH + 4) * (q.TW + 4) > 256) return false;      // (the staging gives every tile position a thread)
    q.tiles_x = (x.W + q.TW - 1) / q.TW;
    const int tiles = q.tiles_x * ((x.H + q.TH - 1) / q.TH);
    q.x_ld = x.ld; q.x_cs = x.cs; q.x_bs = x.bs; q.y_ld = out.ld; q.y_cs = out.cs; q.y_bs = out.bs;
    if (pool_src) { q.x = pool_src->p; q.x_ld = pool_src->ld; q.x_cs = pool_src->cs; q.x_bs = pool_src->bs; q.pool = 1; }
    q.w1 = w.f_w1; q.w2 = w.f_w2; q.wsc = w.has_sc ? w.f_sc : nullptr;
    q.b1 = w.c1.bias; q.b2 = w.c2.bias; q.bsc = w.has_sc ? w.sc.bias : nullptr;
    q.wlines = w.f_lines; q.wnext = next ? next->f_w1 : nullptr; q.wnext_lines = next ? next->f_lines : 0;
    if (q.wlines > 1536 || q.wnext_lines > 1536) return false;
    auto stride = [](int n) { int v = n / 32 * 32 + 16; return v < n ? v + 32 : v; };      // >= n and 16 mod 32: the k-slots of a read alternate between the two bank halves
    q.XS = stride((q.TH + 4) * (q.TW + 4)); q.YS = stride((q.TH + 2) * (q.TW + 2));
    const size_t lds = ((size_t)q.Cin4 * 4 * q.XS + (size_t)q.Cout * q.YS) * sizeof(float);
    if (lds > 64 * 1024) return false;
    if (pool_dst) {          // second output: the pooled tensor (tiles of eight columns; even image and tile sizes)
        if (q.TW != 8 || (q.TH & 1) || (x.H & 1) || (x.W & 1) || pool_dst->H * 2 != x.H || pool_dst->W * 2 != x.W || pool_dst->C != w.co) return false;
        q.ypool = pool_dst->p; q.p_ld = pool_dst->ld; q.p_cs = pool_dst->cs; q.p_bs = pool_dst->bs;
    }
    if (dry) return true;
    q.tiles = tiles;
    const dim3 grid((unsigned)(tiles + (q.wnext ? 1 : 0)), (unsigned)x.B);
    const double flops = 2.0 * w.co * (double)x.H * x.W * (9.0 * w.ci + 9.0 * w.co + (w.has_sc ? w.ci : 0)) * x.B;
    pl.igemm_flops += flops; pl.n_igemm++;
    Plan *plp = &pl;
    { char d[176]; snprintf(d, sizeof d, "rmb M=%d N=%d K=%d B=%d nph=1 tile=%dx%d grid=%ux%u lds=%zu sc=%d", w.co, x.H * x.W, 9 * w.ci + 9 * w.co + (w.has_sc ? w.ci : 0), x.B, q.TH, q.TW, grid.x, grid.y, lds, (int)w.has_sc); pl.descs.push_back(d); }
    const int desc_id = (int)pl.descs.size() - 1;
    pl.ops.push_back([=](hipStream_t s) {
        ProfEvent *pe = nullptr;
        if (plp->profile) {
            if (plp->prof_used == plp->prof.size()) { ProfEvent e; HIPCHK(hipEventCreate(&e.a)); HIPCHK(hipEventCreate(&e.b)); e.flops = 0; e.bytes = 0; plp->prof.push_back(e); }
            pe = &plp->prof[plp->prof_used++]; pe->flops = flops; pe->bytes = 0; pe->desc = desc_id;
        }
        hipEvent_t ea = pe ? pe->a : nullptr, eb = pe ? pe->b : nullptr;
#define RVC_RMB_LAUNCH(MT_, N1_, N2_)                                                                                         \
        { if (ea) hipExtLaunchKernelGGL((rm_block_kernel<MT_, N1_, N2_>), grid, dim3(256), (uint32_t)lds, s, ea, eb, 0, q);  \
          else hipLaunchKernelGGL((rm_block_kernel<MT_, N1_, N2_>), grid, dim3(256), lds, s, q); }
        if (MT == 1) RVC_RMB_LAUNCH(1, 3, 2) else if (MT == 2) RVC_RMB_LAUNCH(2, 2, 1) else RVC_RMB_LAUNCH(4, 2, 1)
#undef RVC_RMB_LAUNCH
    });
    return true;
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
        float *dw = upload_fragments(panel, 1, c1.M, Kp3, 1);      // plan-lifetime copy
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

static T2 res_block(Plan &pl, const ResBlockW &w, const T2 &x, const T2 &out, const ResBlockW *next = nullptr, const T2 *pool_src = nullptr, const T2 *pool_dst = nullptr)
{
    Arena &A = pl.arena;
    if (add_rm_block_fused(pl, w, x, out, (next && next->f_w1) ? next : nullptr, pool_src, false, pool_dst)) return out;
    if (pool_src || pool_dst) throw std::runtime_error("RMVPE: pooled input without the fused block");
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

T1 build_rmvpe(rvc_engine *e, Plan &pl, int B, size_t L, size_t frame16k, bool update_cache)
{
    ModelRM &m = *e->rm;
    Arena &A = pl.arena;
    // fused shallow blocks (rm_block_kernel): where the f0 branch runs on its own single-XCD partition -- up to four streams, and not the v1 engines of up to
    // three streams, whose branch has two XCDs (the kernel's 32 workgroups per stream and its L2 warming are sized for one: v1 at one stream measured 1.92 ->
    // 1.94 ms with it, v2 1.99 -> 1.965)
    pl.rm_fuse = e->partitioned && (g_ncu - e->cv_cus) * 8 == g_ncu && B <= 4;
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
    T2 pool_from; bool pooled_in_block = false;         // the previous level's output when its pooling is folded into this level's first (fused) block
    for (int lv = 0; lv < m.levels; lv++) {
        const int co = m.enc[lv][0].co;
        T2 p = make_t2(A, B, co, H / 2, W / 2);
        // ... and a level whose LAST block is a fused block with eight-column tiles writes the pooled tensor as a second output of that block
        const bool next_takes_pool = lv + 1 < m.levels && (H % 2) == 0 && (W % 2) == 0 && test_opt_int("RVC_RM_FUSE", 1) != 3 &&
                                     [&]() { T2 o = p; o.C = m.enc[lv + 1][0].co; return add_rm_block_fused(pl, m.enc[lv + 1][0], p, o, nullptr, &x, true); }();
        bool pooled_by_block = false;
        for (int j = 0; j < m.n_blocks; j++) {
            T2 out = (j == m.n_blocks - 1) ? cat[lv].chans(co, co) : make_t2(A, B, co, H, W);
            const bool last = j == m.n_blocks - 1;
            const T2 xin = x;
            pooled_by_block = last && !next_takes_pool && test_opt_int("RVC_RM_FUSE", 1) != 3 && add_rm_block_fused(pl, m.enc[lv][j], xin, out, nullptr, (j == 0 && pooled_in_block) ? &pool_from : nullptr, true, &p);
            x = res_block(pl, m.enc[lv][j], xin, out, j + 1 < m.n_blocks ? &m.enc[lv][j + 1] : (lv + 1 < m.levels ? &m.enc[lv + 1][0] : nullptr), (j == 0 && pooled_in_block) ? &pool_from : nullptr,
                          pooled_by_block ? &p : nullptr);
        }
        if (pl.with_taps) { char nm[32]; snprintf(nm, sizeof nm, "rm.enc%d", lv); add_tap2(pl, nm, x); }
        // the next level's first block stages AvgPool2d(2, 2) of this level's output itself when it is a fused block (one more launch off the f0 branch)
        pooled_in_block = next_takes_pool;
        if (pooled_in_block) pool_from = x;
        else if (!pooled_by_block) {
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
        for (int j = 0; j < m.n_blocks; j++) { T2 out = make_t2(A, B, co, H, W); x = res_block(pl, m.dec[lv][j], x, out, j + 1 < m.n_blocks ? &m.dec[lv][j + 1] : (lv + 1 < m.levels ? &m.dec[lv + 1][0] : nullptr)); }
        if (pl.with_taps) { char nm[32]; snprintf(nm, sizeof nm, "rm.dec%d", lv); add_tap2(pl, nm, x); }
    }
    const int Hg = m.gru_hidden, I = 3 * m.n_mels;
    T1 feat = make_t1(A, B, I, Tm, 0), gi = make_t1(A, B, 6 * Hg, Tm, 0), gout = make_t1(A, B, 2 * Hg, Tm, 0), sal = make_t1(A, B, m.n_out, Tm, 0);
    if (H == Tm && W == m.n_mels && test_opt_int("RVC_RM_FUSE", 1) != 0) {
        // (round 6) the head convolution writes the GRU's input layout itself -- feat[c * n_mels + mel][t] = conv[c][t][mel]: channel stride n_mels rows, image row
        // (time) stride 1, image column (mel) stride one row of `feat` -- instead of an image and a transposing launch behind it
        T2 ft; ft.p = feat.p; ft.B = B; ft.C = 3; ft.H = H; ft.W = W; ft.cs = m.n_mels * feat.ld; ft.ld = 1; ft.bs = feat.bs;
        ConvOpts o; o.y_ws = feat.ld;
        add_conv2d(pl, m.cnn, x, ft, o);
    } else {
        T2 cn = make_t2(A, B, 3, H, W);
        add_conv2d(pl, m.cnn, x, cn);
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
            // eager launches: the tags advance by Tm per launch (host-side counter of this plan), so the granules of earlier chunks are stale by construction and
            // the buffer is zeroed only in front of the plan's first launch (and when the 32-bit tag space runs out); a captured graph bakes its arguments and keeps the
            // memset node + epoch 0
            auto next_epoch = std::make_shared<unsigned>(0u);
            auto dirty = std::make_shared<bool>(true);
            pl.ops.push_back([=](hipStream_t s) {
                GruMultiP g2 = gp;
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing(s, &cs);
                if (cs != hipStreamCaptureStatusNone) { HIPCHK(hipMemsetAsync(g2.gran, 0, gbytes, s)); g2.epoch = 0; *dirty = true; }
                else if (*dirty || *next_epoch > 0xFFF00000u) { HIPCHK(hipMemsetAsync(g2.gran, 0, gbytes, s)); g2.epoch = 0; *next_epoch = (unsigned)Tm; *dirty = false; }
                else { g2.epoch = *next_epoch; *next_epoch += (unsigned)Tm; }
                hipLaunchKernelGGL(gru_multi_kernel, g3, dim3(384), lds3, s, g2);
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
void build_pitch_post(rvc_engine *e, Plan &pl, int B, const T1 &sal, bool update_cache, size_t frame16k, size_t hubert_length,
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


void rmvpe_kernel_attrs()
{
    HIPCHK(hipFuncSetAttribute((const void *)gru_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}
}  // namespace rvc
