// model_synth.hip -- the synthesizer (text encoder, prior sample, flows, NSF source, HiFiGAN decoder) as a plan (reference: rvc/src/rvc.rs:182-214, ort::Session::run at rvc.rs:195)
#include "engine_int.h"

namespace rvc {

// ------------------------------- synthesizer ------------------------------------------
// NSF harmonic source: depends only on the f0 branch, so it is queued on that branch's stream
T1 build_nsf_source(rvc_engine *e, Plan &pl, int B, float *d_pitchf)
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
std::vector<T1> build_noise_convs(rvc_engine *e, Plan &pl, int B, const T1 &src)
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

void build_synth(rvc_engine *e, Plan &pl, int B, const T1 &phone, const T1 &src, float *d_pitchf, int *d_pitch, int src_join_sid,
                        const std::vector<T1> *nz)
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
        const bool fuse_ln = B <= LN_FOLD_MAX_STREAMS && m.has_folded && !pl.plain_plan && !test_opt("RVC_NO_LN_FUSE");
        bool raw = false; const float *raw_g = nullptr, *raw_b = nullptr; float *raw_st = nullptr;
        for (int l = 0; l < m.enc_layers; l++) {
            ModelSY::Layer &Ly = m.layers[l];
            if (raw) { raw_st = A.floats((size_t)2 * R + 16); ConvOpts o; o.ln_wsum = Ly.qkv_wsum; o.ln_stats_out = raw_st; o.ln_rows = H; add_conv1d(pl, Ly.qkv_f, x, qkv, 1, 0, 1, o); }
            else add_conv1d(pl, Ly.qkv, x, qkv, 1, 0, 1);
            AttnP ap{}; ap.qkv = qkv.p; ap.out = att.p; ap.E = H; ap.T = R; ap.heads = m.heads; ap.cs = qkv.ld; ap.bs = qkv.bs; ap.o_cs = att.ld; ap.o_bs = att.bs;
            ap.scale = 1.0f / sqrtf((float)kc); ap.rel_k = Ly.rel_k; ap.rel_v = Ly.rel_v; ap.window = m.window;
            const size_t small_lds = ((size_t)2 * kc * Tp + 2 * (2 * m.window + 1) * kc + 4 * kc + 4 * 64) * sizeof(float);
            // the matrix-core form (VALU form at one stream: 12.4 us per layer of dependent LDS reads; round 6: at every stream count -- 21.5 -> ~12 us per launch at 64
            // streams, step 5 / 8 / 16 / 64 streams 4.585 / 6.156 / 11.03 / 34.24 -> 4.577 / 6.145 / 11.00 / 34.20 ms; test hook RVC_RELPOS_MFMA_MAX = 4: the round-5 rule)
            const int a_tp = R | 1, a_nr = 2 * m.window + 1, a_jf = (R + 15) / 16, a_pw = (a_nr + 15) / 16 * 16, a_nrp = (a_nr + 3) / 4 * 4;
            const size_t mfma_lds = ((size_t)kc * 16 + 2 * (size_t)kc * a_tp + (size_t)a_pw * kc + (size_t)a_nrp * kc + 16 * a_jf * 16 + 2 * 16 * a_pw + 64) * sizeof(float);
            if (B <= test_opt_int("RVC_RELPOS_MFMA_MAX", 1 << 20) && R <= 64 && kc % 16 == 0 && mfma_lds <= 160 * 1024 && !tune_env("RVC_NO_SMALL_ATTN") && !tune_env("RVC_ATTN_VALU") && !tune_env("RVC_NO_SMALL_ATTN_MFMA")) {
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
            pl.ops.push_back([=](hipStream_t s) { hipLaunchKernelGGL(flip_channels_kernel, grid, dim3(256), 0, s, zi.p, zi.ld, zi.bs, zo.p, zo.ld, zo.bs, I, R); });
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
        // Chains issued one after the other (many streams): the average is taken by the chains' last convolutions themselves -- chain j's
        // epilogue stores (j = 0) or adds (j > 0) its output x 1 / n_rb -- instead of three stored tensors and an averaging launch per
        // stage (round 6: 4 launches and ~1 GB of reads + writes per 64-stream step).  The split-bf16 kernels have no accumulating
        // epilogue: those plans keep the averaging launch.
        const bool mean_in_epilogue = !fused && !pl.bf3 && test_opt_int("RVC_MEAN3", 0) == 0;     // (test hook RVC_MEAN3 = 1: the averaging launch)
        for (int j = 0; j < m.n_rb && !fused; j++) {
            const int k = m.rb_k[j];
            T1 ra = make_t1(A, B, co, Tn, DH), rb = make_t1(A, B, co, Tn, DH), tt = make_t1(A, B, co, Tn, DH), fin = mean_in_epilogue ? xs : make_t1(A, B, co, Tn, 0);
            T1 cur = u;
            for (int q = 0; q < m.n_rbd; q++) {
                const int d = m.rb_d[q];
                { ConvOpts o; o.pre_act = ACT_LRELU; o.pre_slope = 0.1f; o.act = ACT_LRELU; o.slope = 0.1f; add_conv1d(pl, m.rbs[i][j][q].first, cur, tt, 1, (k * d - d) / 2, d, o); }
                const bool last = q == m.n_rbd - 1;
                T1 dst = last ? fin : (cur.p == ra.p ? rb : ra);
                ConvOpts o; o.res = cur.p; o.res_cs = cur.ld; o.res_bs = cur.bs;
                if (last && mean_in_epilogue) { o.scale = 1.0f / (float)m.n_rb; o.accumulate = j > 0; }
                add_conv1d(pl, m.rbs[i][j][q].second, tt, dst, 1, (k - 1) / 2, 1, o);
                cur = dst;
            }
            finals.push_back(fin);
        }
        if (!mean_in_epilogue) {
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
    pl.out_direct_ok = pl.audio.ld == Tc && !pl.with_taps && pl.final_out_honoured;     // (queue_igemm says whether the launch path it chose writes the caller's buffer)
    if (pl.audio.ld != Tc) {
        // make the output rows contiguous [B][N] for the device-pointer API
        T1 a2; a2.p = A.floats((size_t)B * Tc); a2.B = B; a2.C = 1; a2.T = Tc; a2.ld = Tc; a2.halo = 0; a2.bs = Tc;
        T1 a1 = pl.audio;
        pl.ops.push_back([=](hipStream_t s) { HIPCHK(hipMemcpy2DAsync(a2.p, (size_t)Tc * 4, a1.p, (size_t)a1.bs * 4, (size_t)Tc * 4, B, hipMemcpyDeviceToDevice, s)); });
        pl.audio = a2;
    }
    add_stamp(pl, "sy.audio");
}


void synth_kernel_attrs()
{
    HIPCHK(hipFuncSetAttribute((const void *)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)relpos_attention_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)relpos_attention_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}
}  // namespace rvc
