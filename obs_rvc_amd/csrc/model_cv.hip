// model_cv.hip -- ContentVec / HuBERT-base feature extractor as a plan (reference: rvc/src/rvc.rs:81-97, ort::Session::run at rvc.rs:92)
#include "engine_int.h"

namespace rvc {

// ------------------------------- ContentVec ------------------------------------------
T1 build_contentvec(rvc_engine *e, Plan &pl, int B, size_t L)
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
    const bool fuse_ln = B <= LN_FOLD_MAX_STREAMS && m.has_folded && !pl.plain_plan && !test_opt("RVC_NO_LN_FUSE");
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
        { ConvOpts o; o.bf3 = pl.bf3; add_conv1d(pl, Ly.qkv, h2, qkv, 1, 0, 1, o); }
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
        { ConvOpts o; o.bf3 = pl.bf3; o.res = h2.p; o.res_cs = h2.ld; o.res_bs = h2.bs; add_conv1d(pl, Ly.o, att, h2, 1, 0, 1, o); }
        add_layernorm(pl, h2, Ly.ln1_g, Ly.ln1_b);
        { ConvOpts o; o.bf3 = pl.bf3; o.act = ACT_GELU; add_conv1d(pl, Ly.ff1, h2, ff, 1, 0, 1, o); }
        { ConvOpts o; o.bf3 = pl.bf3; o.res = h2.p; o.res_cs = h2.ld; o.res_bs = h2.bs; add_conv1d(pl, Ly.ff2, ff, h2, 1, 0, 1, o); }
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


void cv_kernel_attrs()
{
    HIPCHK(hipFuncSetAttribute((const void *)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
}
}  // namespace rvc
