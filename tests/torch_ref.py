"""Independent torch-CPU implementations of the three networks, used ONLY to
pin the C oracle's dense layers (the oracle's im2col+sgemm code vs torch's conv/linear/GRU kernels, and
ContentVec vs the HuggingFace HuBERT class).  Test infrastructure; never imported by the product."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float()


# ------------------------------------------------------------------------------------------------
# ContentVec through transformers.HubertModel with the blob's weights copied in
# ------------------------------------------------------------------------------------------------
def contentvec_hf(cfg, tens, wav: np.ndarray) -> np.ndarray:
    """Returns (C_out, T) like the oracle's contentvec_forward."""
    from transformers import HubertConfig, HubertModel

    E, C = int(cfg["embed"]), int(cfg["conv_dim"])
    hc = HubertConfig(
        hidden_size=E, num_hidden_layers=int(cfg["run_layers"]), num_attention_heads=int(cfg["heads"]),
        intermediate_size=int(cfg["ffn"]), hidden_act="gelu", hidden_dropout=0.0, activation_dropout=0.0,
        attention_dropout=0.0, feat_proj_dropout=0.0, final_dropout=0.0, layerdrop=0.0, feat_proj_layer_norm=True,
        feat_extract_norm="group", feat_extract_activation="gelu", conv_dim=[C] * 7,
        conv_stride=[int(cfg["conv_s%d" % i]) for i in range(7)], conv_kernel=[int(cfg["conv_k%d" % i]) for i in range(7)],
        conv_bias=False, num_conv_pos_embeddings=int(cfg["pos_k"]), num_conv_pos_embedding_groups=int(cfg["pos_groups"]),
        do_stable_layer_norm=False, apply_spec_augment=False, layer_norm_eps=1e-5)
    m = HubertModel(hc).eval()
    sd = {}
    for i in range(7):
        sd["feature_extractor.conv_layers.%d.conv.weight" % i] = _t(tens["cv.conv%d.w" % i])
    sd["feature_extractor.conv_layers.0.layer_norm.weight"] = _t(tens["cv.gn.g"])
    sd["feature_extractor.conv_layers.0.layer_norm.bias"] = _t(tens["cv.gn.b"])
    sd["feature_projection.layer_norm.weight"] = _t(tens["cv.ln0.g"])
    sd["feature_projection.layer_norm.bias"] = _t(tens["cv.ln0.b"])
    sd["feature_projection.projection.weight"] = _t(tens["cv.proj.w"])
    sd["feature_projection.projection.bias"] = _t(tens["cv.proj.b"])
    sd["encoder.layer_norm.weight"] = _t(tens["cv.enc_ln.g"])
    sd["encoder.layer_norm.bias"] = _t(tens["cv.enc_ln.b"])
    for l in range(int(cfg["run_layers"])):
        p, q = "encoder.layers.%d." % l, "cv.l%d." % l
        for hf, mine in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v"), ("out_proj", "o")):
            sd[p + "attention.%s.weight" % hf] = _t(tens[q + mine + ".w"])
            sd[p + "attention.%s.bias" % hf] = _t(tens[q + mine + ".b"])
        sd[p + "layer_norm.weight"] = _t(tens[q + "ln1.g"]); sd[p + "layer_norm.bias"] = _t(tens[q + "ln1.b"])
        sd[p + "feed_forward.intermediate_dense.weight"] = _t(tens[q + "ff1.w"]); sd[p + "feed_forward.intermediate_dense.bias"] = _t(tens[q + "ff1.b"])
        sd[p + "feed_forward.output_dense.weight"] = _t(tens[q + "ff2.w"]); sd[p + "feed_forward.output_dense.bias"] = _t(tens[q + "ff2.b"])
        sd[p + "final_layer_norm.weight"] = _t(tens[q + "ln2.g"]); sd[p + "final_layer_norm.bias"] = _t(tens[q + "ln2.b"])
    missing, unexpected = m.load_state_dict(sd, strict=False)
    # the positional conv is weight-normed in HF; bypass the parametrisation by calling the conv functionally below
    missing = [k for k in missing if "pos_conv_embed" not in k and "masked_spec_embed" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    pos_w, pos_b = _t(tens["cv.pos.w"]), _t(tens["cv.pos.b"])
    with torch.no_grad():
        x = _t(wav)[None]
        feats = m.feature_extractor(x).transpose(1, 2)        # (1, T, C)
        h = m.feature_projection(feats)
        if isinstance(h, tuple):
            h = h[0]
        pk = int(cfg["pos_k"])
        pc = F.conv1d(h.transpose(1, 2), pos_w, pos_b, padding=pk // 2, groups=int(cfg["pos_groups"]))
        if pk % 2 == 0:
            pc = pc[:, :, :-1]
        h = h + F.gelu(pc).transpose(1, 2)
        h = m.encoder.layer_norm(h)
        for layer in m.encoder.layers:
            out = layer(h)
            h = out[0] if isinstance(out, tuple) else out
        if int(cfg["out_dim"]) != E:
            h = F.linear(h, _t(tens["cv.final_proj.w"]), _t(tens["cv.final_proj.b"]))
    return h[0].transpose(0, 1).contiguous().numpy()


# ------------------------------------------------------------------------------------------------
# RMVPE network
# ------------------------------------------------------------------------------------------------
def _cbr(t, pre, x):
    y = F.relu(F.conv2d(x, _t(t[pre + "c1.w"]), _t(t[pre + "c1.b"]), padding=1))
    y = F.relu(F.conv2d(y, _t(t[pre + "c2.w"]), _t(t[pre + "c2.b"]), padding=1))
    if pre + "sc.w" in t:
        w = _t(t[pre + "sc.w"])
        return y + F.conv2d(x, w[:, :, None, None], _t(t[pre + "sc.b"]))
    return y + x


def rmvpe_salience(cfg, t, mel: np.ndarray) -> np.ndarray:
    """mel (128, Tm) -> salience (Tm, 360)."""
    levels, nb, inter = int(cfg["levels"]), int(cfg["n_blocks"]), int(cfg["inter_layers"])
    H = int(cfg["gru_hidden"])
    with torch.no_grad():
        x = _t(mel).transpose(0, 1)[None, None] * float(t["rm.bn0"][0]) + float(t["rm.bn0"][1])
        skips = []
        for lv in range(levels):
            for j in range(nb):
                x = _cbr(t, "rm.enc%d.b%d." % (lv, j), x)
            skips.append(x)
            x = F.avg_pool2d(x, 2)
        for lv in range(inter):
            for j in range(nb):
                x = _cbr(t, "rm.int%d.b%d." % (lv, j), x)
        for lv in range(levels):
            x = F.relu(F.conv_transpose2d(x, _t(t["rm.dec%d.up.w" % lv]), _t(t["rm.dec%d.up.b" % lv]), stride=2, padding=1, output_padding=1))
            x = torch.cat([x, skips[-1 - lv]], dim=1)
            for j in range(nb):
                x = _cbr(t, "rm.dec%d.b%d." % (lv, j), x)
        x = F.conv2d(x, _t(t["rm.cnn.w"]), _t(t["rm.cnn.b"]), padding=1)
        x = x.transpose(1, 2).flatten(-2)                 # (1, Tm, 3*n_mels)
        gru = torch.nn.GRU(x.shape[-1], H, num_layers=1, batch_first=True, bidirectional=True)
        gru.weight_ih_l0.copy_(_t(t["rm.gru.w_ih_f"])); gru.weight_hh_l0.copy_(_t(t["rm.gru.w_hh_f"]))
        gru.bias_ih_l0.copy_(_t(t["rm.gru.b_ih_f"])); gru.bias_hh_l0.copy_(_t(t["rm.gru.b_hh_f"]))
        gru.weight_ih_l0_reverse.copy_(_t(t["rm.gru.w_ih_b"])); gru.weight_hh_l0_reverse.copy_(_t(t["rm.gru.w_hh_b"]))
        gru.bias_ih_l0_reverse.copy_(_t(t["rm.gru.b_ih_b"])); gru.bias_hh_l0_reverse.copy_(_t(t["rm.gru.b_hh_b"]))
        y, _ = gru(x)
        y = torch.sigmoid(F.linear(y, _t(t["rm.fc.w"]), _t(t["rm.fc.b"])))
    return y[0].numpy()


# ------------------------------------------------------------------------------------------------
# synthesizer
# ------------------------------------------------------------------------------------------------
def _ln_c(x, g, b):
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), g, b, 1e-5).transpose(1, 2)


def _rel_attn(t, l, x, heads, window):
    pre = "sy.enc.l%d." % l
    q = F.conv1d(x, _t(t[pre + "q.w"])[:, :, None], _t(t[pre + "q.b"]))
    k = F.conv1d(x, _t(t[pre + "k.w"])[:, :, None], _t(t[pre + "k.b"]))
    v = F.conv1d(x, _t(t[pre + "v.w"])[:, :, None], _t(t[pre + "v.b"]))
    b, d, T = q.shape
    kc = d // heads
    q = q.view(b, heads, kc, T).transpose(2, 3)
    k = k.view(b, heads, kc, T).transpose(2, 3)
    v = v.view(b, heads, kc, T).transpose(2, 3)
    scores = torch.matmul(q / math.sqrt(kc), k.transpose(-2, -1))
    rk, rv = _t(t[pre + "rel_k"]), _t(t[pre + "rel_v"])
    idx = torch.arange(T)[None, :] - torch.arange(T)[:, None]           # j - i
    valid = (idx.abs() <= window)
    e = (idx + window).clamp(0, 2 * window)
    relk = rk[e] * valid[..., None]                                      # (T, T, kc)
    scores = scores + torch.einsum("bhid,ijd->bhij", q / math.sqrt(kc), relk)
    p = F.softmax(scores, dim=-1)
    out = torch.matmul(p, v)
    relv = rv[e] * valid[..., None]
    out = out + torch.einsum("bhij,ijd->bhid", p, relv)
    out = out.transpose(2, 3).contiguous().view(b, d, T)
    return F.conv1d(out, _t(t[pre + "o.w"])[:, :, None], _t(t[pre + "o.b"]))


def synth_until_z(cfg, t, phone: np.ndarray, pitch: np.ndarray, eps: np.ndarray):
    """TextEncoder + prior + reverse flow.  phone (R, C), pitch int (R,), eps (inter, R) -> (enc_out, stats, z)."""
    Hd, I, heads, window = int(cfg["hidden"]), int(cfg["inter"]), int(cfg["heads"]), int(cfg["window"])
    ek, wk, wl, fn = int(cfg["enc_k"]), int(cfg["wn_k"]), int(cfg["wn_layers"]), int(cfg["flow_n"])
    g = _t(t["sy.g"])[None, :, None]
    with torch.no_grad():
        x = F.linear(_t(phone)[None], _t(t["sy.enc.phone.w"]), _t(t["sy.enc.phone.b"])) + _t(t["sy.enc.pitch_emb"])[torch.from_numpy(pitch.astype(np.int64))][None]
        x = F.leaky_relu(x * math.sqrt(Hd), 0.1).transpose(1, 2)
        for l in range(int(cfg["enc_layers"])):
            pre = "sy.enc.l%d." % l
            x = _ln_c(x + _rel_attn(t, l, x, heads, window), _t(t[pre + "ln1.g"]), _t(t[pre + "ln1.b"]))
            y = F.conv1d(F.relu(F.conv1d(x, _t(t[pre + "ff1.w"]), _t(t[pre + "ff1.b"]), padding=ek // 2)), _t(t[pre + "ff2.w"]), _t(t[pre + "ff2.b"]), padding=ek // 2)
            x = _ln_c(x + y, _t(t[pre + "ln2.g"]), _t(t[pre + "ln2.b"]))
        enc = x
        stats = F.conv1d(x, _t(t["sy.enc.proj.w"])[:, :, None], _t(t["sy.enc.proj.b"]))
        m, logs = stats[:, :I], stats[:, I:]
        z = m + torch.exp(logs) * _t(eps)[None] * 0.66666
        half = I // 2
        for fi in reversed(range(fn)):
            z = torch.flip(z, [1])
            pre = "sy.flow%d." % fi
            x0, x1 = z[:, :half], z[:, half:]
            h = F.conv1d(x0, _t(t[pre + "pre.w"])[:, :, None], _t(t[pre + "pre.b"]))
            cond = F.conv1d(g, _t(t[pre + "cond.w"])[:, :, None], _t(t[pre + "cond.b"]))
            out = torch.zeros_like(h)
            for j in range(wl):
                a = F.conv1d(h, _t(t[pre + "in%d.w" % j]), _t(t[pre + "in%d.b" % j]), padding=(wk - 1) // 2) + cond[:, j * 2 * Hd:(j + 1) * 2 * Hd]
                acts = torch.tanh(a[:, :Hd]) * torch.sigmoid(a[:, Hd:])
                rs = F.conv1d(acts, _t(t[pre + "rs%d.w" % j])[:, :, None], _t(t[pre + "rs%d.b" % j]))
                if j < wl - 1:
                    h = h + rs[:, :Hd]
                    out = out + rs[:, Hd:]
                else:
                    out = out + rs
            mm = F.conv1d(out, _t(t[pre + "post.w"])[:, :, None], _t(t[pre + "post.b"]))
            z = torch.cat([x0, x1 - mm], 1)
    return enc[0].numpy(), stats[0].numpy(), z[0].numpy()


def synth_decoder(cfg, t, z: np.ndarray, src: np.ndarray) -> np.ndarray:
    """NSF-HiFiGAN decoder given the latent z (inter, R) and the harmonic source (N,)."""
    n_ups, n_rb, n_rbd = int(cfg["n_ups"]), int(cfg["n_rb"]), int(cfg["n_rbd"])
    rates = [int(cfg["up_rate%d" % i]) for i in range(n_ups)]
    kerns = [int(cfg["up_kernel%d" % i]) for i in range(n_ups)]
    g = _t(t["sy.g"])[None, :, None]
    with torch.no_grad():
        x = F.conv1d(_t(z)[None], _t(t["sy.dec.pre.w"]), _t(t["sy.dec.pre.b"]), padding=3) + F.conv1d(g, _t(t["sy.dec.cond.w"])[:, :, None], _t(t["sy.dec.cond.b"]))
        s = _t(src)[None, None]
        for i in range(n_ups):
            x = F.leaky_relu(x, 0.1)
            x = F.conv_transpose1d(x, _t(t["sy.dec.up%d.w" % i]), _t(t["sy.dec.up%d.b" % i]), stride=rates[i], padding=(kerns[i] - rates[i]) // 2)
            if i + 1 < n_ups:
                sf = int(np.prod(rates[i + 1:]))
                x = x + F.conv1d(s, _t(t["sy.dec.nc%d.w" % i]), _t(t["sy.dec.nc%d.b" % i]), stride=sf, padding=sf // 2)
            else:
                x = x + F.conv1d(s, _t(t["sy.dec.nc%d.w" % i]), _t(t["sy.dec.nc%d.b" % i]))
            xs = None
            for j in range(n_rb):
                k = int(cfg["rb_k%d" % j])
                r = x
                for m in range(n_rbd):
                    d = int(cfg["rb_d%d" % m])
                    xt = F.conv1d(F.leaky_relu(r, 0.1), _t(t["sy.dec.rb%d_%d.c1_%d.w" % (i, j, m)]), _t(t["sy.dec.rb%d_%d.c1_%d.b" % (i, j, m)]), dilation=d, padding=(k * d - d) // 2)
                    xt = F.conv1d(F.leaky_relu(xt, 0.1), _t(t["sy.dec.rb%d_%d.c2_%d.w" % (i, j, m)]), _t(t["sy.dec.rb%d_%d.c2_%d.b" % (i, j, m)]), padding=(k - 1) // 2)
                    r = xt + r
                xs = r if xs is None else xs + r
            x = xs / n_rb
        x = torch.tanh(F.conv1d(F.leaky_relu(x, 0.01), _t(t["sy.dec.post.w"]), None, padding=3))
    return x[0, 0].numpy()
