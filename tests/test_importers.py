"""Row f4: importing the files users of the reference have (ONNX graphs, PyTorch checkpoints, Faiss indexes).
The ContentVec name table is pinned against transformers.HubertModel end to end (import -> blob -> oracle forward vs the HF
forward); the synthesizer / RMVPE tables are exercised on state dicts rebuilt from the synthetic zoo in upstream naming
(weight-norm pairs, unfolded BatchNorm, speaker table) -- the names themselves are unpinned (no real file in this image)."""
import os

import numpy as np
import pytest

from common import rel_rms, voice_signal, zoo
from obs_rvc_amd import importers as IM
from obs_rvc_amd import onnx_reader as OR
from obs_rvc_amd import weights as W


def test_onnx_reader_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    inits = {"a.weight": rng.standard_normal((3, 4, 5)).astype(np.float32), "ids": np.arange(-3, 4, dtype=np.int64),
             "half": rng.standard_normal(7).astype(np.float16), "scalar": np.float32(3.5).reshape(()), "d": rng.standard_normal(3)}
    nodes = [{"op_type": "MatMul", "input": ["x", "a.weight"], "output": ["y"], "name": "mm"}, {"op_type": "Relu", "input": ["y"], "output": ["z"]}]
    for raw in (True, False):
        p = str(tmp_path / ("m%d.onnx" % raw))
        OR.write_onnx(p, inits, nodes, raw=raw)
        got, gn = OR.read_onnx(p)
        assert set(got) == set(inits) and all(got[k].shape == inits[k].shape and got[k].dtype == inits[k].dtype and np.array_equal(got[k], inits[k]) for k in inits)
        assert [n["op_type"] for n in gn] == ["MatMul", "Relu"] and gn[0]["input"] == ["x", "a.weight"] and gn[0]["name"] == "mm"
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad.onnx"), "wb").write(b"\x3a\xff\xff\xff\x0f")        # graph field with a length past the end
        OR.read_onnx(str(tmp_path / "bad.onnx"))


def _hf_model(embed=48, layers=2, heads=4, ffn=96, conv=32, pos_k=16, groups=4, seed=0):
    import torch
    from transformers import HubertConfig, HubertModel
    torch.manual_seed(seed)
    hc = HubertConfig(hidden_size=embed, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ffn, hidden_act="gelu",
                      hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0, final_dropout=0.0, layerdrop=0.0,
                      feat_proj_layer_norm=True, feat_extract_norm="group", feat_extract_activation="gelu", conv_dim=[conv] * 7,
                      conv_stride=list(W.CV_CONV_S), conv_kernel=list(W.CV_CONV_K), conv_bias=False, num_conv_pos_embeddings=pos_k,
                      num_conv_pos_embedding_groups=groups, do_stable_layer_norm=False, apply_spec_augment=False, layer_norm_eps=1e-5)
    m = HubertModel(hc).eval()
    with torch.no_grad():
        for p in m.parameters():          # default init leaves biases at zero and norms at one: perturb everything
            p.add_(0.05 * torch.randn_like(p))
    return m


def _oracle_hubert(tmp_path, cfg, tens, wav, version=2):
    from oracle import oracle as O
    d = tmp_path / ("data%d" % version)
    os.makedirs(d / "contentvec", exist_ok=True)
    W.write_blob(str(d / "contentvec" / W.cv_blob_name(version)), cfg, tens)
    ora = O.OracleRvcInfer(str(d)); ora.load_contentvec(version)
    return ora.hubert(wav)[0]


def test_contentvec_import_is_pinned_on_hf_hubert(tmp_path):
    import torch
    m = _hf_model()
    named = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    cfg, tens = IM.import_contentvec(named, version=2, heads=4, pos_groups=4)
    assert (cfg["embed"], cfg["conv_dim"], cfg["ffn"], cfg["layers"], cfg["run_layers"], cfg["pos_k"], cfg["out_dim"]) == (48, 32, 96, 2, 2, 16, 48)
    wav = voice_signal(12000, seed=2)
    got = _oracle_hubert(tmp_path, cfg, tens, wav)
    with torch.no_grad():
        ref = m(torch.from_numpy(wav)[None]).last_hidden_state[0].T.numpy()
    assert got.shape == ref.shape and rel_rms(got, ref) < 2e-5
    # the same weights under fairseq's names (and the old weight_g / weight_v spelling of the weight-normed positional conv)
    ren = {}
    for k, v in named.items():
        k2 = (k.replace(".conv.weight", ".0.weight") if k.startswith("feature_extractor") else k)
        k2 = k2.replace("feature_extractor.conv_layers.0.layer_norm", "feature_extractor.conv_layers.0.2")
        k2 = k2.replace("feature_projection.layer_norm", "layer_norm").replace("feature_projection.projection", "post_extract_proj")
        k2 = k2.replace("encoder.pos_conv_embed.conv.parametrizations.weight.original0", "encoder.pos_conv.0.weight_g")
        k2 = k2.replace("encoder.pos_conv_embed.conv.parametrizations.weight.original1", "encoder.pos_conv.0.weight_v")
        k2 = k2.replace("encoder.pos_conv_embed.conv.weight_g", "encoder.pos_conv.0.weight_g").replace("encoder.pos_conv_embed.conv.weight_v", "encoder.pos_conv.0.weight_v")
        k2 = k2.replace("encoder.pos_conv_embed.conv.bias", "encoder.pos_conv.0.bias")
        k2 = k2.replace(".attention.", ".self_attn.").replace("feed_forward.intermediate_dense", "fc1").replace("feed_forward.output_dense", "fc2")
        if ".layers." in k2 and ".layer_norm." in k2 and "final" not in k2:
            k2 = k2.replace(".layer_norm.", ".self_attn_layer_norm.")
        ren["model." + k2] = v
    cfg2, tens2 = IM.import_contentvec(ren, version=2, heads=4, pos_groups=4)
    assert cfg2 == cfg and all(np.array_equal(tens2[k], tens[k]) for k in tens)
    with pytest.raises(IM.ImportError_) as ei:
        IM.import_contentvec({k: v for k, v in named.items() if "layers.1.feed_forward" not in k}, heads=4, pos_groups=4)
    assert "ff1" in str(ei.value) or "intermediate_dense" in str(ei.value)


def test_contentvec_import_from_onnx_with_anonymous_linear_weights(tmp_path):
    m = _hf_model(seed=1)
    named = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    inits, nodes = {}, []
    for k, v in named.items():
        lin = k.endswith(".weight") and v.ndim == 2 and (k[:-7] + ".bias") in named
        if lin:                                  # exporter: y = MatMul(x, W^T) ; y = Add(y, bias)
            anon = "onnx::MatMul_%d" % len(nodes)
            inits[anon] = np.ascontiguousarray(v.T)
            nodes.append({"op_type": "MatMul", "input": ["t%d" % len(nodes), anon], "output": ["mm%d" % len(nodes)]})
            nodes.append({"op_type": "Add", "input": [k[:-7] + ".bias", nodes[-1]["output"][0]], "output": ["t%d" % len(nodes)]})
        else:
            inits[k] = v
    p = str(tmp_path / "vec-768-layer-12.onnx")
    OR.write_onnx(p, inits, nodes)
    cfg, tens = IM.import_contentvec(IM.load_named_tensors(p), heads=4, pos_groups=4)
    cfg0, tens0 = IM.import_contentvec(named, heads=4, pos_groups=4)
    assert cfg == cfg0 and all(np.array_equal(tens[k], tens0[k]) for k in tens0)
    out = str(tmp_path / "cv.rvcw")
    IM.main(["contentvec", p, out, "--version", "2"]) if False else IM.convert("contentvec", p, out, version=2, heads=4, pos_groups=4)
    c2, t2 = W.read_blob(out)
    assert int(c2["embed"]) == 48 and np.array_equal(t2["cv.l1.ff2.w"], tens0["cv.l1.ff2.w"])


def _wn_pair(w, rng):
    """weight -> (g, v) with w = g * v / |v| (weight_norm dim = 0) and a v that is NOT w."""
    v = w * rng.uniform(0.5, 2.0, size=(w.shape[0],) + (1,) * (w.ndim - 1)).astype(np.float32)
    g = np.sqrt((w.astype(np.float64) ** 2).sum(axis=tuple(range(1, w.ndim)), keepdims=True)).astype(np.float32)
    return g, v


def test_synth_import_round_trip_in_upstream_naming():
    cfg, t = W.read_blob(zoo("tiny")["model"])
    rng = np.random.default_rng(1)
    sd = {}
    emb = rng.standard_normal((4, int(cfg["gin"]))).astype(np.float32); emb[2] = t["sy.g"]
    sd["emb_g.weight"] = emb
    sd["enc_p.emb_phone.weight"], sd["enc_p.emb_phone.bias"], sd["enc_p.emb_pitch.weight"] = t["sy.enc.phone.w"], t["sy.enc.phone.b"], t["sy.enc.pitch_emb"]
    for i in range(int(cfg["enc_layers"])):
        a, q = "enc_p.encoder.attn_layers.%d." % i, "sy.enc.l%d." % i
        for n in "qkvo":
            sd[a + "conv_%s.weight" % n], sd[a + "conv_%s.bias" % n] = t[q + n + ".w"][:, :, None], t[q + n + ".b"]
        sd[a + "emb_rel_k"], sd[a + "emb_rel_v"] = t[q + "rel_k"][None], t[q + "rel_v"][None]
        for k, up in (("ln1", "norm_layers_1"), ("ln2", "norm_layers_2")):
            sd["enc_p.encoder.%s.%d.gamma" % (up, i)], sd["enc_p.encoder.%s.%d.beta" % (up, i)] = t[q + k + ".g"], t[q + k + ".b"]
        for k, up in (("ff1", "conv_1"), ("ff2", "conv_2")):
            sd["enc_p.encoder.ffn_layers.%d.%s.weight" % (i, up)], sd["enc_p.encoder.ffn_layers.%d.%s.bias" % (i, up)] = t[q + k + ".w"], t[q + k + ".b"]
    sd["enc_p.proj.weight"], sd["enc_p.proj.bias"] = t["sy.enc.proj.w"][:, :, None], t["sy.enc.proj.b"]

    def wn(dst, w):
        sd[dst + ".weight_g"], sd[dst + ".weight_v"] = _wn_pair(w, rng)

    for i in range(int(cfg["flow_n"])):
        f, q = "flow.flows.%d." % (2 * i), "sy.flow%d." % i
        sd[f + "pre.weight"], sd[f + "pre.bias"] = t[q + "pre.w"][:, :, None], t[q + "pre.b"]
        wn(f + "enc.cond_layer", t[q + "cond.w"][:, :, None]); sd[f + "enc.cond_layer.bias"] = t[q + "cond.b"]
        for j in range(int(cfg["wn_layers"])):
            wn(f + "enc.in_layers.%d" % j, t[q + "in%d.w" % j]); sd[f + "enc.in_layers.%d.bias" % j] = t[q + "in%d.b" % j]
            wn(f + "enc.res_skip_layers.%d" % j, t[q + "rs%d.w" % j][:, :, None]); sd[f + "enc.res_skip_layers.%d.bias" % j] = t[q + "rs%d.b" % j]
        sd[f + "post.weight"], sd[f + "post.bias"] = t[q + "post.w"][:, :, None], t[q + "post.b"]
    sd["dec.conv_pre.weight"], sd["dec.conv_pre.bias"] = t["sy.dec.pre.w"], t["sy.dec.pre.b"]
    sd["dec.cond.weight"], sd["dec.cond.bias"] = t["sy.dec.cond.w"][:, :, None], t["sy.dec.cond.b"]
    sd["dec.m_source.l_linear.weight"], sd["dec.m_source.l_linear.bias"] = t["sy.src"][:1].reshape(1, 1), t["sy.src"][1:]
    n_rb, n_rbd = int(cfg["n_rb"]), int(cfg["n_rbd"])
    for i in range(int(cfg["n_ups"])):
        wn("dec.ups.%d" % i, t["sy.dec.up%d.w" % i]); sd["dec.ups.%d.bias" % i] = t["sy.dec.up%d.b" % i]
        sd["dec.noise_convs.%d.weight" % i], sd["dec.noise_convs.%d.bias" % i] = t["sy.dec.nc%d.w" % i], t["sy.dec.nc%d.b" % i]
        for j in range(n_rb):
            for m in range(n_rbd):
                r, q = "dec.resblocks.%d." % (i * n_rb + j), "sy.dec.rb%d_%d." % (i, j)
                wn(r + "convs1.%d" % m, t[q + "c1_%d.w" % m]); sd[r + "convs1.%d.bias" % m] = t[q + "c1_%d.b" % m]
                wn(r + "convs2.%d" % m, t[q + "c2_%d.w" % m]); sd[r + "convs2.%d.bias" % m] = t[q + "c2_%d.b" % m]
    sd["dec.conv_post.weight"] = t["sy.dec.post.w"]
    # upsample rates are recovered from the noise-conv kernels except the first one (kernel = 2 * rate is a HiFiGAN convention
    # the tiny preset does not follow): pass the rates, check the inference on the noise convs separately
    rates = [int(cfg["up_rate%d" % i]) for i in range(int(cfg["n_ups"]))]
    c2, t2 = IM.import_synth({"weight." + k: v for k, v in sd.items()}, sid=2, sr=int(cfg["sr"]), up_rates=rates, heads=int(cfg["heads"]))
    assert set(t2) == set(t)
    for k in t:
        assert t2[k].shape == t[k].shape and np.allclose(t2[k], t[k], rtol=2e-6, atol=1e-7), k
    for k in cfg:
        assert float(c2[k]) == float(cfg[k]), k
    # without explicit rates the first one follows from sr (the noise convs pin the others); with nothing to go on the import fails
    c3, _ = IM.import_synth(sd, sid=2, sr=int(cfg["sr"]))
    assert [c3["up_rate%d" % i] for i in range(4)] == rates and c3["sr"] == cfg["sr"]
    with pytest.raises(IM.ImportError_):
        IM.import_synth(sd, sid=2)
    with pytest.raises(IM.ImportError_):
        IM.import_synth(sd, sid=2, sr=int(cfg["sr"]) * 2, up_rates=rates)       # sr must be 100 * prod(rates)
    with pytest.raises(IM.ImportError_):
        IM.import_synth({k: v for k, v in sd.items() if "resblocks.3.convs2.0" not in k}, sid=0)


def test_rmvpe_import_folds_batchnorm():
    cfg, t = W.read_blob(os.path.join(zoo("tiny")["data"], "f0", "rmvpe.rvcw"))
    rng = np.random.default_rng(2)
    sd = {}

    def unfold(dst_conv, dst_bn, w, b, out_axis=0):
        """(folded w, b) -> raw conv weight + BatchNorm statistics that fold back to them."""
        co = w.shape[out_axis]
        gamma, var = rng.uniform(0.5, 1.5, co).astype(np.float32), rng.uniform(0.5, 2.0, co).astype(np.float32)
        s = gamma / np.sqrt(var + 1e-5)
        shape = [1] * w.ndim; shape[out_axis] = -1
        mean = rng.standard_normal(co).astype(np.float32) * 0.1
        sd[dst_conv + ".weight"] = (w / s.reshape(shape)).astype(np.float32)
        sd[dst_bn + ".weight"], sd[dst_bn + ".running_var"], sd[dst_bn + ".running_mean"] = gamma, var, mean
        sd[dst_bn + ".bias"] = (b + mean * s).astype(np.float32)

    def block(src, dst):
        unfold(dst + ".conv.0", dst + ".conv.1", t[src + "c1.w"], t[src + "c1.b"]); unfold(dst + ".conv.3", dst + ".conv.4", t[src + "c2.w"], t[src + "c2.b"])
        if src + "sc.w" in t:
            sd[dst + ".shortcut.weight"], sd[dst + ".shortcut.bias"] = t[src + "sc.w"][:, :, None, None], t[src + "sc.b"]

    sc, sh = t["rm.bn0"]
    sd["unet.encoder.bn.weight"], sd["unet.encoder.bn.running_var"] = np.float32([sc * 2.0]), np.float32([4.0 - 1e-5])
    sd["unet.encoder.bn.running_mean"], sd["unet.encoder.bn.bias"] = np.float32([0.3]), np.float32([sh + 0.3 * sc])
    L, nb = int(cfg["levels"]), int(cfg["n_blocks"])
    for lv in range(L):
        for j in range(nb):
            block("rm.enc%d.b%d." % (lv, j), "unet.encoder.layers.%d.conv.%d" % (lv, j))
    for lv in range(int(cfg["inter_layers"])):
        for j in range(nb):
            block("rm.int%d.b%d." % (lv, j), "unet.intermediate.layers.%d.conv.%d" % (lv, j))
    for lv in range(L):
        unfold("unet.decoder.layers.%d.conv1.0" % lv, "unet.decoder.layers.%d.conv1.1" % lv, t["rm.dec%d.up.w" % lv], t["rm.dec%d.up.b" % lv], out_axis=1)
        for j in range(nb):
            block("rm.dec%d.b%d." % (lv, j), "unet.decoder.layers.%d.conv2.%d" % (lv, j))
    sd["cnn.weight"], sd["cnn.bias"] = t["rm.cnn.w"], t["rm.cnn.b"]
    for d, suf in (("f", ""), ("b", "_reverse")):
        for a, b in (("w_ih_", "weight_ih_l0"), ("w_hh_", "weight_hh_l0"), ("b_ih_", "bias_ih_l0"), ("b_hh_", "bias_hh_l0")):
            sd["fc.0.gru." + b + suf] = t["rm.gru." + a + d]
    sd["fc.1.weight"], sd["fc.1.bias"] = t["rm.fc.w"], t["rm.fc.b"]
    c2, t2 = IM.import_rmvpe(sd)
    assert set(t2) == set(t)
    for k in t:
        assert t2[k].shape == t[k].shape and np.allclose(t2[k], t[k], rtol=3e-5, atol=2e-6), k
    for k in cfg:
        assert float(c2[k]) == float(cfg[k]), k


def test_faiss_index_reader(tmp_path):
    from obs_rvc_amd import faiss_index as FI
    rng = np.random.default_rng(3)
    v = rng.standard_normal((500, 24)).astype(np.float32)
    p = str(tmp_path / "flat.index")
    FI.write_flat(p, v)
    assert np.array_equal(FI.read_index(p), v)
    cents = v[rng.choice(500, 16, replace=False)] + 0.01
    cents[5] = 100.0                                     # an empty list
    for sparse in (False, True):
        q = str(tmp_path / ("ivf%d.index" % sparse))
        FI.write_ivf_flat(q, v, cents, sparse_sizes=sparse)
        got = FI.read_index(q)
        assert got.dtype == np.float32 and np.array_equal(got, v)      # back in id order although stored list by list
    raw = open(p, "rb").read()
    assert raw[:4] == b"IxF2" and len(raw) == 4 + (4 + 8 + 8 + 8 + 1 + 4) + 8 + v.nbytes
    open(str(tmp_path / "trunc.index"), "wb").write(raw[:-10])
    with pytest.raises(FI.IndexFormatError):
        FI.read_index(str(tmp_path / "trunc.index"))
    open(str(tmp_path / "pq.index"), "wb").write(b"IxPq" + raw[4:])
    with pytest.raises(FI.IndexFormatError):
        FI.read_index(str(tmp_path / "pq.index"))


def test_faiss_reader_against_independent_fixture():
    # tests/golden/faiss_*.index are written by tests/golden/make_faiss_fixture.c, a separate C restatement of faiss's
    # index_write.cpp macro sequence (IndexFlatL2; IndexIVFFlat with "full" and with "sprs" list sizes, ids stored out of order,
    # and -- unlike IwSq / IwPQ -- no code_size field between the direct map and the inverted lists)
    from obs_rvc_amd import faiss_index as FI
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    exp = np.array([[(i * 7 + j * 3) % 11 - 5 + 0.25 * j for j in range(4)] for i in range(7)], np.float32)
    for name in ("faiss_flat.index", "faiss_ivf.index", "faiss_ivf_sparse.index"):
        got = FI.read_index(os.path.join(gold, name))
        assert got.dtype == np.float32 and np.array_equal(got, exp), name
    raw = open(os.path.join(gold, "faiss_ivf.index"), "rb").read()
    # byte-level: header (33) + nlist, nprobe (16) + quantizer IxF2 (4 + 33 + 8 + 3*4*4) + direct map (1 + 8) + "ilar" nlist code_size (20)
    # + "full" + sizes vector (4 + 8 + 24) + 7 * (16 + 8)
    assert len(raw) == 4 + 33 + 16 + (4 + 33 + 8 + 48) + 9 + 20 + 36 + 7 * 24
    assert raw[4 + 33 + 16 + 93 + 9:][:4] == b"ilar"


def test_synth_import_of_official_rate_kernel_pairs(tmp_path):
    # 40 kHz v1/v2 models pair upsample kernels [16, 16, 4, 4] with rates [10, 10, 2, 2] (NOT kernel = 2 * rate): the rates must
    # come from the checkpoint's config list (entries 12 / 14 / 17) or from sr, never from the kernel size
    import torch
    m = _upstream_synth(up_init=32, rates=(10, 10, 2, 2), kernels=(16, 16, 4, 4))
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    config = [1025, 32, 16, 16, 32, 2, 2, 3, 0.0, "1", [3, 5], [[1, 3], [1, 3]], [10, 10, 2, 2], 32, [16, 16, 4, 4], 3, 8, 40000]
    c1, _ = IM.import_synth(sd, sid=0, config=config)
    assert [c1["up_rate%d" % i] for i in range(4)] == [10, 10, 2, 2] and c1["sr"] == 40000 and c1["heads"] == 2
    c2, _ = IM.import_synth(sd, sid=0, sr=40000)
    assert [c2["up_rate%d" % i] for i in range(4)] == [10, 10, 2, 2]
    with pytest.raises(IM.ImportError_):
        IM.import_synth(sd, sid=0)                                   # nothing pins the first rate
    with pytest.raises(IM.ImportError_):
        IM.import_synth(sd, sid=0, up_rates=[8, 10, 2, 2], sr=40000)  # (16 - 8) / 2 is a valid padding, but sr is not 100 * prod
    # the same through the files users have: {"weight": sd, "config": [...]} + the CLI, no --sr / --up-rates given
    pth, out = str(tmp_path / "voice40k.pth"), str(tmp_path / "voice40k.rvcw")
    torch.save({"weight": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, "config": config, "version": "v2", "sr": "40k"}, pth)
    assert IM.load_checkpoint_config(pth)[12] == [10, 10, 2, 2]
    IM.main(["synth", pth, out])
    cfg, _ = W.read_blob(out)
    assert [int(cfg["up_rate%d" % i]) for i in range(4)] == [10, 10, 2, 2] and int(cfg["sr"]) == 40000
    # a bare state dict (no config entry) needs --sr or --up-rates
    bare = str(tmp_path / "bare.pth")
    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}, bare)
    with pytest.raises(IM.ImportError_):
        IM.main(["synth", bare, out])
    IM.main(["synth", bare, out, "--up-rates", "10,10,2,2"])
    assert int(W.read_blob(out)[0]["sr"]) == 40000


def test_checkpoint_files_and_cli(tmp_path):
    # the three container formats load_named_tensors accepts (.pth as RVC ships it: {"weight": state_dict, "config": [...]},
    # .safetensors, .onnx) and the command-line entry point
    import torch
    from safetensors.numpy import save_file
    m = _hf_model(seed=2)
    named = {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}
    pth = str(tmp_path / "hubert.pth")
    torch.save({"weight": {k: torch.from_numpy(v) for k, v in named.items()}, "config": [1, 2, 3], "version": "v2"}, pth)
    sft = str(tmp_path / "hubert.safetensors")
    save_file(named, sft)
    a, b = IM.load_named_tensors(pth), IM.load_named_tensors(sft)
    assert set(a) == set(b) == set(named) and all(np.array_equal(a[k], named[k]) and np.array_equal(b[k], named[k]) for k in named)
    # half-precision checkpoints are widened to f32
    torch.save({k: torch.from_numpy(v).half() for k, v in named.items()}, str(tmp_path / "half.pt"))
    h = IM.load_named_tensors(str(tmp_path / "half.pt"))
    assert all(v.dtype == np.float32 for v in h.values())
    # CLI: rmvpe from a .pth written in upstream naming (reuse the synthetic zoo through the exporter of the test above is overkill:
    # the ContentVec route exercises the same main())
    out = str(tmp_path / "vec-768-layer-12.rvcw")
    with pytest.raises(SystemExit):
        IM.main(["contentvec"])                               # argparse: missing paths
    onnx = str(tmp_path / "cv.onnx")
    OR.write_onnx(onnx, named)
    with pytest.raises(IM.ImportError_):
        IM.main(["contentvec", onnx, out, "--version", "2"])  # default pos_groups 16 does not fit this 48-wide toy model
    cfg, t = IM.import_contentvec(IM.load_named_tensors(onnx), heads=4, pos_groups=4)
    assert cfg["embed"] == 48 and "cv.l1.ff2.w" in t


def _upstream_rmvpe(en_out=4, n_blocks=2, levels=5, inter=2, hidden=32, seed=0):
    """RMVPE's E2E network written as upstream structures it (module and parameter names as in its rmvpe.py, from the public
    architecture): DeepUnet(Encoder / Intermediate / Decoder of ConvBlockRes) -> Conv2d(.., 3) -> BiGRU -> Linear -> Sigmoid."""
    import torch
    import torch.nn as nn

    class ConvBlockRes(nn.Module):
        def __init__(self, ci, co):
            super().__init__()
            self.conv = nn.Sequential(nn.Conv2d(ci, co, 3, 1, 1, bias=False), nn.BatchNorm2d(co, momentum=0.01), nn.ReLU(),
                                      nn.Conv2d(co, co, 3, 1, 1, bias=False), nn.BatchNorm2d(co, momentum=0.01), nn.ReLU())
            self.is_shortcut = ci != co
            if self.is_shortcut:
                self.shortcut = nn.Conv2d(ci, co, 1)

        def forward(self, x):
            return self.conv(x) + (self.shortcut(x) if self.is_shortcut else x)

    class ResEncoderBlock(nn.Module):
        def __init__(self, ci, co, pool, nb):
            super().__init__()
            self.conv = nn.ModuleList([ConvBlockRes(ci, co)] + [ConvBlockRes(co, co) for _ in range(nb - 1)])
            self.pool = nn.AvgPool2d(2) if pool else None

        def forward(self, x):
            for c in self.conv:
                x = c(x)
            return (x, self.pool(x)) if self.pool is not None else x

    class Encoder(nn.Module):
        def __init__(self):
            super().__init__()
            self.bn = nn.BatchNorm2d(1, momentum=0.01)
            self.layers = nn.ModuleList()
            ci, co = 1, en_out
            for _ in range(levels):
                self.layers.append(ResEncoderBlock(ci, co, True, n_blocks)); ci, co = co, co * 2
            self.out_channel = co

        def forward(self, x):
            skips = []
            x = self.bn(x)
            for l in self.layers:
                t, x = l(x); skips.append(t)
            return x, skips

    class Intermediate(nn.Module):
        def __init__(self, ci, co):
            super().__init__()
            self.layers = nn.ModuleList([ResEncoderBlock(ci, co, False, n_blocks)] + [ResEncoderBlock(co, co, False, n_blocks) for _ in range(inter - 1)])

        def forward(self, x):
            for l in self.layers:
                x = l(x)
            return x

    class ResDecoderBlock(nn.Module):
        def __init__(self, ci, co):
            super().__init__()
            self.conv1 = nn.Sequential(nn.ConvTranspose2d(ci, co, 3, 2, 1, 1, bias=False), nn.BatchNorm2d(co, momentum=0.01), nn.ReLU())
            self.conv2 = nn.ModuleList([ConvBlockRes(co * 2, co)] + [ConvBlockRes(co, co) for _ in range(n_blocks - 1)])

        def forward(self, x, skip):
            x = torch.cat((self.conv1(x), skip), dim=1)
            for c in self.conv2:
                x = c(x)
            return x

    class Decoder(nn.Module):
        def __init__(self, ci):
            super().__init__()
            self.layers = nn.ModuleList()
            for _ in range(levels):
                self.layers.append(ResDecoderBlock(ci, ci // 2)); ci //= 2

        def forward(self, x, skips):
            for i, l in enumerate(self.layers):
                x = l(x, skips[-1 - i])
            return x

    class DeepUnet(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = Encoder()
            self.intermediate = Intermediate(self.encoder.out_channel // 2, self.encoder.out_channel)
            self.decoder = Decoder(self.encoder.out_channel)

        def forward(self, x):
            x, skips = self.encoder(x)
            return self.decoder(self.intermediate(x), skips)

    class BiGRU(nn.Module):
        def __init__(self, i, h):
            super().__init__()
            self.gru = nn.GRU(i, h, num_layers=1, batch_first=True, bidirectional=True)

        def forward(self, x):
            return self.gru(x)[0]

    class E2E(nn.Module):
        def __init__(self):
            super().__init__()
            self.unet = DeepUnet()
            self.cnn = nn.Conv2d(en_out, 3, 3, padding=1)
            self.fc = nn.Sequential(BiGRU(3 * 128, hidden), nn.Linear(2 * hidden, 360), nn.Dropout(0.25), nn.Sigmoid())

        def forward(self, mel):                       # mel (B, 128, T)
            x = self.cnn(self.unet(mel.transpose(-1, -2).unsqueeze(1))).transpose(1, 2).flatten(-2)
            return self.fc(x)                         # (B, T, 360)

    torch.manual_seed(seed)
    m = E2E().eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm2d):       # non-trivial running statistics and affine
                mod.running_mean.normal_(0, 0.2); mod.running_var.uniform_(0.5, 1.5); mod.weight.uniform_(0.7, 1.3); mod.bias.normal_(0, 0.1)
        m.unet.encoder.bn.running_mean.fill_(-4.0); m.unet.encoder.bn.running_var.fill_(9.0)   # log-mel is roughly in [-12, 2]
    return m


def test_rmvpe_import_is_checked_against_an_upstream_structured_module(tmp_path):
    # import_rmvpe (names, BatchNorm folding, concat order, GRU gate layout) checked end to end: an nn.Module with upstream's
    # structure and parameter names -> state dict -> blob -> oracle forward  ==  the module's own forward on the oracle's log-mel
    import torch
    from oracle import oracle as O
    m = _upstream_rmvpe()
    named = {k: v.detach().numpy() for k, v in m.state_dict().items() if "num_batches_tracked" not in k}
    cfg, tens = IM.import_rmvpe(named)
    assert (cfg["en_out"], cfg["levels"], cfg["n_blocks"], cfg["inter_layers"], cfg["gru_hidden"], cfg["n_out"]) == (4, 5, 2, 2, 32, 360)
    d = tmp_path / "data"
    os.makedirs(d / "f0"); os.makedirs(d / "contentvec")
    W.write_blob(str(d / "f0" / "rmvpe.rvcw"), cfg, tens)
    ora = O.OracleRvcInfer(str(d)); ora.load_f0(1); ora.enable_taps(True)
    try:
        ora.pitch(voice_signal(35840, seed=4), 0, 2560)
    except Exception as ex:                            # an untrained head may decode to a bin where the reference panics: taps are still there
        assert "Panic" in str(ex)
    mel = ora.tap("rm.mel").reshape(128, 32)
    got = ora.tap("rm.sal").reshape(32, 360)
    with torch.no_grad():
        ref = m(torch.from_numpy(np.ascontiguousarray(mel))[None])[0].numpy()
    assert ref.shape == got.shape and rel_rms(got, ref) < 5e-5, rel_rms(got, ref)


def _upstream_synth(phone_dim=48, inter=16, hidden=16, filt=32, heads=2, layers=2, k=3, window=4, n_flows=2, wn_layers=2, gin=8,
                    up_init=32, rates=(4, 3, 2, 2), kernels=(8, 7, 4, 4), rb_k=(3, 5), rb_d=(1, 3), n_spk=3, seed=0):
    """RVC's SynthesizerTrnMs768NSFsid written as upstream structures it (module / parameter names of infer_pack's models.py,
    attentions.py, modules.py, from the public architecture; the relative-position attention is written directly from its definition).
    Returns (module, helpers) where the decoder takes the harmonic source as an input (upstream draws it from SineGen)."""
    import math
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from torch.nn.utils import weight_norm

    class LayerNorm(nn.Module):
        def __init__(self, c):
            super().__init__()
            self.gamma, self.beta = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))

        def forward(self, x):
            return F.layer_norm(x.transpose(1, -1), (x.shape[1],), self.gamma, self.beta, 1e-5).transpose(1, -1)

    class MultiHeadAttention(nn.Module):
        def __init__(self, c, n_heads, window_size):
            super().__init__()
            self.n_heads, self.k_channels, self.w = n_heads, c // n_heads, window_size
            self.conv_q, self.conv_k, self.conv_v, self.conv_o = (nn.Conv1d(c, c, 1) for _ in range(4))
            self.emb_rel_k = nn.Parameter(torch.randn(1, 2 * window_size + 1, self.k_channels) * self.k_channels ** -0.5)
            self.emb_rel_v = nn.Parameter(torch.randn(1, 2 * window_size + 1, self.k_channels) * self.k_channels ** -0.5)

        def forward(self, x, c):
            b, d, t = x.shape
            q, kk, v = (f(x).view(b, self.n_heads, self.k_channels, t).transpose(2, 3) for f in (self.conv_q, self.conv_k, self.conv_v))
            q = q / math.sqrt(self.k_channels)
            scores = q @ kk.transpose(-2, -1)
            rel = torch.arange(t)[None, :] - torch.arange(t)[:, None]                 # j - i
            inside = (rel.abs() <= self.w)
            idx = (rel + self.w).clamp(0, 2 * self.w)
            rk, rv = self.emb_rel_k[0][idx], self.emb_rel_v[0][idx]                       # (t, t, kc)
            scores = scores + torch.einsum("bhid,ijd->bhij", q, rk) * inside
            p = F.softmax(scores, dim=-1)
            out = p @ v + torch.einsum("bhij,ijd->bhid", p * inside, rv)
            return self.conv_o(out.transpose(2, 3).contiguous().view(b, d, t))

    class FFN(nn.Module):
        def __init__(self, c, f, ks):
            super().__init__()
            self.ks = ks
            self.conv_1, self.conv_2 = nn.Conv1d(c, f, ks), nn.Conv1d(f, c, ks)

        def forward(self, x, mask):
            pad = ((self.ks - 1) // 2, self.ks // 2)
            x = torch.relu(self.conv_1(F.pad(x * mask, pad)))
            return self.conv_2(F.pad(x * mask, pad)) * mask

    class Encoder(nn.Module):
        def __init__(self):
            super().__init__()
            self.attn_layers = nn.ModuleList(MultiHeadAttention(hidden, heads, window) for _ in range(layers))
            self.norm_layers_1 = nn.ModuleList(LayerNorm(hidden) for _ in range(layers))
            self.ffn_layers = nn.ModuleList(FFN(hidden, filt, k) for _ in range(layers))
            self.norm_layers_2 = nn.ModuleList(LayerNorm(hidden) for _ in range(layers))

        def forward(self, x, mask):
            x = x * mask
            for a, n1, f, n2 in zip(self.attn_layers, self.norm_layers_1, self.ffn_layers, self.norm_layers_2):
                x = n1(x + a(x, x))
                x = n2(x + f(x, mask))
            return x * mask

    class TextEncoder(nn.Module):
        def __init__(self):
            super().__init__()
            self.emb_phone, self.emb_pitch = nn.Linear(phone_dim, hidden), nn.Embedding(256, hidden)
            self.encoder, self.proj = Encoder(), nn.Conv1d(hidden, inter * 2, 1)

        def forward(self, phone, pitch):
            x = F.leaky_relu((self.emb_phone(phone) + self.emb_pitch(pitch)) * math.sqrt(hidden), 0.1).transpose(1, -1)
            mask = torch.ones(1, 1, x.shape[-1])
            x = self.encoder(x, mask)
            return torch.split(self.proj(x) * mask, inter, dim=1)

    class WN(nn.Module):
        def __init__(self, ks):
            super().__init__()
            self.in_layers = nn.ModuleList(weight_norm(nn.Conv1d(hidden, 2 * hidden, ks, padding=(ks - 1) // 2)) for _ in range(wn_layers))
            self.res_skip_layers = nn.ModuleList(weight_norm(nn.Conv1d(hidden, 2 * hidden if i < wn_layers - 1 else hidden, 1)) for i in range(wn_layers))
            self.cond_layer = weight_norm(nn.Conv1d(gin, 2 * hidden * wn_layers, 1))

        def forward(self, x, mask, g):
            out = torch.zeros_like(x)
            g = self.cond_layer(g)
            for i in range(wn_layers):
                a = self.in_layers[i](x) + g[:, i * 2 * hidden:(i + 1) * 2 * hidden]
                acts = torch.tanh(a[:, :hidden]) * torch.sigmoid(a[:, hidden:])
                rs = self.res_skip_layers[i](acts)
                if i < wn_layers - 1:
                    x = (x + rs[:, :hidden]) * mask; out = out + rs[:, hidden:]
                else:
                    out = out + rs
            return out * mask

    class ResidualCouplingLayer(nn.Module):
        def __init__(self):
            super().__init__()
            self.pre, self.enc, self.post = nn.Conv1d(inter // 2, hidden, 1), WN(5), nn.Conv1d(hidden, inter // 2, 1)

        def forward(self, x, mask, g, reverse):
            x0, x1 = torch.split(x, [inter // 2] * 2, 1)
            m = self.post(self.enc(self.pre(x0) * mask, mask, g)) * mask
            x1 = (x1 - m) * mask if reverse else m + x1 * mask
            return torch.cat([x0, x1], 1)

    class Flip(nn.Module):
        def forward(self, x, mask, g, reverse):
            return torch.flip(x, [1])

    class ResidualCouplingBlock(nn.Module):
        def __init__(self):
            super().__init__()
            self.flows = nn.ModuleList()
            for _ in range(n_flows):
                self.flows.append(ResidualCouplingLayer()); self.flows.append(Flip())

        def forward(self, x, mask, g, reverse=True):
            for f in (reversed(self.flows) if reverse else self.flows):
                x = f(x, mask, g, reverse)
            return x

    class ResBlock1(nn.Module):
        def __init__(self, ch, ks):
            super().__init__()
            self.convs1 = nn.ModuleList(weight_norm(nn.Conv1d(ch, ch, ks, 1, dilation=d, padding=(ks * d - d) // 2)) for d in rb_d)
            self.convs2 = nn.ModuleList(weight_norm(nn.Conv1d(ch, ch, ks, 1, dilation=1, padding=(ks - 1) // 2)) for _ in rb_d)

        def forward(self, x):
            for c1, c2 in zip(self.convs1, self.convs2):
                x = c2(F.leaky_relu(c1(F.leaky_relu(x, 0.1)), 0.1)) + x
            return x

    class SourceModuleHnNSF(nn.Module):
        def __init__(self):
            super().__init__()
            self.l_linear = nn.Linear(1, 1)

    class GeneratorNSF(nn.Module):
        def __init__(self):
            super().__init__()
            self.m_source = SourceModuleHnNSF()
            self.conv_pre = nn.Conv1d(inter, up_init, 7, 1, padding=3)
            self.ups, self.noise_convs, self.resblocks = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
            ch = up_init
            for i, (u, kk) in enumerate(zip(rates, kernels)):
                self.ups.append(weight_norm(nn.ConvTranspose1d(ch, ch // 2, kk, u, padding=(kk - u) // 2)))
                ch //= 2
                if i + 1 < len(rates):
                    sf = int(np.prod(rates[i + 1:]))
                    self.noise_convs.append(nn.Conv1d(1, ch, kernel_size=sf * 2, stride=sf, padding=sf // 2))
                else:
                    self.noise_convs.append(nn.Conv1d(1, ch, kernel_size=1))
                for ks in rb_k:
                    self.resblocks.append(ResBlock1(ch, ks))
            self.conv_post = nn.Conv1d(ch, 1, 7, 1, padding=3, bias=False)
            self.cond = nn.Conv1d(gin, up_init, 1)

        def forward(self, x, har_source, g):           # har_source (1, 1, T * upp): upstream computes it as m_source(f0, upp)
            x = self.conv_pre(x) + self.cond(g)
            nk = len(rb_k)
            for i in range(len(rates)):
                x = self.ups[i](F.leaky_relu(x, 0.1)) + self.noise_convs[i](har_source)
                x = sum(self.resblocks[i * nk + j](x) for j in range(nk)) / nk
            return torch.tanh(self.conv_post(F.leaky_relu(x)))

    class Synth(nn.Module):
        def __init__(self):
            super().__init__()
            self.enc_p, self.dec, self.flow, self.emb_g = TextEncoder(), GeneratorNSF(), ResidualCouplingBlock(), nn.Embedding(n_spk, gin)

    torch.manual_seed(seed)
    m = Synth().eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    return m


def test_synth_import_is_checked_against_an_upstream_structured_module(tmp_path):
    # import_synth (names, weight-norm folding, baked speaker row, flow order and flips, rel-pos tables, noise-conv geometry) checked
    # end to end: upstream-structured nn.Module -> state dict -> blob -> oracle forward == the module's forward, stage by stage, on
    # the oracle's own phone / pitch / prior noise / harmonic source
    import torch
    from oracle import oracle as O
    m = _upstream_synth()
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    assert any(k.endswith("weight_g") or "parametrizations" in k for k in sd)          # weight-normed layers are exported as pairs
    cfg, tens = IM.import_synth(sd, sid=1, sr=4800, up_rates=[4, 3, 2, 2], heads=2)
    model = str(tmp_path / "upstream.rvcw")
    W.write_blob(model, cfg, tens)
    z = zoo("tiny")
    ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(2); ora.load_f0(1); ora.load_model(model); ora.set_noise_seed(3, 1); ora.enable_taps(True)
    R = 21
    audio = ora.infer(voice_signal(35840, seed=4), 2560, 12, 200, R)
    phone = torch.from_numpy(np.ascontiguousarray(ora.tap("phone").reshape(R, -1)))[None]
    pitch = torch.from_numpy(ora.tap("pitch").astype(np.int64))[None]
    I = 16
    with torch.no_grad():
        g = m.emb_g(torch.tensor([1])).unsqueeze(-1)
        mp, logs = m.enc_p(phone, pitch)
        stats = torch.cat([mp, logs], 1)[0].numpy()
        assert rel_rms(ora.tap("sy.stats").reshape(stats.shape), stats) < 5e-5
        eps = torch.from_numpy(O.philox_normal(3, 1, 0, 0, I * R).reshape(I, R))[None]
        zp = mp + torch.exp(logs) * eps * 0.66666
        mask = torch.ones(1, 1, R)
        zz = m.flow(zp, mask, g, reverse=True)
        assert rel_rms(ora.tap("sy.z").reshape(I, R), zz[0].numpy()) < 5e-5
        src = torch.from_numpy(np.ascontiguousarray(ora.tap("sy.src")).reshape(1, 1, -1))
        ref = m.dec(zz, src, g)[0, 0].numpy()
    assert ref.shape == audio.shape and np.sqrt(np.mean((ref - audio) ** 2)) < 5e-5


def _anonymised_rmvpe_onnx(path, m, fold_bn=True):
    """Write the graph a torch.onnx export of RMVPE `E2E` produces, as far as the importer can see it: nodes in execution order,
    every Conv + BatchNorm pair folded into one Conv with anonymous operands (`onnx::Conv_<n>`) when fold_bn, the GRU packed as
    ONNX packs it (W / R / B, gate order z, r, h, directions stacked), the Linear as MatMul(x, W^T) + Add."""
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    inits, nodes, cnt = {}, [], [0]

    def fresh(kind):
        cnt[0] += 1
        return "onnx::%s_%d" % (kind, 100 + 7 * cnt[0])

    def bn_params(prefix):
        return [sd[prefix + k] for k in ("weight", "bias", "running_mean", "running_var")]

    def conv(x, wname, bnprefix=None, bias=None, transpose=False):
        w = sd[wname]
        op = "ConvTranspose" if transpose else "Conv"
        if bnprefix is not None and fold_bn:
            g, b, mu, var = bn_params(bnprefix)
            sc = g / np.sqrt(var + 1e-5)
            shape = [1, -1, 1, 1] if transpose else [-1, 1, 1, 1]
            wf, bf = w * sc.reshape(shape), b + ((sd[bias] if bias else 0.0) - mu) * sc
            wn, bn_ = fresh(op), fresh(op)
            inits[wn], inits[bn_] = wf.astype(np.float32), bf.astype(np.float32)
            y = "t%d" % len(nodes)
            nodes.append({"op_type": op, "input": [x, wn, bn_], "output": [y]})
            return y
        wn = fresh(op)
        inits[wn] = w
        ins = [x, wn]
        if bias:
            bn_ = fresh(op); inits[bn_] = sd[bias]; ins.append(bn_)
        y = "t%d" % len(nodes)
        nodes.append({"op_type": op, "input": ins, "output": [y]})
        if bnprefix is not None:
            names = []
            for k, arr in zip("gbmv", bn_params(bnprefix)):
                n = fresh("BatchNormalization"); inits[n] = arr; names.append(n)
            y2 = "t%d" % len(nodes)
            nodes.append({"op_type": "BatchNormalization", "input": [y] + names, "output": [y2]})
            y = y2
        return y

    def relu(x):
        y = "t%d" % len(nodes); nodes.append({"op_type": "Relu", "input": [x], "output": [y]}); return y

    def block(x, prefix):
        y = relu(conv(x, prefix + ".conv.0.weight", prefix + ".conv.1."))
        y = relu(conv(y, prefix + ".conv.3.weight", prefix + ".conv.4."))
        if prefix + ".shortcut.weight" in sd:
            sc = conv(x, prefix + ".shortcut.weight", None, prefix + ".shortcut.bias")
        else:
            sc = x
        z = "t%d" % len(nodes); nodes.append({"op_type": "Add", "input": [y, sc], "output": [z]}); return z

    names = []
    for k, arr in zip("gbmv", bn_params("unet.encoder.bn.")):
        n = fresh("BatchNormalization"); inits[n] = arr; names.append(n)
    nodes.append({"op_type": "BatchNormalization", "input": ["input"] + names, "output": ["t0"]})
    x = "t0"
    levels = len(m.unet.encoder.layers)
    nb = len(m.unet.encoder.layers[0].conv)
    skips = []
    for lv in range(levels):
        for j in range(nb):
            x = block(x, "unet.encoder.layers.%d.conv.%d" % (lv, j))
        skips.append(x)
        y = "t%d" % len(nodes); nodes.append({"op_type": "AveragePool", "input": [x], "output": [y]}); x = y
    for lv in range(len(m.unet.intermediate.layers)):
        for j in range(nb):
            x = block(x, "unet.intermediate.layers.%d.conv.%d" % (lv, j))
    for lv in range(levels):
        p = "unet.decoder.layers.%d." % lv
        x = relu(conv(x, p + "conv1.0.weight", p + "conv1.1.", transpose=True))
        y = "t%d" % len(nodes); nodes.append({"op_type": "Concat", "input": [x, skips[levels - 1 - lv]], "output": [y]}); x = y
        for j in range(nb):
            x = block(x, p + "conv2.%d" % j)
    x = conv(x, "cnn.weight", None, "cnn.bias")
    H = sd["fc.0.gru.weight_hh_l0"].shape[1]

    def rzn_to_zrh(a):
        return np.concatenate([a[H:2 * H], a[0:H], a[2 * H:3 * H]], axis=0)
    Wp = np.stack([rzn_to_zrh(sd["fc.0.gru.weight_ih_l0" + s_]) for s_ in ("", "_reverse")])
    Rp = np.stack([rzn_to_zrh(sd["fc.0.gru.weight_hh_l0" + s_]) for s_ in ("", "_reverse")])
    Bp = np.stack([np.concatenate([rzn_to_zrh(sd["fc.0.gru.bias_ih_l0" + s_]), rzn_to_zrh(sd["fc.0.gru.bias_hh_l0" + s_])]) for s_ in ("", "_reverse")])
    gn = [fresh("GRU") for _ in range(3)]
    inits[gn[0]], inits[gn[1]], inits[gn[2]] = Wp.astype(np.float32), Rp.astype(np.float32), Bp.astype(np.float32)
    nodes.append({"op_type": "GRU", "input": [x] + gn, "output": ["gru_y"]})
    mm = fresh("MatMul"); inits[mm] = np.ascontiguousarray(sd["fc.1.weight"].T)
    ab = fresh("Add"); inits[ab] = sd["fc.1.bias"]
    nodes.append({"op_type": "MatMul", "input": ["gru_y", mm], "output": ["mm_y"]})
    nodes.append({"op_type": "Add", "input": [ab, "mm_y"], "output": ["lin_y"]})
    nodes.append({"op_type": "Sigmoid", "input": ["lin_y"], "output": ["output"]})
    OR.write_onnx(path, inits, nodes)


def test_rmvpe_structural_import_of_a_name_anonymised_onnx(tmp_path):
    # what a user's `rmvpe.onnx` looks like (BatchNorm folded by the exporter, anonymous operands, GRU in ONNX gate order): the
    # name-based importer has nothing to hold on to, the structural one must produce the same blob as the named import of the
    # checkpoint the graph was exported from -- with folded and with un-folded BatchNorm nodes
    m = _upstream_rmvpe()
    named = {k: v.detach().numpy() for k, v in m.state_dict().items() if "num_batches_tracked" not in k}
    cfg0, t0 = IM.import_rmvpe(named)
    for fold in (True, False):
        p = str(tmp_path / ("rmvpe_%d.onnx" % fold))
        _anonymised_rmvpe_onnx(p, m, fold_bn=fold)
        inits, nodes = OR.read_onnx(p)
        assert not any("unet" in k for k in inits)
        with pytest.raises(IM.ImportError_):
            IM.import_rmvpe(IM.load_named_tensors(p))                    # by name: nothing matches
        cfg, t = IM.import_rmvpe_structural(inits, nodes)
        assert cfg == cfg0, (cfg, cfg0)
        assert set(t) == set(t0)
        for k in t0:
            assert t[k].shape == t0[k].shape and np.allclose(t[k], t0[k], rtol=2e-6, atol=1e-7), k
    # the CLI route picks the structural importer by itself
    out = str(tmp_path / "rmvpe.rvcw")
    IM.main(["rmvpe", str(tmp_path / "rmvpe_1.onnx"), out])
    c2, t2 = W.read_blob(out)
    assert int(c2["levels"]) == cfg0["levels"] and np.allclose(t2["rm.gru.w_hh_b"], t0["rm.gru.w_hh_b"], rtol=2e-6, atol=1e-7)
    # a graph that does not fit the U-Net's channel algebra is refused, not guessed at
    inits, nodes = OR.read_onnx(str(tmp_path / "rmvpe_1.onnx"))
    first_ct = next(i for i, nd in enumerate(nodes) if nd["op_type"] == "ConvTranspose")
    with pytest.raises(IM.ImportError_):
        IM.import_rmvpe_structural(inits, nodes[:first_ct - 6] + nodes[first_ct:])
