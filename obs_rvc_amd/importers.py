"""Weight importers: the on-disk formats users of the reference have -> the engine's RVCW blobs.

The reference loads three ONNX graphs through ONNX Runtime (rvc/src/models.rs:48-76): `<data>/contentvec/vec-{768-layer-12,
256-layer-9}.onnx`, `<data>/f0/rmvpe.onnx` (models.rs:58-61) and the user's synthesizer `<model>.onnx` (models.rs:72).  Those
graphs are exports of three public PyTorch models; this module maps the tensors of either form -- ONNX initializers
(obs_rvc_amd.onnx_reader, no onnx package needed) or a PyTorch checkpoint / safetensors file -- onto the blob layout of
obs_rvc_amd.weights (same names and shapes as the synthetic model zoo), folding what inference folds anyway:
weight-norm (g * v / |v|), BatchNorm into the preceding convolution, the speaker embedding row `emb_g[sid]`.

Tensor names follow the upstream state dicts (fairseq HubertModel or transformers.HubertModel for ContentVec; RVC's
`SynthesizerTrnMs{256,768}NSFsid`; RMVPE's `E2E`).  An ONNX exporter keeps those names for Conv / ConvTranspose / Embedding /
norm parameters and biases but stores `nn.Linear` weights as anonymous transposed MatMul operands: `load_named_tensors`
re-attaches them through the graph (MatMul -> Add whose bias initializer is `<layer>.bias`).  Exports whose initializers were
renamed wholesale (Conv + BatchNorm fused by the exporter's constant folding: the usual state of a public `rmvpe.onnx`) cannot be
mapped by name: `import_rmvpe_structural` assigns them by graph topology and checks every assignment against the U-Net's channel
algebra.  The reference's custom 3-input synthesizer export (rvc/src/rvc.rs:193-203, speaker row and conditioning folded into
constants) is not covered: convert the `.pth` the ONNX file was exported from.  A name table that does not fit fails with the list
of missing tensors instead of guessing.  No real checkpoint exists in this image: the ContentVec table is pinned
against transformers.HubertModel, the RMVPE and synthesizer tables are checked end to end against nn.Modules written with upstream's
structure and parameter names (tests/test_importers.py) -- a real file may still differ in details these tests cannot see.

CLI:  python -m obs_rvc_amd.importers contentvec|rmvpe|synth <input> <output.rvcw> [--version 2] [--sid 0] [--sr 48000]
                                      [--up-rates 10,10,2,2]
"""
from __future__ import annotations

import argparse
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import weights as W
from .onnx_reader import read_onnx

Named = Dict[str, np.ndarray]


class ImportError_(ValueError):
    """Raised with the names that could not be found."""


# --------------------------------------------------------------------------------------------- loading
def load_checkpoint_config(path: str) -> Optional[list]:
    """The hyper-parameter list an RVC model .pth carries next to its weights ({"weight": sd, "config": [...], "sr": ..}): upstream
    order [spec_channels, segment_size, inter, hidden, filter, n_heads, n_layers, kernel_size, p_dropout, resblock,
    resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates, upsample_initial_channel, upsample_kernel_sizes,
    spk_embed_dim, gin_channels, sr].  None for formats that have no such entry (ONNX, safetensors, bare state dicts)."""
    if os.path.splitext(path)[1].lower() in (".onnx", ".safetensors"):
        return None
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)
    cfg = obj.get("config") if isinstance(obj, dict) else None
    return list(cfg) if isinstance(cfg, (list, tuple)) and len(cfg) >= 18 else None


def load_named_tensors(path: str) -> Named:
    ext = os.path.splitext(path)[1].lower()
    if ext == ".onnx":
        inits, nodes = read_onnx(path)
        return _resolve_linear_weights(inits, nodes)
    if ext == ".safetensors":
        from safetensors.numpy import load_file
        return {k: np.asarray(v) for k, v in load_file(path).items()}
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)     # plain tensors / containers only: no code execution
    for key in ("weight", "model", "state_dict"):                     # RVC: {"weight": sd, "config": [...]}; fairseq: {"model": sd}
        if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
            obj = obj[key]
            break
    out: Named = {}
    for k, v in obj.items():
        if hasattr(v, "detach"):
            out[k] = v.detach().to(torch.float32 if v.is_floating_point() else v.dtype).cpu().numpy()
    return out


def _resolve_linear_weights(inits: Named, nodes: List[dict]) -> Named:
    """Give anonymous MatMul operands their parameter name: x @ W followed by Add(bias named '<p>.bias') => '<p>.weight' = W^T."""
    out = dict(inits)
    consumers: Dict[str, List[dict]] = {}
    for nd in nodes:
        for i in nd["input"]:
            consumers.setdefault(i, []).append(nd)
    for nd in nodes:
        if nd["op_type"] not in ("MatMul", "Gemm") or len(nd["input"]) < 2:
            continue
        wname = nd["input"][1]
        if wname not in inits or inits[wname].ndim != 2:
            continue
        bias = None
        if nd["op_type"] == "Gemm" and len(nd["input"]) > 2 and nd["input"][2] in inits:
            bias = nd["input"][2]
        else:
            for c in consumers.get(nd["output"][0], []):
                if c["op_type"] == "Add":
                    for i in c["input"]:
                        if i in inits and i.endswith(".bias"):
                            bias = i
        if bias and bias.endswith(".bias"):
            target = bias[:-5] + ".weight"
            if target not in out:
                w = inits[wname]
                out[target] = np.ascontiguousarray(w.T) if w.shape[0] != inits[bias].shape[0] or nd["op_type"] == "MatMul" else w
    return out


# --------------------------------------------------------------------------------------------- helpers
class _Src:
    def __init__(self, named: Named, prefixes=("",)):
        self.n, self.prefixes, self.missing = named, prefixes, []

    def has(self, name: str) -> bool:
        return any(p + name in self.n for p in self.prefixes)

    def get(self, name: str) -> np.ndarray:
        for p in self.prefixes:
            if p + name in self.n:
                return np.asarray(self.n[p + name], dtype=np.float32)
        self.missing.append(name)
        return np.zeros((1,), np.float32)

    def weight(self, prefix: str) -> np.ndarray:
        """`prefix.weight`, or the weight-norm pair in either of torch's spellings: w = g * v / |v| (norm over the dims g collapses)."""
        if self.has(prefix + ".weight"):
            return self.get(prefix + ".weight")
        for gk, vk in ((".weight_g", ".weight_v"), (".parametrizations.weight.original0", ".parametrizations.weight.original1")):
            if self.has(prefix + gk) and self.has(prefix + vk):
                g, v = self.get(prefix + gk).astype(np.float64), self.get(prefix + vk).astype(np.float64)
                axes = tuple(i for i in range(v.ndim) if g.shape[i] == 1 and v.shape[i] != 1)
                nrm = np.sqrt((v * v).sum(axis=axes, keepdims=True))
                return (v * (g / nrm)).astype(np.float32)
        self.missing.append(prefix + ".weight")
        return np.zeros((1,), np.float32)

    def check(self, what: str):
        if self.missing:
            raise ImportError_("%s: %d tensors not found (first: %s)" % (what, len(self.missing), ", ".join(self.missing[:8])))


def _fold_bn(w: np.ndarray, conv_b: Optional[np.ndarray], gamma, beta, mean, var, eps=1e-5, out_axis=0):
    s = (gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + eps))
    shape = [1] * w.ndim; shape[out_axis] = -1
    wf = w.astype(np.float64) * s.reshape(shape)
    b0 = conv_b.astype(np.float64) if conv_b is not None else 0.0
    bf = beta.astype(np.float64) + (b0 - mean.astype(np.float64)) * s
    return wf.astype(np.float32), bf.astype(np.float32)


def _sq(a: np.ndarray, nd: int) -> np.ndarray:
    """Drop a trailing kernel axis of size 1 (Conv1d k=1 stored as Linear in the blob)."""
    while a.ndim > nd and a.shape[-1] == 1:
        a = a[..., 0]
    return a


# --------------------------------------------------------------------------------------------- ContentVec
def import_contentvec(named: Named, version: int = 2, heads: int = 12, pos_groups: int = 16) -> Tuple[dict, Named]:
    """fairseq HubertModel / ContentVec checkpoint or transformers.HubertModel state dict -> (cfg, tensors) of weights.make_contentvec.
    version 2 = 768-d layer-12 output, version 1 = layer 9 + final_proj to 256 (rvc-common/src/enums.rs:10-23)."""
    s = _Src(named, ("", "model.", "hubert.", "w2v_encoder.w2v_model."))
    hf = s.has("feature_extractor.conv_layers.0.conv.weight")
    t: Named = {}
    ks, ss = [], W.CV_CONV_S
    for i in range(7):
        w = s.get("feature_extractor.conv_layers.%d.%s" % (i, "conv.weight" if hf else "0.weight"))
        t["cv.conv%d.w" % i] = w; ks.append(int(w.shape[-1]))
    gn = "feature_extractor.conv_layers.0." + ("layer_norm" if hf else "2")
    t["cv.gn.g"], t["cv.gn.b"] = s.get(gn + ".weight"), s.get(gn + ".bias")
    ln0, proj = ("feature_projection.layer_norm", "feature_projection.projection") if hf else ("layer_norm", "post_extract_proj")
    t["cv.ln0.g"], t["cv.ln0.b"] = s.get(ln0 + ".weight"), s.get(ln0 + ".bias")
    t["cv.proj.w"], t["cv.proj.b"] = s.get(proj + ".weight"), s.get(proj + ".bias")
    pos = "encoder.pos_conv_embed.conv" if hf else "encoder.pos_conv.0"
    t["cv.pos.w"], t["cv.pos.b"] = s.weight(pos), s.get(pos + ".bias")
    t["cv.enc_ln.g"], t["cv.enc_ln.b"] = s.get("encoder.layer_norm.weight"), s.get("encoder.layer_norm.bias")
    n_layers = 0
    while s.has("encoder.layers.%d.%s.q_proj.weight" % (n_layers, "attention" if hf else "self_attn")):
        n_layers += 1
    run_layers = n_layers if version == 2 else min(9, n_layers)
    att = "attention" if hf else "self_attn"
    names = dict(ln1="layer_norm" if hf else "self_attn_layer_norm", ff1="feed_forward.intermediate_dense" if hf else "fc1",
                 ff2="feed_forward.output_dense" if hf else "fc2", ln2="final_layer_norm")
    for l in range(run_layers):
        p, q = "encoder.layers.%d." % l, "cv.l%d." % l
        for up, mine in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v"), ("out_proj", "o")):
            t[q + mine + ".w"], t[q + mine + ".b"] = s.get(p + att + "." + up + ".weight"), s.get(p + att + "." + up + ".bias")
        for mine in ("ln1", "ln2"):
            t[q + mine + ".g"], t[q + mine + ".b"] = s.get(p + names[mine] + ".weight"), s.get(p + names[mine] + ".bias")
        for mine in ("ff1", "ff2"):
            t[q + mine + ".w"], t[q + mine + ".b"] = s.get(p + names[mine] + ".weight"), s.get(p + names[mine] + ".bias")
    E, C = int(t["cv.proj.w"].shape[0]), int(t["cv.conv0.w"].shape[0])
    out_dim = E
    if version == 1:
        t["cv.final_proj.w"], t["cv.final_proj.b"] = s.get("final_proj.weight"), s.get("final_proj.bias")
        out_dim = int(t["cv.final_proj.w"].shape[0])
    s.check("ContentVec")
    cfg = dict(kind=1, n_conv=7, conv_dim=C, embed=E, heads=heads, ffn=int(t["cv.l0.ff1.w"].shape[0]), layers=n_layers,
               pos_k=int(t["cv.pos.w"].shape[-1]), pos_groups=pos_groups, run_layers=run_layers, out_dim=out_dim)
    for i in range(7):
        cfg["conv_k%d" % i] = ks[i]; cfg["conv_s%d" % i] = ss[i]
    if t["cv.pos.w"].shape[1] * pos_groups != E:
        raise ImportError_("ContentVec: positional conv has %d input channels per group, expected %d" % (t["cv.pos.w"].shape[1], E // pos_groups))
    return cfg, t


# --------------------------------------------------------------------------------------------- RMVPE
def _rm_block(s: _Src, t: Named, src: str, dst: str):
    """ConvBlockRes: conv.0 (3x3, no bias) + conv.1 (BN) + ReLU + conv.3 + conv.4 (BN) + ReLU, optional 1x1 shortcut."""
    for ci, bi, mine in ((0, 1, "c1"), (3, 4, "c2")):
        w = s.get("%s.conv.%d.weight" % (src, ci))
        cb = s.get("%s.conv.%d.bias" % (src, ci)) if s.has("%s.conv.%d.bias" % (src, ci)) else None
        bn = "%s.conv.%d." % (src, bi)
        t[dst + mine + ".w"], t[dst + mine + ".b"] = _fold_bn(w, cb, s.get(bn + "weight"), s.get(bn + "bias"), s.get(bn + "running_mean"), s.get(bn + "running_var"))
    if s.has(src + ".shortcut.weight"):
        t[dst + "sc.w"], t[dst + "sc.b"] = _sq(s.get(src + ".shortcut.weight"), 2), s.get(src + ".shortcut.bias")


def import_rmvpe(named: Named) -> Tuple[dict, Named]:
    """RMVPE `E2E(4, 1, (2, 2))` state dict (rmvpe.pt) -> (cfg, tensors) of weights.make_rmvpe.  BatchNorm (eval) is folded."""
    s = _Src(named, ("", "model."))
    t: Named = {}
    bn = "unet.encoder.bn."
    g, b, m, v = (s.get(bn + k) for k in ("weight", "bias", "running_mean", "running_var"))
    sc = float(g[0]) / float(np.sqrt(v[0] + 1e-5)) if g.size else 1.0
    t["rm.bn0"] = np.array([sc, float(b[0]) - float(m[0]) * sc if b.size else 0.0], np.float32)
    levels = 0
    while s.has("unet.encoder.layers.%d.conv.0.conv.0.weight" % levels):
        levels += 1
    n_blocks = 0
    while s.has("unet.encoder.layers.0.conv.%d.conv.0.weight" % n_blocks):
        n_blocks += 1
    inter = 0
    while s.has("unet.intermediate.layers.%d.conv.0.conv.0.weight" % inter):
        inter += 1
    for lv in range(levels):
        for j in range(n_blocks):
            _rm_block(s, t, "unet.encoder.layers.%d.conv.%d" % (lv, j), "rm.enc%d.b%d." % (lv, j))
    for lv in range(inter):
        for j in range(n_blocks):
            _rm_block(s, t, "unet.intermediate.layers.%d.conv.%d" % (lv, j), "rm.int%d.b%d." % (lv, j))
    for lv in range(levels):
        p = "unet.decoder.layers.%d." % lv
        w = s.get(p + "conv1.0.weight")                     # ConvTranspose2d [Cin][Cout][3][3], no bias, then BN over Cout
        t["rm.dec%d.up.w" % lv], t["rm.dec%d.up.b" % lv] = _fold_bn(w, None, s.get(p + "conv1.1.weight"), s.get(p + "conv1.1.bias"),
                                                                  s.get(p + "conv1.1.running_mean"), s.get(p + "conv1.1.running_var"), out_axis=1)
        for j in range(n_blocks):
            _rm_block(s, t, p + "conv2.%d" % j, "rm.dec%d.b%d." % (lv, j))
    t["rm.cnn.w"], t["rm.cnn.b"] = s.get("cnn.weight"), s.get("cnn.bias")
    for d, suf in (("f", ""), ("b", "_reverse")):
        t["rm.gru.w_ih_" + d], t["rm.gru.w_hh_" + d] = s.get("fc.0.gru.weight_ih_l0" + suf), s.get("fc.0.gru.weight_hh_l0" + suf)
        t["rm.gru.b_ih_" + d], t["rm.gru.b_hh_" + d] = s.get("fc.0.gru.bias_ih_l0" + suf), s.get("fc.0.gru.bias_hh_l0" + suf)
    t["rm.fc.w"], t["rm.fc.b"] = s.get("fc.1.weight"), s.get("fc.1.bias")
    s.check("RMVPE")
    H = int(t["rm.gru.w_hh_f"].shape[1])
    cfg = dict(kind=2, en_out=int(t["rm.enc0.b0.c1.w"].shape[0]), levels=levels, n_blocks=n_blocks, inter_layers=inter,
               n_mels=int(t["rm.gru.w_ih_f"].shape[1]) // 3, gru_hidden=H, n_out=int(t["rm.fc.w"].shape[0]))
    return cfg, t


def import_rmvpe_structural(inits: Named, nodes: List[dict]) -> Tuple[dict, Named]:
    """RMVPE from an ONNX graph whose initializers carry NO usable names (public `rmvpe.onnx` exports: the exporter folds every
    Conv + BatchNorm pair and names the result `onnx::Conv_123`, packs the GRU as `onnx::GRU_*` in ONNX's own gate order, stores
    the Linear as an anonymous MatMul operand).  The tensors are assigned by TOPOLOGY instead: ONNX stores nodes in execution
    order, and the order in which `E2E.forward` runs its weight-bearing ops is fixed by the architecture --

        [BatchNormalization on the mel]  encoder: per level n_blocks x (conv3x3, conv3x3, [1x1 shortcut iff Cin != Cout])
        intermediate layers (same blocks)  decoder: per level ConvTranspose 3x3 s2 then n_blocks blocks  conv3x3 (-> 3)  GRU  MatMul(+Add)

    Levels = number of ConvTranspose nodes; a level starts at every block that carries a shortcut; the intermediate layers are the
    blocks between the last encoder level and the first ConvTranspose.  Every assignment is checked against the channel algebra of
    the U-Net (Cout doubles per encoder level, the decoder's first block of a level sees 2 x Cout inputs, ...); a graph that does
    not fit raises instead of producing a silently wrong blob.  A Conv that still has its BatchNormalization behind it (an export
    without constant folding) is folded here."""
    producers = {}
    for nd in nodes:
        for o in nd["output"]:
            producers[o] = nd
    consumers: Dict[str, List[dict]] = {}
    for nd in nodes:
        for i in nd["input"]:
            consumers.setdefault(i, []).append(nd)

    def init_in(nd, k):
        return inits[nd["input"][k]] if len(nd["input"]) > k and nd["input"][k] in inits else None

    toks = []          # (kind, dict)
    for nd in nodes:
        op = nd["op_type"]
        if op in ("Conv", "ConvTranspose"):
            w = init_in(nd, 1)
            if w is None or w.ndim != 4:
                continue
            b = init_in(nd, 2)
            bn = None
            for c in consumers.get(nd["output"][0], []):
                if c["op_type"] == "BatchNormalization" and all(x in inits for x in c["input"][1:5]):
                    bn = [inits[x] for x in c["input"][1:5]]
            toks.append(("CT" if op == "ConvTranspose" else "C", dict(w=np.asarray(w, np.float32), b=None if b is None else np.asarray(b, np.float32), bn=bn)))
        elif op == "BatchNormalization" and all(x in inits for x in nd["input"][1:5]):
            src = producers.get(nd["input"][0])
            if src is None or src["op_type"] not in ("Conv", "ConvTranspose"):
                toks.append(("BN", dict(p=[np.asarray(inits[x], np.float32) for x in nd["input"][1:5]])))
        elif op == "GRU":
            toks.append(("GRU", dict(W=np.asarray(init_in(nd, 1), np.float32), R=np.asarray(init_in(nd, 2), np.float32), B=np.asarray(init_in(nd, 3), np.float32))))
        elif op in ("MatMul", "Gemm"):
            w = init_in(nd, 1)
            if w is None or w.ndim != 2:
                continue
            b = init_in(nd, 2) if op == "Gemm" else None
            if b is None:
                for c in consumers.get(nd["output"][0], []):
                    if c["op_type"] == "Add":
                        for i in c["input"]:
                            if i in inits and inits[i].ndim == 1:
                                b = inits[i]
            toks.append(("FC", dict(w=np.asarray(w, np.float32), b=None if b is None else np.asarray(b, np.float32), transposed=(op == "MatMul"))))

    def fail(msg):
        raise ImportError_("RMVPE (structural): " + msg)

    def folded(tok, out_axis=0):
        w, b, bn = tok["w"], tok["b"], tok["bn"]
        if bn is not None:
            return _fold_bn(w, b, bn[0], bn[1], bn[2], bn[3], out_axis=out_axis)
        if b is None:
            fail("a convolution has neither a bias nor a BatchNormalization behind it (BN neither folded nor present)")
        return w, b

    t: Named = {}
    pos = 0
    if toks and toks[0][0] == "BN":
        g, b, m, v = toks[0][1]["p"]
        sc = float(g[0]) / float(np.sqrt(v[0] + 1e-5))
        t["rm.bn0"] = np.array([sc, float(b[0]) - float(m[0]) * sc], np.float32)
        pos = 1
    else:
        fail("the graph does not start with the input BatchNormalization")
    levels = sum(1 for k, _ in toks if k == "CT")
    if levels < 1:
        fail("no ConvTranspose nodes: not a U-Net")

    def read_block(ci_expect):
        nonlocal pos
        if pos + 1 >= len(toks) or toks[pos][0] != "C" or toks[pos + 1][0] != "C":
            fail("expected two 3x3 convolutions at weight-op %d" % pos)
        a, b2 = toks[pos][1], toks[pos + 1][1]
        if a["w"].shape[2:] != (3, 3) or b2["w"].shape[2:] != (3, 3):
            fail("block at weight-op %d is not conv3x3 + conv3x3" % pos)
        co, ci = int(a["w"].shape[0]), int(a["w"].shape[1])
        if ci_expect is not None and ci != ci_expect:
            fail("block at weight-op %d takes %d channels, the U-Net supplies %d" % (pos, ci, ci_expect))
        if b2["w"].shape[:2] != (co, co):
            fail("second convolution of the block at weight-op %d is %s, expected (%d, %d, 3, 3)" % (pos, b2["w"].shape, co, co))
        pos += 2
        sc = None
        if ci != co:
            if pos >= len(toks) or toks[pos][0] != "C" or toks[pos][1]["w"].shape != (co, ci, 1, 1):
                fail("block with %d -> %d channels has no 1x1 shortcut behind it" % (ci, co))
            sc = toks[pos][1]
            pos += 1
        return a, b2, sc, ci, co

    def store(dst, blk):
        a, b2, sc, _, _ = blk
        t[dst + "c1.w"], t[dst + "c1.b"] = folded(a)
        t[dst + "c2.w"], t[dst + "c2.b"] = folded(b2)
        if sc is not None:
            t[dst + "sc.w"], t[dst + "sc.b"] = _sq(sc["w"], 2), (sc["b"] if sc["b"] is not None else np.zeros(sc["w"].shape[0], np.float32))

    first_ct = next(i for i, (k, _) in enumerate(toks) if k == "CT")
    pre = []
    c_in = 1
    while pos < first_ct:
        blk = read_block(c_in)
        pre.append(blk)
        c_in = blk[4]
    starts = [i for i, blk in enumerate(pre) if blk[2] is not None]         # a level / the first intermediate layer starts at a shortcut
    if len(starts) < levels + 1 or starts[0] != 0:
        fail("%d blocks with a shortcut before the first ConvTranspose, expected at least %d (encoder levels + intermediate)" % (len(starts), levels + 1))
    n_blocks = starts[1] - starts[0]
    if any(starts[i + 1] - starts[i] != n_blocks for i in range(levels)) or (len(pre) - levels * n_blocks) % n_blocks != 0:
        fail("blocks per level are not uniform")
    inter = (len(pre) - levels * n_blocks) // n_blocks
    for lv in range(levels):
        for j in range(n_blocks):
            store("rm.enc%d.b%d." % (lv, j), pre[lv * n_blocks + j])
        if lv and pre[lv * n_blocks][4] != 2 * pre[(lv - 1) * n_blocks][4]:
            fail("encoder level %d does not double the channel count" % lv)
    for lv in range(inter):
        for j in range(n_blocks):
            store("rm.int%d.b%d." % (lv, j), pre[(levels + lv) * n_blocks + j])
    for lv in range(levels):
        if toks[pos][0] != "CT":
            fail("expected the ConvTranspose of decoder level %d at weight-op %d" % (lv, pos))
        up = toks[pos][1]
        pos += 1
        ci, co = int(up["w"].shape[0]), int(up["w"].shape[1])
        if ci != c_in or up["w"].shape[2:] != (3, 3):
            fail("decoder level %d upsamples %s, expected %d input channels, 3x3" % (lv, up["w"].shape, c_in))
        t["rm.dec%d.up.w" % lv], t["rm.dec%d.up.b" % lv] = folded(up, out_axis=1)
        c_in = 2 * co                                                       # concat with the encoder skip of the same size
        for j in range(n_blocks):
            blk = read_block(c_in)
            store("rm.dec%d.b%d." % (lv, j), blk)
            c_in = blk[4]
    if pos >= len(toks) or toks[pos][0] != "C" or toks[pos][1]["w"].shape[1] != c_in:
        fail("expected the final 3x3 convolution after the decoder")
    t["rm.cnn.w"], t["rm.cnn.b"] = toks[pos][1]["w"], (toks[pos][1]["b"] if toks[pos][1]["b"] is not None else np.zeros(toks[pos][1]["w"].shape[0], np.float32))
    pos += 1
    if pos >= len(toks) or toks[pos][0] != "GRU":
        fail("expected the GRU after the final convolution")
    gru = toks[pos][1]
    pos += 1
    H = gru["R"].shape[2]
    if gru["W"].shape[0] != 2 or gru["W"].shape[1] != 3 * H or gru["B"].shape != (2, 6 * H):
        fail("GRU operands %s / %s / %s are not a bidirectional single layer" % (gru["W"].shape, gru["R"].shape, gru["B"].shape))

    def zrh_to_rzn(a):          # ONNX gate order (z, r, h) -> PyTorch (r, z, n), along axis 0 of a [3H, ...] block
        return np.concatenate([a[H:2 * H], a[0:H], a[2 * H:3 * H]], axis=0)
    for d, name in enumerate(("f", "b")):                                   # direction 0 = forward, 1 = reverse
        t["rm.gru.w_ih_" + name], t["rm.gru.w_hh_" + name] = zrh_to_rzn(gru["W"][d]), zrh_to_rzn(gru["R"][d])
        t["rm.gru.b_ih_" + name], t["rm.gru.b_hh_" + name] = zrh_to_rzn(gru["B"][d, :3 * H]), zrh_to_rzn(gru["B"][d, 3 * H:])
    if pos >= len(toks) or toks[pos][0] != "FC":
        fail("expected the output Linear after the GRU")
    fc = toks[pos][1]
    w = fc["w"].T if (fc["transposed"] or fc["w"].shape[0] == 2 * H) and fc["w"].shape[0] == 2 * H else fc["w"]
    if w.shape[1] != 2 * H or fc["b"] is None:
        fail("output Linear is %s, expected (n_out, %d) with a bias" % (fc["w"].shape, 2 * H))
    t["rm.fc.w"], t["rm.fc.b"] = np.ascontiguousarray(w), fc["b"]
    cfg = dict(kind=2, en_out=int(t["rm.enc0.b0.c1.w"].shape[0]), levels=levels, n_blocks=n_blocks, inter_layers=inter,
               n_mels=int(t["rm.gru.w_ih_f"].shape[1]) // 3, gru_hidden=int(H), n_out=int(t["rm.fc.w"].shape[0]))
    if t["rm.cnn.w"].shape[0] * cfg["n_mels"] != t["rm.gru.w_ih_f"].shape[1]:
        fail("GRU input width does not equal 3 x n_mels")
    return cfg, t


# --------------------------------------------------------------------------------------------- synthesizer
def import_synth(named: Named, sid: int = 0, sr: Optional[int] = None, up_rates: Optional[List[int]] = None,
                 heads: Optional[int] = None, window: int = 10, config: Optional[list] = None) -> Tuple[dict, Named]:
    """RVC `SynthesizerTrnMs{256,768}NSFsid` state dict (the "weight" entry of a model .pth) -> (cfg, tensors) of weights.make_synth.
    The speaker embedding row `emb_g.weight[sid]` is baked, as in the reference's 3-input export (rvc/src/rvc.rs:186-203).

    Upsample rates cannot be read off the weights (official configs pair kernel 16 with rate 10 *and* with rate 8): they come from,
    in this order, `up_rates`, the checkpoint's `config` list (entries 12 / 14 / 17 = upsample_rates / upsample_kernel_sizes / sr,
    see load_checkpoint_config), or `sr` (the noise convs pin every rate but the first: rate0 = (sr / 100) / prod(rates[1:])).
    Without any of the three the import fails instead of guessing.  The result always satisfies sr == 100 * prod(rates), which the
    engine checks at load."""
    if config is not None:
        if up_rates is None:
            up_rates = [int(r) for r in config[12]]
        if sr is None:
            s_ = config[17]
            sr = int({"32k": 32000, "40k": 40000, "48k": 48000}.get(s_, s_)) if not isinstance(s_, (int, float)) else int(s_)
        if heads is None:
            heads = int(config[5])
    if heads is None:
        heads = 2
    s = _Src(named, ("", "weight."))
    t: Named = {}
    t["sy.g"] = s.get("emb_g.weight")[sid] if s.has("emb_g.weight") else s.get("emb_g.weight")
    t["sy.enc.phone.w"], t["sy.enc.phone.b"] = s.get("enc_p.emb_phone.weight"), s.get("enc_p.emb_phone.bias")
    t["sy.enc.pitch_emb"] = s.get("enc_p.emb_pitch.weight")
    L = 0
    while s.has("enc_p.encoder.attn_layers.%d.conv_q.weight" % L):
        L += 1
    for i in range(L):
        a, q = "enc_p.encoder.attn_layers.%d." % i, "sy.enc.l%d." % i
        for n in "qkvo":
            t[q + n + ".w"], t[q + n + ".b"] = _sq(s.get(a + "conv_%s.weight" % n), 2), s.get(a + "conv_%s.bias" % n)
        t[q + "rel_k"], t[q + "rel_v"] = s.get(a + "emb_rel_k")[0], s.get(a + "emb_rel_v")[0]
        for k, up in (("ln1", "norm_layers_1"), ("ln2", "norm_layers_2")):
            t[q + k + ".g"], t[q + k + ".b"] = s.get("enc_p.encoder.%s.%d.gamma" % (up, i)), s.get("enc_p.encoder.%s.%d.beta" % (up, i))
        for k, up in (("ff1", "conv_1"), ("ff2", "conv_2")):
            t[q + k + ".w"], t[q + k + ".b"] = s.get("enc_p.encoder.ffn_layers.%d.%s.weight" % (i, up)), s.get("enc_p.encoder.ffn_layers.%d.%s.bias" % (i, up))
    t["sy.enc.proj.w"], t["sy.enc.proj.b"] = _sq(s.get("enc_p.proj.weight"), 2), s.get("enc_p.proj.bias")
    n_flow = 0
    while s.has("flow.flows.%d.pre.weight" % (2 * n_flow)):
        n_flow += 1
    wn_layers = 0
    for i in range(n_flow):
        f, q = "flow.flows.%d." % (2 * i), "sy.flow%d." % i
        t[q + "pre.w"], t[q + "pre.b"] = _sq(s.get(f + "pre.weight"), 2), s.get(f + "pre.bias")
        t[q + "cond.w"], t[q + "cond.b"] = _sq(s.weight(f + "enc.cond_layer"), 2), s.get(f + "enc.cond_layer.bias")
        j = 0
        while s.has(f + "enc.in_layers.%d.bias" % j):
            t[q + "in%d.w" % j], t[q + "in%d.b" % j] = s.weight(f + "enc.in_layers.%d" % j), s.get(f + "enc.in_layers.%d.bias" % j)
            t[q + "rs%d.w" % j], t[q + "rs%d.b" % j] = _sq(s.weight(f + "enc.res_skip_layers.%d" % j), 2), s.get(f + "enc.res_skip_layers.%d.bias" % j)
            j += 1
        wn_layers = j
        t[q + "post.w"], t[q + "post.b"] = _sq(s.get(f + "post.weight"), 2), s.get(f + "post.bias")
    t["sy.dec.pre.w"], t["sy.dec.pre.b"] = s.get("dec.conv_pre.weight"), s.get("dec.conv_pre.bias")
    t["sy.dec.cond.w"], t["sy.dec.cond.b"] = _sq(s.get("dec.cond.weight"), 2), s.get("dec.cond.bias")
    t["sy.src"] = np.array([float(s.get("dec.m_source.l_linear.weight").reshape(-1)[0]), float(s.get("dec.m_source.l_linear.bias").reshape(-1)[0])], np.float32)
    n_ups = 0
    while s.has("dec.ups.%d.bias" % n_ups):
        n_ups += 1
    n_res = 0
    while s.has("dec.resblocks.%d.convs1.0.bias" % n_res):
        n_res += 1
    n_rb = n_res // max(n_ups, 1)
    kernels, rb_k, rb_d = [], [], []
    n_rbd = 0
    while s.has("dec.resblocks.0.convs1.%d.bias" % n_rbd):
        n_rbd += 1
    for i in range(n_ups):
        t["sy.dec.up%d.w" % i], t["sy.dec.up%d.b" % i] = s.weight("dec.ups.%d" % i), s.get("dec.ups.%d.bias" % i)
        kernels.append(int(t["sy.dec.up%d.w" % i].shape[-1]))
        t["sy.dec.nc%d.w" % i], t["sy.dec.nc%d.b" % i] = s.get("dec.noise_convs.%d.weight" % i), s.get("dec.noise_convs.%d.bias" % i)
        for j in range(n_rb):
            r, q = "dec.resblocks.%d." % (i * n_rb + j), "sy.dec.rb%d_%d." % (i, j)
            for m in range(n_rbd):
                t[q + "c1_%d.w" % m], t[q + "c1_%d.b" % m] = s.weight(r + "convs1.%d" % m), s.get(r + "convs1.%d.bias" % m)
                t[q + "c2_%d.w" % m], t[q + "c2_%d.b" % m] = s.weight(r + "convs2.%d" % m), s.get(r + "convs2.%d.bias" % m)
            if i == 0:
                rb_k.append(int(t[q + "c1_0.w"].shape[-1]))
    t["sy.dec.post.w"] = s.get("dec.conv_post.weight")
    s.check("synthesizer")
    # the noise conv of stage i spans 2 * prod(rates after i) samples (kernel): that pins every rate but the first
    tail = [int(t["sy.dec.nc%d.w" % i].shape[-1]) // 2 for i in range(n_ups - 1)]      # prod(rates[i+1:])
    tail_rates = [tail[i] // (tail[i + 1] if i + 1 < len(tail) else 1) for i in range(len(tail))]
    if up_rates is None:
        if not sr:
            raise ImportError_("synthesizer: the first upsample rate is not recoverable from the weights (kernel %d is used with rate %d "
                               "and other rates in official configs); pass up_rates, the checkpoint's config list or sr" % (kernels[0], kernels[0] // 2))
        tp = int(np.prod(tail_rates)) if tail_rates else 1
        if sr % (100 * tp) != 0:
            raise ImportError_("synthesizer: sr %d is not 100 * (a multiple of the tail rates' product %d)" % (sr, tp))
        up_rates = [sr // (100 * tp)] + tail_rates
    up_rates = [int(r) for r in up_rates]
    if len(up_rates) != n_ups or up_rates[1:] != tail_rates:
        raise ImportError_("synthesizer: upsample rates %s do not match the noise-conv geometry of the weights (rates after the first: %s)" % (up_rates, tail_rates))
    for i in range(n_ups):
        if kernels[i] < up_rates[i] or (kernels[i] - up_rates[i]) % 2 != 0:
            raise ImportError_("synthesizer: upsample kernel %d / rate %d of stage %d is not a HiFiGAN pair (padding (k - r) / 2)" % (kernels[i], up_rates[i], i))
    if config is not None and [int(k) for k in config[14]] != kernels:
        raise ImportError_("synthesizer: config upsample_kernel_sizes %s != kernels in the weights %s" % (list(config[14]), kernels))
    if sr and int(sr) != 100 * int(np.prod(up_rates)):
        raise ImportError_("synthesizer: sr %d != 100 * prod(upsample rates %s) -- the f0 frame rate is 100 Hz" % (sr, up_rates))
    rb_d = [1, 3, 5][:n_rbd]
    Hd, I = int(t["sy.enc.phone.w"].shape[0]), int(t["sy.enc.proj.w"].shape[0]) // 2
    cfg = dict(kind=3, phone_dim=int(t["sy.enc.phone.w"].shape[1]), n_rb=n_rb, n_rbd=n_rbd, inter=I, hidden=Hd,
               filter=int(t["sy.enc.l0.ff1.w"].shape[0]), heads=heads, enc_layers=L, enc_k=int(t["sy.enc.l0.ff1.w"].shape[-1]),
               window=(int(t["sy.enc.l0.rel_k"].shape[0]) - 1) // 2, flow_n=n_flow, wn_layers=wn_layers, wn_k=int(t["sy.flow0.in0.w"].shape[-1]),
               gin=int(t["sy.g"].shape[0]), up_init=int(t["sy.dec.pre.w"].shape[0]), n_ups=n_ups,
               sr=int(sr) if sr else 100 * int(np.prod(up_rates)))
    for i in range(n_ups):
        cfg["up_rate%d" % i] = int(up_rates[i]); cfg["up_kernel%d" % i] = kernels[i]
    for j, k in enumerate(rb_k):
        cfg["rb_k%d" % j] = k
    for m, d in enumerate(rb_d):
        cfg["rb_d%d" % m] = d
    return cfg, t


def convert(kind: str, src: str, dst: str, **kw) -> None:
    named = load_named_tensors(src)
    if kind == "synth" and kw.get("config") is None:
        kw["config"] = load_checkpoint_config(src)
    if kind == "rmvpe" and src.lower().endswith(".onnx") and not any(k.endswith("unet.encoder.bn.weight") for k in named):
        # name-anonymised export (Conv + BN folded by the exporter, GRU / Linear operands renamed): assign by topology
        inits, nodes = read_onnx(src)
        cfg, t = import_rmvpe_structural(inits, nodes)
        W.write_blob(dst, cfg, {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in t.items()})
        return
    cfg, t = {"contentvec": import_contentvec, "rmvpe": import_rmvpe, "synth": import_synth}[kind](named, **kw)
    W.write_blob(dst, cfg, {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in t.items()})


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("kind", choices=["contentvec", "rmvpe", "synth"])
    ap.add_argument("src"); ap.add_argument("dst")
    ap.add_argument("--version", type=int, default=2); ap.add_argument("--sid", type=int, default=0); ap.add_argument("--sr", type=int, default=None)
    ap.add_argument("--up-rates", default=None, help="synthesizer upsample rates, e.g. 10,10,2,2 (default: the checkpoint's config list, else from --sr)")
    a = ap.parse_args(argv)
    ur = [int(v) for v in a.up_rates.split(",")] if a.up_rates else None
    kw = dict(version=a.version) if a.kind == "contentvec" else (dict(sid=a.sid, sr=a.sr, up_rates=ur) if a.kind == "synth" else {})
    convert(a.kind, a.src, a.dst, **kw)
    print("wrote", a.dst)


if __name__ == "__main__":
    main()
