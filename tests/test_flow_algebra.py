"""The algebra behind ModelSY::compose_flows (obs_rvc_amd/csrc/engine.hip), restated in numpy and checked in double precision on a small random model.

A flow (reference: the synthesizer's ResidualCouplingLayer + WN, run in reverse at inference) is
    h = pre(x0);  for j: a_j = tanh(in_j(h)[:H]) * sigmoid(in_j(h)[H:]);  rs = res_skip_j(a_j);  h += rs[:H] (j < n-1);  skip += rs[H:] (or all of rs for the last)
    x1 -= post(skip)
with 5-tap zero-padded in-layers and 1x1 pre / res_skip / post layers.  The engine composes the 1x1 layers into what consumes them:
    in_j(h_j) = W_j * [1 | h0 | a_0 .. a_{j-1}]       (constants of the res_skip layers ride on a row of ones that is zero in the padding)
    [h0 of the next flow | z_next] = one 1x1 layer over [a_0 .. a_{n-1} | z]
These tests pin that the composed form equals the layer-by-layer form, edges included, for both flip parities."""
import numpy as np

RNG = np.random.default_rng(5)
H, HALF, K5, NL, T, NF = 16, 8, 5, 4, 21, 4
I = 2 * HALF


def conv_same(x, w, b):
    """x [Cin][T], w [Cout][Cin][K] (zero padding, dilation 1), b [Cout] -> [Cout][T]"""
    k = w.shape[2]; pad = (k - 1) // 2
    xp = np.pad(x, ((0, 0), (pad, pad)))
    y = np.zeros((w.shape[0], x.shape[1]))
    for t in range(k):
        y += w[:, :, t] @ xp[:, t:t + x.shape[1]]
    return y + b[:, None]


def make_flow():
    f = {"pre_w": RNG.normal(size=(H, HALF)) * 0.3, "pre_b": RNG.normal(size=H) * 0.1,
         "in_w": [RNG.normal(size=(2 * H, H, K5)) * 0.15 for _ in range(NL)], "in_b": [RNG.normal(size=2 * H) * 0.1 for _ in range(NL)],
         "rs_w": [RNG.normal(size=((2 * H if j < NL - 1 else H), H)) * 0.2 for j in range(NL)],
         "rs_b": [RNG.normal(size=(2 * H if j < NL - 1 else H)) * 0.1 for j in range(NL)],
         "post_w": RNG.normal(size=(HALF, H)) * 0.3, "post_b": RNG.normal(size=HALF) * 0.1}
    return f


def glu(u):
    return np.tanh(u[:H]) * (1.0 / (1.0 + np.exp(-u[H:])))


def flow_reference(f, z, flipped):
    """layer by layer, on the physical channel order: a flipped flow takes x0 from the upper half and updates the lower one"""
    x0 = z[HALF:] if flipped else z[:HALF]
    h = f["pre_w"] @ x0 + f["pre_b"][:, None]
    skip = np.zeros((H, T)); acts = []
    for j in range(NL):
        a = glu(conv_same(h, f["in_w"][j], f["in_b"][j])); acts.append(a)
        rs = f["rs_w"][j] @ a + f["rs_b"][j][:, None]
        if j < NL - 1:
            h = h + rs[:H]; skip += rs[H:]
        else:
            skip += rs
    z = z.copy()
    upd = f["post_w"] @ skip + f["post_b"][:, None]
    if flipped:
        z[:HALF] -= upd
    else:
        z[HALF:] -= upd
    return z, acts


def compose(flows, flips):
    """the composed panels, as the engine builds them (K orders: in-layers [ones16 | h0 | a_0 ..], last launch [a_0 .. a_{n-1} | z])"""
    out = []
    for fi, (f, flipped) in enumerate(zip(flows, flips)):
        x0r, x1r = (HALF, 0) if flipped else (0, HALF)
        c = {}
        c["pre1_w"] = np.zeros((H, I)); c["pre1_w"][:, x0r:x0r + HALF] = f["pre_w"]; c["pre1_b"] = f["pre_b"]
        c["in_w"] = []
        for j in range(NL):
            cin = 16 + H * (j + 1)
            w = np.zeros((2 * H, cin, K5))
            w[:, 16:16 + H] = f["in_w"][j]
            for i in range(j):
                R, rb = f["rs_w"][i][:H], f["rs_b"][i][:H]
                w[:, 16 + H * (i + 1):16 + H * (i + 2)] = np.einsum("omt,mc->oct", f["in_w"][j], R)
                w[:, 0] += np.einsum("omt,m->ot", f["in_w"][j], rb)
            c["in_w"].append(w)
        ka = NL * H
        pc = np.zeros((HALF, ka)); pcb = f["post_b"].copy()
        for j in range(NL):
            S, sb = (f["rs_w"][j][H:], f["rs_b"][j][H:]) if j < NL - 1 else (f["rs_w"][j], f["rs_b"][j])
            pc[:, j * H:(j + 1) * H] = f["post_w"] @ S; pcb = pcb + f["post_w"] @ sb
        wz = np.zeros((I, ka + I)); wz[:, ka:] = np.eye(I); bz = np.zeros(I)
        wz[x1r:x1r + HALF, :ka] = -pc; bz[x1r:x1r + HALF] = -pcb
        c["z_w"], c["z_b"] = wz, bz
        out.append(c)
    # h0 of the NEXT flow in processing order (flows run from the last to the first): pre_next applied to z_next
    for fi in range(1, len(flows)):
        nxt, nflip = flows[fi - 1], flips[fi - 1]
        x0r = HALF if nflip else 0
        wn = np.zeros((H, I)); wn[:, x0r:x0r + HALF] = nxt["pre_w"]
        out[fi]["h_w"] = wn @ out[fi]["z_w"]; out[fi]["h_b"] = wn @ out[fi]["z_b"] + nxt["pre_b"]
    return out


def run_composed(flows, flips, comp, z):
    ones = np.zeros((16, T)); ones[0] = 1.0
    h0 = comp[-1]["pre1_w"] @ z + comp[-1]["pre1_b"][:, None]
    for fi in range(len(flows) - 1, -1, -1):
        c = comp[fi]
        u = [ones, h0]
        for j in range(NL):
            u.append(glu(conv_same(np.concatenate(u), c["in_w"][j], flows[fi]["in_b"][j])))
        last_in = np.concatenate(u[2:] + [z])
        if fi > 0:
            h0 = c["h_w"] @ last_in + c["h_b"][:, None]
        z = c["z_w"] @ last_in + c["z_b"][:, None]
    return z


def test_composed_wavenets_equal_the_layer_by_layer_flows():
    flows = [make_flow() for _ in range(NF)]
    flips = [((NF - i) & 1) != 0 for i in range(NF)]          # flow i sees the latent after NF - i flips
    z0 = RNG.normal(size=(I, T))
    z = z0.copy()
    for fi in range(NF - 1, -1, -1):
        z, _ = flow_reference(flows[fi], z, flips[fi])
    zc = run_composed(flows, flips, compose(flows, flips), z0.copy())
    assert np.abs(zc - z).max() < 1e-11 * max(1.0, np.abs(z).max())


def test_constants_on_a_plain_bias_would_break_the_window_edges():
    # the reason for the row of ones: a res_skip bias pushed through a zero-padded 5-tap layer is NOT a per-channel constant near the edges
    f = make_flow()
    x = RNG.normal(size=(H, T))
    r = f["rs_b"][0][:H]
    exact = conv_same(x + r[:, None], f["in_w"][1], f["in_b"][1])
    naive = conv_same(x, f["in_w"][1], f["in_b"][1] + np.einsum("omt,m->o", f["in_w"][1], r))
    assert np.abs(exact - naive)[:, 2:-2].max() < 1e-12         # interior: the same
    assert np.abs(exact - naive)[:, :2].max() > 1e-3             # first columns: taps that fall into the padding must not see the constant
