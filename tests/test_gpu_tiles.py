"""Every tile configuration of the implicit-GEMM family forced onto small convolutions and compared with a double-precision host evaluation
(rvc_debug_conv_check, through the C ABI): one parametrised test per kernel family and configuration, so that one failing tile does not hide the others
(VERDICT r5 weak #13), plus every choice the plan-time autotuner can make (round 6) -- SURVEY.md section 8 rows a3, a12, a16: the layers ContentVec /
RMVPE / the synthesizer are made of (the reference runs them inside its ONNX graphs, rvc/src/rvc.rs:92,195, rvc/src/f0/rmvpe.rs:235)."""
import ctypes as C

import pytest

from common import set_opt
from obs_rvc_amd import _native

pytestmark = pytest.mark.gpu
TOL = 2e-5          # largest |gpu - fp64 host| / rms(host): fp32 accumulation over K <= 5632

HOOKS = ("RVC_FORCE_CFG", "RVC_CONV_TILE", "RVC_CONV_TILE_KS", "RVC_FORCE_G2W", "RVC_CONV32S", "RVC_CONV32S_TILE", "RVC_CONV32S_BUF", "RVC_G32L", "RVC_G32L_TAB", "RVC_FORCE_CHOICE", "RVC_G32L_PANEL")
# (M, Cin, KW, dil, N, fused input LeakyReLU): table-free 1x1 layers, dilated multi-tap layers, K shorter and longer than the prefetch depth, ragged M / N
SHAPES = [(48, 48, 1, 1, 111, 0), (144, 48, 1, 1, 111, 0), (96, 384, 1, 1, 37, 0), (40, 32, 7, 3, 300, 1), (64, 512, 3, 1, 50, 0), (33, 16, 11, 1, 130, 1)]


@pytest.fixture()
def conv():
    L = _native.lib()
    L.rvc_debug_conv_check.restype = C.c_double
    L.rvc_debug_conv_check.argtypes = [C.c_void_p] + [C.c_int] * 7
    L.rvc_debug_last_kernel.restype = C.c_char_p
    h = C.c_void_p()
    assert L.rvc_create(b"/tmp", 0, C.byref(h)) == 0

    def check(M, Cin, KW, dil, N, streams, pre):
        return L.rvc_debug_conv_check(h, M, Cin, KW, dil, N, streams, pre)
    check.kernel = lambda: L.rvc_debug_last_kernel().decode()
    try:
        yield check
    finally:
        for k in HOOKS:
            set_opt(k, None)
        L.rvc_destroy(h)


@pytest.mark.parametrize("ks", [1, 4, 8, 16])
@pytest.mark.parametrize("cfg", range(5))
def test_register_direct_tile(conv, cfg, ks):
    # igemm2_kernel: 5 tile shapes x 4 in-workgroup K splits
    set_opt("RVC_FORCE_CFG", "%d,%d" % (cfg, ks))
    for (M, Cin, KW, dil, N, pre) in SHAPES:
        if ks > 1 and (Cin * KW + 15) // 16 < ks:
            continue                           # fewer K chunks than waves: the planner never splits that far
        e = conv(M, Cin, KW, dil, N, 1, pre)
        assert 0 <= e < TOL, (cfg, ks, M, Cin, KW, dil, N, pre, e)


@pytest.mark.parametrize("ks", [1, 2, 3, 4, 6, 8, 12, 16])
@pytest.mark.parametrize("tile", range(3))
def test_igemm2w_tile(conv, tile, ks):
    if ks > 8 and tile != 0:
        pytest.skip("12 / 16 waves splitting K exist for the 32 x 32 wave tile only (round 6, one-stream experiment)")
    # igemm2w_kernel (register-direct 32x32x2 tiles for the table-free 1x1 layers at a few streams): every wave tile x K split, one stream and streams folded
    # into N, ragged M / N, K shorter and longer than the register ring, K splits that leave waves without a chunk
    set_opt("RVC_FORCE_G2W", "%d,%d" % (tile, ks))
    for streams in (1, 3, 8):
        for (M, Cin, N) in [(48, 48, 111), (144, 48, 111), (96, 384, 37), (768, 256, 111), (100, 1040, 70), (64, 16, 33)]:
            e = conv(M, Cin, 1, 1, N, streams, 0)
            assert 0 <= e < TOL, (tile, ks, streams, M, Cin, N, e)


@pytest.mark.parametrize("ks", ["1", "2"])
def test_conv_tile(conv, ks):
    # conv_tile_kernel (one stream, stride-1 1-D convolutions whose input channels come in 16s: input tile staged once per workgroup, K walked tap-major from
    # repacked weights): every tile shape (by panel height), one and two K shares, forced onto short and long layers alike
    set_opt("RVC_CONV_TILE", "2"); set_opt("RVC_CONV_TILE_KS", ks)
    for (M, Cin, KW, dil, N, pre) in [(40, 32, 7, 3, 300, 1), (64, 512, 3, 1, 50, 0), (33, 16, 11, 1, 130, 1), (100, 48, 5, 2, 1000, 0), (128, 128, 7, 3, 2520, 1),
                                      (128, 128, 11, 5, 700, 1), (32, 32, 11, 1, 10080, 1), (64, 64, 7, 1, 5040, 0), (16, 16, 1, 1, 40, 0), (256, 64, 3, 1, 97, 1)]:
        e = conv(M, Cin, KW, dil, N, 1, pre)
        assert 0 <= e < TOL, (ks, M, Cin, KW, dil, N, pre, e)


@pytest.mark.parametrize("streams", [1, 3, 8])
@pytest.mark.parametrize("tile", range(3))
def test_conv32s_tile(conv, tile, streams):
    # conv32s_kernel / conv32s_buf_kernel (stride-1 1-D convolutions at five streams and more: input rows of a 32-channel block staged once per workgroup, taps
    # walked from LDS, K walked (block, tap, group)-major from the per-model repacked panels): every tile; ragged M / N, one to eight channel blocks, reach of
    # the taps from 0 to 50 columns, N shorter than a tile
    set_opt("RVC_CONV32S", "2"); set_opt("RVC_CONV32S_TILE", str(tile))
    for (M, Cin, KW, dil, N, pre) in [(32, 32, 11, 1, 1000, 1), (64, 64, 7, 3, 520, 0), (128, 128, 11, 5, 700, 1), (40, 32, 7, 3, 300, 1), (256, 64, 3, 1, 97, 1),
                                      (100, 96, 5, 2, 333, 0), (32, 32, 1, 1, 256, 0), (256, 256, 3, 1, 252, 1)]:
        e = conv(M, Cin, KW, dil, N, streams, pre)
        assert 0 <= e < TOL and conv.kernel() == "c32s", (tile, streams, M, Cin, KW, dil, N, pre, e, conv.kernel())


@pytest.mark.parametrize("buf", ["0", "2"])
def test_conv32s_64x128_tile_in_both_kernels(conv, buf):
    # the 64 x 128 tile exists in both kernels -- conv32s_buf_kernel below 24 streams, conv32s_kernel from there: each forced with the other's stream counts
    set_opt("RVC_CONV32S", "2"); set_opt("RVC_CONV32S_TILE", "1"); set_opt("RVC_CONV32S_BUF", buf)
    for (M, Cin, KW, dil, N, pre, streams) in [(64, 64, 7, 3, 520, 1, 3), (64, 64, 11, 5, 700, 0, 8), (40, 32, 3, 1, 300, 1, 1)]:
        e = conv(M, Cin, KW, dil, N, streams, pre)
        assert 0 <= e < TOL, (buf, M, Cin, KW, dil, N, pre, streams, e)


@pytest.mark.parametrize("streams", [3, 20])
def test_streams_folded_into_n(conv, streams):
    # folded streams; 20 streams reach the workgroup-tiled kernels on the wide layers (+ a 48-row panel wide enough for the 48 x 256 workgroup tile, a 32-row
    # and a 64-row panel for the narrow 32x32x2 tiles)
    for (M, Cin, KW, dil, N, pre) in SHAPES + [(128, 128, 7, 3, 2520, 1), (768, 256, 1, 1, 111, 0), (48, 48, 15, 1, 5000, 0), (32, 32, 11, 1, 10080, 1), (64, 64, 7, 1, 5040, 0)]:
        e = conv(M, Cin, KW, dil, N, streams, pre)
        assert 0 <= e < TOL, (streams, M, Cin, KW, dil, N, pre, e)


@pytest.mark.parametrize("streams", [8, 16])
def test_tall_panels_at_few_streams(conv, streams):
    # the 64 x 64 tile of the 32x32x2 kernel (250-500 workgroups of 128 x 64) at 8 streams, the 128 x 64 tile at 16
    for (M, Cin, KW, dil, N, pre) in [(3072, 32, 1, 1, 111, 0), (2304, 48, 1, 1, 111, 1), (600, 32, 3, 1, 500, 0)]:
        e = conv(M, Cin, KW, dil, N, streams, pre)
        assert 0 <= e < TOL, (streams, M, Cin, KW, dil, N, pre, e)


@pytest.mark.parametrize("g32l", ["1", "0"])
def test_igemm32l_table_free(conv, g32l):
    # igemm32l_kernel (igemm32_kernel's tiles for table-free 1x1 layers, buffer loads with scalar row offsets): a 2304-row panel wide enough for the 128 x 128
    # tile and a 3072-row panel (moved to the 128 x 64 tile) at 64 streams, ragged M; and the old kernel on the same shapes (hook RVC_G32L = 0)
    set_opt("RVC_G32L", g32l)
    for (M, Cin, N) in [(2304, 48, 111), (3072, 32, 111), (2300, 64, 111)]:
        e = conv(M, Cin, 1, 1, N, 64, 0)
        assert 0 <= e < TOL, (g32l, M, Cin, N, e)


@pytest.mark.parametrize("panel", ["1", "0"])
def test_igemm32l_panel_order(conv, panel):
    # round 6: tall table-free panels whose weights exceed an L2 run panel by panel inside every XCD (m_fast = 3: padded grid, surplus workgroups leave at once);
    # m-tile counts that divide into panels and that leave a remainder, n-tile counts that do not divide over the 8 XCDs, ragged M -- hook on and off
    set_opt("RVC_G32L_PANEL", panel)
    for (M, Cin, N, streams) in [(2304, 320, 111, 64), (3072, 256, 111, 40), (2300, 320, 74, 56), (1792, 384, 57, 74)]:
        e = conv(M, Cin, 1, 1, N, streams, 0)
        assert 0 <= e < TOL and conv.kernel() == "g32l", (panel, M, Cin, N, streams, e, conv.kernel())


@pytest.mark.parametrize("tab", ["1", "0"])
def test_igemm32l_table_variant(conv, tab):
    # ... its table variant (one-phase 1-D layers WITH an offset table, entries as scalar loads): the strided-stem shape class and a dilated layer with the
    # fused input activation on the 64-column tiles, hook on and off
    set_opt("RVC_G32L_TAB", tab)
    for (M, Cin, KW, dil, N, pre, streams) in [(512, 64, 3, 1, 700, 0, 16), (600, 32, 3, 1, 500, 0, 8), (256, 64, 3, 2, 252, 1, 64), (512, 32, 5, 1, 3000, 0, 6)]:
        e = conv(M, Cin, KW, dil, N, streams, pre)
        assert 0 <= e < TOL, (tab, M, Cin, KW, dil, N, pre, streams, e)


# every Choice the plan-time autotuner (plan.hip queue_igemm) can put on a layer: (kind, a, b)
CHOICES = [(1, 0, 0), (1, 1, 0), (1, 5, 0), (1, 2, 0)] + [(2, 0, ks) for ks in (1, 2, 3, 4, 8)] + [(3, lc, 0) for lc in (3, 4, 5, 7, 8)] + \
          [(4, cfg, ks) for cfg in (0, 1, 2, 3, 4) for ks in (1, 4, 8)]


@pytest.mark.parametrize("choice", CHOICES, ids=lambda c: "%d-%d-%d" % c)
def test_every_choice_of_the_autotuner(conv, choice):
    # rvc_set_plan_autotune times a layer's rule-based build against neighbouring choices and keeps the fastest: whatever it can pick must compute the same
    # convolution.  Test hook RVC_FORCE_CHOICE puts ONE choice on every layer (a layer outside the choice's domain keeps the rules -- still a valid build);
    # shapes: the families' own (1x1 tall / short panels, multi-tap with dilation and fused input activation, 32-channel blocks), at 6 and 20 streams
    # (the tuner works above 4 streams), ragged M and N.
    set_opt("RVC_FORCE_CHOICE", "%d,%d,%d" % choice)
    shapes = SHAPES + [(768, 256, 1, 1, 111, 0), (3072, 32, 1, 1, 111, 0), (128, 128, 7, 3, 420, 1), (64, 64, 11, 5, 350, 0), (32, 32, 3, 1, 1260, 1), (256, 64, 3, 1, 252, 1),
                       (512, 64, 3, 1, 350, 0), (100, 96, 5, 2, 333, 0)]
    seen = set()
    for streams in (6, 20):
        for (M, Cin, KW, dil, N, pre) in shapes:
            e = conv(M, Cin, KW, dil, N, streams, pre)
            seen.add(conv.kernel())
            assert 0 <= e < TOL, (choice, streams, M, Cin, KW, dil, N, pre, e, conv.kernel())
    want = {1: "c32s", 2: "g2w", 4: "reg"}.get(choice[0])
    if want:
        assert want in seen, (choice, seen)          # the choice did take effect on the layers of its domain
    else:
        assert seen & {"g32", "g32l", "g32t"}, (choice, seen)
