"""Round-5 GPU tests (through the C ABI, against the oracle): the plan cache (LRU, sized by the caller), the retrieval's hand-off
time-out recovered instead of failing the chunk, the one-launch retrieval with eight full-size streams, and end-to-end parity at the
stream counts whose tile rules the planner picks by itself (8 and 32) -- SURVEY.md section 8 rows a1, a6, a16."""
import numpy as np
import pytest

from common import BASELINE_160MS as g, derive, rms, set_opt, voice_signal, zoo
from obs_rvc_amd import weights as W

pytestmark = pytest.mark.gpu
PCM_TOL = 1e-3          # north_star: +-1e-3 RMS on the float PCM output


def _engine(z, streams=1, seed=(1234, 0)):
    from obs_rvc_amd.rvc import RvcInfer
    eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(); eng.load_model(z["model"])
    if streams > 1:
        eng.set_streams(streams)
    eng.set_noise_seed(*seed)
    return eng


def _oracle(z, seed, stream):
    from oracle import oracle as O
    o = O.OracleRvcInfer(z["data"]); o.load_contentvec(2); o.load_f0(1); o.load_model(z["model"]); o.set_noise_seed(seed, stream)
    return o


def _geometries(n):
    """n distinct call geometries of the 160 ms ring (every plugin instance has its own crossfade / context: obs-rvc/src/lib.rs:200-227)"""
    return [(g.input_buffer_16k_size, g.sample_frame_16k, g.skip_head - 2 * k, g.model_return_length + k) for k in range(n)]


def test_plan_cache_is_lru_and_sized_by_the_caller():
    # VERDICT r4 weak #9 / ADVICE r4 medium: the cache evicted in insertion order and held 8 plans, undocumented.  Ten geometries round-robin:
    # with the cache sized 10 nothing is rebuilt after the first lap; with the default 8 every call of a 10-geometry rotation is a miss (LRU's
    # worst case, documented in INTEGRATION.md) -- and a geometry that keeps being used is never the one evicted.
    z = zoo("tiny")
    eng = _engine(z)
    ora = _oracle(z, 1234, 0)
    x = voice_signal(g.input_buffer_16k_size, seed=3)
    geos = _geometries(10)
    eng.set_plan_cache(10)
    assert eng.plan_cache_info()["capacity"] == 10
    for lap in range(3):
        for (n, f, sh, rl) in geos:
            ye = eng.infer(x, f, 12, sh, rl)
            yo = ora.infer(x, f, 12, sh, rl)
            assert rms(ye - yo) < PCM_TOL
        info = eng.plan_cache_info()
        assert info["builds"] == 10 and info["cached"] == 10, (lap, info)
    # default size: the rotation through 10 misses every time ...
    eng.set_plan_cache(8)
    assert eng.plan_cache_info()["cached"] == 8
    b0 = eng.plan_cache_info()["builds"]
    for (n, f, sh, rl) in geos:
        eng.infer(x, f, 12, sh, rl); ora.infer(x, f, 12, sh, rl)
    assert eng.plan_cache_info()["builds"] > b0
    # ... but a plan that is used between the others stays (least RECENTLY used goes first; insertion order would have dropped it)
    hot = geos[0]
    eng.infer(x, hot[1], 12, hot[2], hot[3]); ora.infer(x, hot[1], 12, hot[2], hot[3])
    b1 = eng.plan_cache_info()["builds"]
    for (n, f, sh, rl) in geos[1:9]:
        eng.infer(x, f, 12, sh, rl); ora.infer(x, f, 12, sh, rl)
        eng.infer(x, hot[1], 12, hot[2], hot[3]); ora.infer(x, hot[1], 12, hot[2], hot[3])
    built = eng.plan_cache_info()["builds"] - b1
    assert built <= 8, built          # only the eight cold geometries; the hot one was never rebuilt
    ye = eng.infer(x, hot[1], 12, hot[2], hot[3]); yo = ora.infer(x, hot[1], 12, hot[2], hot[3])
    assert rms(ye - yo) < PCM_TOL
    with pytest.raises(Exception):
        eng.set_plan_cache(1)
    eng.close()


def test_batch_g_with_a_full_cache_keeps_the_plans_of_its_own_call():
    # with 8 plans cached and the first bucket's plan at the FRONT of the cache, building the second bucket's plan evicted
    # the plan just fetched (a hit did not refresh it), and a call with two geometries was rejected as "more than 8 different geometries".
    z = zoo("tiny")
    S = 3
    eng = _engine(z, S, (9, 40))
    oras = [_oracle(z, 9, 40 + s) for s in range(S)]
    q2 = derive(48000, 0.30, 0.07, 2.0, 48000)
    G0 = (g.input_buffer_16k_size, g.sample_frame_16k, g.skip_head, g.model_return_length)
    G1 = (g.input_buffer_16k_size, g.sample_frame_16k, g.skip_head - 4, g.model_return_length + 2)
    G2 = (q2.input_buffer_16k_size, q2.sample_frame_16k, q2.skip_head, q2.model_return_length)
    shifts = [12, 0, -3]

    def tick(geos, seed):
        xs = [voice_signal(q[0], seed=seed + s) for s, q in enumerate(geos)]
        ys = eng.infer_batch_g(xs, [q[1] for q in geos], shifts, [q[2] for q in geos], [q[3] for q in geos])
        for s in range(S):
            yo = oras[s].infer(xs[s], geos[s][1], shifts[s], geos[s][2], geos[s][3])
            assert rms(ys[s] - yo) < PCM_TOL, s
    tick([G0, G1, G0], 60)                  # cache: [bucket G0 (2 streams), bucket G1 (1 stream)]
    xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=70 + s) for s in range(S)])
    for k in range(6):                      # six more geometries fill the cache (8 plans); bucket G0's plan is the oldest
        eng.infer_batch(xin, g.sample_frame_16k, 0, g.skip_head - 2 * (k + 3), g.model_return_length + k + 3)
        for s in range(S):
            oras[s].infer(xin[s], g.sample_frame_16k, 0, g.skip_head - 2 * (k + 3), g.model_return_length + k + 3)
    assert eng.plan_cache_info()["cached"] == 8
    tick([G0, G2, G0], 80)                  # bucket G0: a hit at the front; bucket G2: a miss whose build must not evict it
    # more buckets than slots IS an error, and says what to do
    eng.set_plan_cache(2)
    with pytest.raises(Exception, match="plan cache"):
        tick([G0, G1, G2], 90)
    eng.close()


def test_get_knn_reports_no_rows_after_a_bucketed_call():
    # after rvc_infer_batch_g the last plan is one bucket's, in bucket-local order -- no rows instead of the wrong rows
    z = zoo("tiny")
    S = 2
    eng = _engine(z, S, (9, 40))
    index = W.make_index(3000, 48, seed=3)
    eng.load_index(index); eng.set_index_rate(0.5)
    q2 = derive(48000, 0.30, 0.07, 2.0, 48000)
    xs = [voice_signal(g.input_buffer_16k_size, seed=1), voice_signal(q2.input_buffer_16k_size, seed=2)]
    eng.infer_batch_g(xs, [g.sample_frame_16k, q2.sample_frame_16k], None, [g.skip_head, q2.skip_head], [g.model_return_length, q2.model_return_length])
    idx, dist = eng.knn(rows_cap=256)
    assert idx.shape[0] == 0
    xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=70 + s) for s in range(S)])
    eng.infer_batch(xin, g.sample_frame_16k, 0, g.skip_head, g.model_return_length)
    idx, dist = eng.knn(rows_cap=256)
    assert idx.shape == (S * g.model_return_length, 4)
    eng.close()


@pytest.fixture
def hooks():
    yield
    for h in ("RVC_KNN_LOSE_TICKET",):
        set_opt(h, None)


@pytest.mark.parametrize("preset,streams", [("tiny", 1), ("tiny", 3), ("full", 1)])
def test_retrieval_hand_off_time_out_is_recovered_not_failed(hooks, preset, streams):
    # a selector of knn_scan_select_kernel that gives up waiting (a workgroup that does not arrive: GPU shared with another
    # process) used to fail the chunk with RVC_BACKEND and rebuild every plan.  The hook makes workgroup 0 of every stream keep its ticket: the
    # hand-off can never complete, the selectors time out, and the engine recomputes the chunk's retrieval through the exhaustive scan and the
    # rest of the chunk.  The call returns normally, hits and PCM equal the oracle's, the stream state evolves as if nothing had happened, and
    # the next chunk (hook cleared) runs the one-launch form again on re-armed counters.
    z = zoo(preset)
    dim = 48 if preset == "tiny" else 768
    index = W.make_index(5000 if preset == "tiny" else 100000, dim, seed=7)
    eng = _engine(z, streams, (77, 5))
    eng.load_index(index); eng.set_index_rate(0.75)
    oras = []
    for s in range(streams):
        o = _oracle(z, 77, 5 + s); o.load_index(index); o.set_index_rate(0.75); oras.append(o)
    R = g.model_return_length

    def chunk(tick):
        xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=300 + 10 * tick + s) for s in range(streams)])
        ye = eng.infer_batch(xin, g.sample_frame_16k, 12, g.skip_head, R) if streams > 1 else eng.infer(xin[0], g.sample_frame_16k, 12, g.skip_head, R)[None]
        ie, de = eng.knn(rows_cap=streams * 64)
        for s in range(streams):
            yo = oras[s].infer(xin[s], g.sample_frame_16k, 12, g.skip_head, R)
            io, do = oras[s].knn()
            assert np.array_equal(ie[s * R:(s + 1) * R], io), (tick, s)
            assert np.allclose(de[s * R:(s + 1) * R], do, rtol=1e-4)
            assert rms(ye[s] - yo) < PCM_TOL, (tick, s, rms(ye[s] - yo))
    chunk(0)
    assert eng.retrieval_recoveries() == 0
    set_opt("RVC_KNN_LOSE_TICKET", 1)
    chunk(1)                                        # time-out -> recovered
    chunk(2)                                        # and again, on counters the recovery re-armed
    assert eng.retrieval_recoveries() == 2
    set_opt("RVC_KNN_LOSE_TICKET", None)
    chunk(3)                                        # the one-launch form again
    assert eng.retrieval_recoveries() == 2
    for s in range(streams):
        assert np.allclose(eng.pitch_cache(s), oras[s].pitch_cache(), rtol=1e-5, atol=1e-3)
    eng.close()


def test_retrieval_eight_streams_full_size():
    # VERDICT r4 weak #1b: the one-launch retrieval at 5-11 streams (no CU partition above 4 streams: the f0 branch shares the CUs, grid = (G, B)
    # with every workgroup expected to be resident) had no test at full size.  8 streams, 100 k x 768 index, hits bit-exact per stream.
    z = zoo("full")
    S = 8
    index = W.make_index(100000, 768, seed=7)
    eng = _engine(z, S, (21, 300))
    eng.load_index(index); eng.set_index_rate(0.75)
    R = g.model_return_length
    oras = {}
    for tick in range(2):
        xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=500 + 10 * tick + s) for s in range(S)])
        ye = eng.infer_batch(xin, g.sample_frame_16k, 12, g.skip_head, R)
        ie, de = eng.knn(rows_cap=S * 64)
        assert ie.shape == (S * R, 4)
        for s in (0, 3, 7):
            if s not in oras:
                o = _oracle(z, 21, 300 + s); o.load_index(index); o.set_index_rate(0.75); oras[s] = o
                for t0 in range(tick):      # (an oracle created late replays the earlier ticks: the pitch cache is state)
                    oras[s].infer(voice_signal(g.input_buffer_16k_size, seed=500 + 10 * t0 + s), g.sample_frame_16k, 12, g.skip_head, R)
            yo = oras[s].infer(xin[s], g.sample_frame_16k, 12, g.skip_head, R)
            io, do = oras[s].knn()
            assert np.array_equal(ie[s * R:(s + 1) * R], io), (tick, s)
            assert np.allclose(de[s * R:(s + 1) * R], do, rtol=1e-4)
            assert rms(ye[s] - yo) < PCM_TOL, (tick, s, rms(ye[s] - yo))
    assert eng.retrieval_recoveries() == 0
    eng.close()


@pytest.mark.parametrize("S", [8, 32])
def test_full_size_end_to_end_at_the_stream_counts_the_bench_times(S):
    # VERDICT r4 weak #1a: the tile rules change at 8 and 32 streams; the full-size end-to-end tests ran 5, 6, 16, 18 and 64 streams, the bench
    # timed 2-32 and checked nothing.  Full-size models, the planner's own choices, three streams of the batch against their oracles.
    z = zoo("full")
    eng = _engine(z, S, (6, 200))
    xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=70 + s) for s in range(S)])
    ye = eng.infer_batch(xin, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
    assert ye.shape == (S, g.model_return_size) and np.isfinite(ye).all()
    for s in (0, S // 2, S - 1):
        yo = _oracle(z, 6, 200 + s).infer(xin[s], g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        assert rms(ye[s] - yo) < PCM_TOL, (s, rms(ye[s] - yo))
    eng.close()


@pytest.mark.parametrize("force,streams", [("2,4", 3), ("1,3", 2), ("0,2", 5), ("2,1", 3), ("2,8", 1)])
def test_forced_igemm2w_end_to_end(force, streams):
    # igemm2w_kernel (register-direct 32x32x2 tiles, K split over the waves of a workgroup) forced onto every table-free 1x1 layer it can take --
    # the transformer projections with their GELU / residual epilogues, the text encoder's 1x1 layers -- through the whole model against the oracle
    # (the kernel-level test covers the tile arithmetic; this one the epilogue operands in a real plan).
    z = zoo("tiny")
    set_opt("RVC_FORCE_G2W", force)
    try:
        eng = _engine(z, streams, (3, 60))
        xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=20 + s) for s in range(streams)])
        for tick in range(2):
            ye = eng.infer_batch(xin, g.sample_frame_16k, 12, g.skip_head, g.model_return_length) if streams > 1 else eng.infer(xin[0], g.sample_frame_16k, 12, g.skip_head, g.model_return_length)[None]
            if tick == 0:
                oras = [_oracle(z, 3, 60 + s) for s in range(streams)]
            for s in range(streams):
                yo = oras[s].infer(xin[s], g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
                assert rms(ye[s] - yo) < PCM_TOL, (force, tick, s, rms(ye[s] - yo))
        eng.close()
    finally:
        set_opt("RVC_FORCE_G2W", None)


@pytest.mark.parametrize("tile,streams", [(0, 6), (1, 3), (2, 1), (2, 7)])
def test_forced_conv32s_end_to_end(tile, streams):
    # conv32s_kernel (staged 32x32x2 convolution of the many-stream decoder) forced onto every stride-1 1-D convolution whose input channels come in 32s,
    # at any stream count, through a whole model against the oracle: the five-stage toy synthesizer has a 32-channel ResBlock stage (kernel sizes
    # 3 / 7 / 11, dilations 1 / 3 / 5, fused input LeakyReLU, residual operands) and 32-channel FFN convolutions in its text encoder.  The kernel-level
    # test covers the tile arithmetic; this one the epilogue operands and the multi-phase launches of a real plan.
    import ctypes
    z = zoo("tiny", 2, "tiny5")
    set_opt("RVC_CONV32S", "2"); set_opt("RVC_CONV32S_TILE", str(tile))
    try:
        eng = _engine(z, streams, (4, 30))
        xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=40 + s) for s in range(streams)])
        oras = [_oracle(z, 4, 30 + s) for s in range(streams)]
        for tick in range(2):
            ye = eng.infer_batch(xin, 2560, 3, 200, 21) if streams > 1 else eng.infer(xin[0], 2560, 3, 200, 21)[None]
            for s in range(streams):
                yo = oras[s].infer(xin[s], 2560, 3, 200, 21)
                assert rms(ye[s] - yo) < PCM_TOL, (tile, tick, s, rms(ye[s] - yo))
        # the kernel did run: its launches are in the plan's profile
        eng.set_profile(True)
        _ = eng.infer_batch(xin, 2560, 3, 200, 21) if streams > 1 else eng.infer(xin[0], 2560, 3, 200, 21)
        lib = eng._L
        lib.rvc_debug_profile_dump.restype = ctypes.c_int
        lib.rvc_debug_profile_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        buf = ctypes.create_string_buffer(1 << 20)
        lib.rvc_debug_profile_dump(eng._h, buf, len(buf))
        assert sum(1 for ln in buf.value.decode().splitlines() if " c32s " in ln) >= 3
        eng.set_profile(False)
        eng.close()
    finally:
        set_opt("RVC_CONV32S", None); set_opt("RVC_CONV32S_TILE", None)


@pytest.mark.parametrize("version,streams", [(2, 12), (1, 24)])
def test_plugin_default_configuration_with_many_streams(version, streams):
    # The many-stream kernels of round 5 (conv32s_kernel / conv32s_buf_kernel for the decoder, igemm32l_kernel for the one-phase 1-D layers) on the families the
    # other many-stream tests do not run: the plugin's own defaults (obs-rvc/src/lib.rs:200-227: 0.30 s chunks, 40 kHz synthesizer with rates 10 * 10 * 2 * 2 --
    # other tile edges, other channel / tap geometry) and the v1 family (256-d ContentVec layer 9 + final_proj, v1 synthesizer), full size, first / middle / last
    # stream of the batch against their own oracles.
    gg = derive(48000, 0.30, 0.07, 2.0, 40000)
    z = zoo("full", version, "full40k")
    from obs_rvc_amd.rvc import RvcInfer
    from oracle import oracle as O
    eng = RvcInfer(z["data"]); eng.load_contentvec(version); eng.load_f0(); eng.load_model(z["model"]); eng.set_streams(streams); eng.set_noise_seed(11, 400)
    xin = np.stack([voice_signal(gg.input_buffer_16k_size, seed=90 + s) for s in range(streams)])
    ye = eng.infer_batch(xin, gg.sample_frame_16k, 5, gg.skip_head, gg.model_return_length)
    assert ye.shape == (streams, gg.model_return_size) and np.isfinite(ye).all()
    for s in (0, streams // 2, streams - 1):
        o = O.OracleRvcInfer(z["data"]); o.load_contentvec(version); o.load_f0(1); o.load_model(z["model"]); o.set_noise_seed(11, 400 + s)
        yo = o.infer(xin[s], gg.sample_frame_16k, 5, gg.skip_head, gg.model_return_length)
        assert rms(ye[s] - yo) < PCM_TOL, (version, s, rms(ye[s] - yo))
    eng.close()


def test_split_bf16_gemms_exploratory_mode():
    # VERDICT r4 next #8 (exploratory, never the headline): rvc_set_gemm_precision(e, 1) runs the 1-D layers with >= 128 output rows (ContentVec's
    # projections and stem, the decoder's 128- / 256-channel stages) as three bf16 matrix-core products per fp32 product (igemm_bf3_kernel) in
    # launches of >= 250 workgroups.  Full-size models, 16 streams: the launches really are the
    # split-bf16 kernel, the PCM still matches the ORACLE (fp32 / fp64 CPU arithmetic) within the north-star tolerance, and the distance to the fp32
    # engine is reported -- the error budget of the mode (2^-16 relative per product) against the fp32 path's own 3-5e-5.
    import ctypes
    z = zoo("full")
    S = 16          # (the mode takes a layer from 250 workgroups of 128 x 128: at 16 streams the 3072- and 2304-row projections, 24 launches)
    xin = np.stack([voice_signal(g.input_buffer_16k_size, seed=70 + s) for s in range(S)])
    outs = {}
    for mode in (0, 1):
        eng = _engine(z, S, (6, 200))
        eng.set_gemm_precision(mode)
        ye = eng.infer_batch(xin, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        eng.set_profile(True)
        ye2 = eng.infer_batch(xin, g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        lib = eng._L
        lib.rvc_debug_profile_dump.restype = ctypes.c_int
        lib.rvc_debug_profile_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
        buf = ctypes.create_string_buffer(1 << 20)
        lib.rvc_debug_profile_dump(eng._h, buf, len(buf))
        n_bf3 = sum(1 for ln in buf.value.decode().splitlines() if ln.split(" ", 2)[2].startswith("bf3 "))
        assert (n_bf3 >= 24) if mode else (n_bf3 == 0), (mode, n_bf3)
        outs[mode] = ye
        eng.close()
    assert np.isfinite(outs[1]).all()
    worst = 0.0
    for s in (0, 7, 15):
        yo = _oracle(z, 6, 200 + s).infer(xin[s], g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
        e32, e3 = rms(outs[0][s] - yo), rms(outs[1][s] - yo)
        worst = max(worst, e3)
        print("stream %d: PCM rms error vs oracle: fp32 path %.3e, split-bf16 path %.3e; between the two paths %.3e" % (s, e32, e3, rms(outs[1][s] - outs[0][s])))
        assert e32 < PCM_TOL and e3 < PCM_TOL, (s, e32, e3)
    assert worst < PCM_TOL


def test_v1_engine_takes_the_two_xcd_f0_partition_beside_a_v2_engine():
    # A one-stream engine of the v1 model (ContentVec stops at layer 9) runs its f0 branch on two XCDs instead of one (engine.hip configure_aux_streams;
    # stream sets with different partitions are separate members of the per-device pool).  Both engines alive in one process, interleaved, each against
    # its own oracle; then the same v1 engine forced back to one XCD (test hook RVC_F0_XCDS) -- the partition must change nothing but the timing.
    from obs_rvc_amd.rvc import RvcInfer
    from oracle import oracle as O
    z1, z2 = zoo("tiny", 1), zoo("tiny", 2)
    e2 = _engine(z2, 1, (5, 50))
    o2 = _oracle(z2, 5, 50)

    def v1_engine():
        e = RvcInfer(z1["data"]); e.load_contentvec(1); e.load_f0(); e.load_model(z1["model"]); e.set_noise_seed(6, 60)
        return e

    def v1_oracle():
        o = O.OracleRvcInfer(z1["data"]); o.load_contentvec(1); o.load_f0(1); o.load_model(z1["model"]); o.set_noise_seed(6, 60)
        return o

    x = voice_signal(g.input_buffer_16k_size, seed=31)
    a = (g.sample_frame_16k, 12, g.skip_head, g.model_return_length)
    e1, o1 = v1_engine(), v1_oracle()
    for tick in range(3):
        y1, y2 = e1.infer(x, *a), e2.infer(x, *a)
        assert rms(y1 - o1.infer(x, *a)) < PCM_TOL and rms(y2 - o2.infer(x, *a)) < PCM_TOL, tick
    e1.close()
    set_opt("RVC_F0_XCDS", "1")
    try:
        e1, o1 = v1_engine(), v1_oracle()
        for tick in range(2):
            assert rms(e1.infer(x, *a) - o1.infer(x, *a)) < PCM_TOL
            assert rms(e2.infer(x, *a) - o2.infer(x, *a)) < PCM_TOL
        e1.close()
    finally:
        set_opt("RVC_F0_XCDS", None)
    e2.close()
