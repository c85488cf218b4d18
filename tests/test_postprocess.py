"""SOLA offset / crossfade and RMS envelope mixing, pinned by the reference's own
golden vectors (obs-rvc/src/tests/sola.rs, envelop_mixing.rs, rt_utils.rs:139-159); and row f1, the plugin's
process_one_frame state machine around infer()."""
import os

import numpy as np
import pytest

from common import BASELINE_160MS, GOLDEN, derive, rms, voice_signal, zoo
from oracle import oracle as O


def _g(name):
    return np.load(os.path.join(GOLDEN, "ref_post_%s.npy" % name))


def test_rms_kat():
    # obs-rvc/src/rt_utils.rs:139-148 (the reference asserts exact equality)
    got = O.rms(np.arange(1, 11, dtype=np.float32), 4, 2)
    assert np.array_equal(got, np.array([1.118034, 2.738613, 4.6368093, 6.595453, 8.573215, 6.726812], np.float32))


def test_lerp_align_corners_kat():
    # rt_utils.rs:151-159
    inp = np.array([0.2353, 0.9068, 0.7870, 0.5878, 0.0097, 0.7160, 0.5812, 0.8901, 0.8822, 0.8547], np.float32)
    assert np.allclose(O.lerp_align_corners(inp, 3), [0.2353, 0.36285, 0.8547], atol=1e-7)
    exp15 = [0.2353, 0.66697854, 0.8725714, 0.79555714, 0.6731714, 0.4639215, 0.09228568, 0.36285, 0.6967429, 0.6100857,
             0.7135856, 0.8895357, 0.8844571, 0.8723786, 0.8547]
    assert np.allclose(O.lerp_align_corners(inp, 15), exp15, atol=1e-6)


def test_sola_golden_321():
    # obs-rvc/src/tests/sola.rs:11-16
    assert O.sola_offset(_g("infer_wav"), _g("sola_buffer"), 1920, 480) == 321


def test_envelop_mixing_golden():
    # obs-rvc/src/tests/envelop_mixing.rs:9-36 (zc = 480, mix rate 0.8, eps 1e-6)
    iw, ow = _g("envelop_input_wav"), _g("envelop_infer_wav")
    n = len(ow)
    r1 = O.lerp_align_corners(O.rms(iw[:n], 1920, 480), n + 1)[:n]
    r2 = np.maximum(O.lerp_align_corners(O.rms(ow, 1920, 480), n + 1), 1e-3)[:n]
    assert np.abs(r1 - _g("envelop_rms1")).max() < 1e-6 and np.abs(r2 - _g("envelop_rms2")).max() < 1e-6
    assert np.abs(O.envelop_mixing(iw, ow, 48000, 0.8) - _g("envelop_infer_wav2")).max() < 1e-6


def test_sola_step_definition():
    rng = np.random.default_rng(0)
    out = rng.standard_normal(10080).astype(np.float32) * 0.1
    sb = out[300:300 + 1920].copy() * 0.5
    off, frame, new_sb = O.sola_step(out, sb, 480, 7680)
    assert off == O.sola_offset(out, sb, 1920, 480) == 300
    fi = np.sin(np.linspace(0, 1, 1920, dtype=np.float32) * np.float32(0.5 * np.pi)) ** 2
    exp = out[off:].copy()
    exp[:1920] = exp[:1920] * fi + sb * (1 - fi)
    assert np.allclose(frame, exp[:7680], atol=1e-6) and np.allclose(new_sb, exp[7680:7680 + 1920], atol=1e-6)


@pytest.mark.gpu
def test_gpu_postprocess_matches_reference_goldens():
    from obs_rvc_amd.rvc import RvcInfer
    z = zoo("tiny")
    eng = RvcInfer(z["data"])
    iw, ow = _g("envelop_input_wav"), _g("envelop_infer_wav")
    mixed = eng.envelop_mixing(iw[:len(ow)], ow, 48000, 0.8)
    assert np.abs(mixed - _g("envelop_infer_wav2")).max() < 1e-6                      # the reference's own tolerance
    infer_wav, sola = _g("infer_wav"), _g("sola_buffer")
    off, frame, nsb = eng.sola_step(infer_wav, sola, 480, 7680)
    assert off == 321                                                                 # obs-rvc/src/tests/sola.rs:15
    o_off, o_frame, o_nsb = O.sola_step(infer_wav, sola, 480, 7680)
    assert o_off == 321 and np.allclose(frame, o_frame, atol=1e-7) and np.allclose(nsb, o_nsb, atol=1e-7)
    # edge cases: all-zero tail (division guard 1e-8), ties -> last maximum
    z0 = np.zeros(10080, np.float32)
    assert eng.sola_step(z0, np.zeros(1920, np.float32), 480, 7680)[0] == O.sola_step(z0, np.zeros(1920, np.float32), 480, 7680)[0] == 480
    with pytest.raises(Exception):
        eng.sola_step(np.zeros(5000, np.float32), np.zeros(1920, np.float32), 480, 7680)


@pytest.mark.gpu
def test_streaming_state_machine_matches_oracle():
    from obs_rvc_amd.rvc import RvcInfer
    from obs_rvc_amd.streaming import StreamingSession
    g = derive(4800, 0.16, 0.07, 2.0, 4800)          # tiny synth runs at 4.8 kHz: host rate == model rate, no resamplers needed
    z = zoo("tiny")
    eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(); eng.load_model(z["model"]); eng.set_noise_seed(3, 0)
    ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(2); ora.load_f0(1); ora.load_model(z["model"]); ora.set_noise_seed(3, 0)
    assert (g.sample_frame_16k, g.input_buffer_16k_size, g.skip_head, g.model_return_length, g.model_return_size) == (2560, 35840, 200, 21, 1008)
    se, so = StreamingSession(eng, g, 12, 0.6), StreamingSession(ora, g, 12, 0.6)
    a16 = voice_signal(2560 * 20, seed=8)
    ahost = a16[::16000 // 4800][: 768 * 20] if False else np.interp(np.arange(768 * 20) / 4800.0, np.arange(len(a16)) / 16000.0, a16).astype(np.float32)
    for c in range(20):
        fe = se.process_one_frame(ahost[c * 768:(c + 1) * 768], a16[c * 2560:(c + 1) * 2560])
        fo = so.process_one_frame(ahost[c * 768:(c + 1) * 768], a16[c * 2560:(c + 1) * 2560])
        assert fe.shape == fo.shape == (768,)
        assert se.last_sola_offset == so.last_sola_offset, c
        assert rms(fe - fo) < 1e-3, (c, rms(fe - fo))


@pytest.mark.gpu
def test_streaming_with_both_resamplers_matches_oracle():
    # the whole plugin-side chain at a 48 kHz host: rubato-style 48k -> 16k converter (with the 2*zc overlap trick of
    # lib.rs:673-679), infer, model rate (tiny synth: 4.8 kHz) -> 48 kHz converter, envelope mixing, SOLA
    from oracle import resample_oracle as RO
    from obs_rvc_amd.resample import FftFixedInOut
    from obs_rvc_amd.rvc import RvcInfer
    from obs_rvc_amd.streaming import StreamingSession
    g = derive(48000, 0.16, 0.07, 2.0, 4800)
    z = zoo("tiny")
    eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(); eng.load_model(z["model"]); eng.set_noise_seed(3, 0)
    ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(2); ora.load_f0(1); ora.load_model(z["model"]); ora.set_noise_seed(3, 0)
    assert (g.sample_frame_size, g.input_buffer_size, g.model_return_size) == (7680, 107520, 1008)
    se = StreamingSession(eng, g, 12, 0.6, 4800, lambda ri, ro, n: FftFixedInOut(eng, ri, ro, n))
    so = StreamingSession(ora, g, 12, 0.6, 4800, lambda ri, ro, n: RO.FftFixedInOut(ri, ro, n))
    assert se.downsampler.input_frames_next() == 8640 and se.upsampler.output_frames_max() == 10080
    a = np.interp(np.arange(7680 * 18) / 48000.0, np.arange(2560 * 18) / 16000.0, voice_signal(2560 * 18, seed=9)).astype(np.float32)
    for c in range(18):
        fe, fo = se.process_one_frame(a[c * 7680:(c + 1) * 7680]), so.process_one_frame(a[c * 7680:(c + 1) * 7680])
        assert fe.shape == fo.shape == (7680,)
        assert np.abs(se.input_buffer_16k - so.input_buffer_16k).max() < 5e-5
        assert rms(fe - fo) < 1e-3, (c, rms(fe - fo))
    # skip-inference mode (lib.rs:224-227, 697-699): the 16 kHz ring is passed through the 16k -> host converter
    ss = StreamingSession(eng, g, 12, 1.0, None, lambda ri, ro, n: FftFixedInOut(eng, ri, ro, n), skip_inference=True)
    sr = StreamingSession(ora, g, 12, 1.0, None, lambda ri, ro, n: RO.FftFixedInOut(ri, ro, n), skip_inference=True)
    for c in range(6):
        fe, fo = ss.process_one_frame(a[c * 7680:(c + 1) * 7680]), sr.process_one_frame(a[c * 7680:(c + 1) * 7680])
        assert rms(fe - fo) < 1e-4
    assert rms(fe) > 1e-3


def test_streaming_host_logic_with_oracle_backends():
    # CPU-only coverage of the state machine + resampler plumbing (the engine and both converters are the oracle's)
    from oracle import resample_oracle as RO
    from obs_rvc_amd.streaming import StreamingSession
    g = derive(48000, 0.16, 0.07, 2.0, 4800)
    ora = O.OracleRvcInfer(zoo("tiny")["data"])
    s = StreamingSession(ora, g, 12, 1.0, None, lambda ri, ro, n: RO.FftFixedInOut(ri, ro, n), skip_inference=True)
    assert s.model_return_size == 21 * 160 and s.upsampler.input_frames_next() == 3360 and s.upsampler.output_frames_max() == 10080
    tone = np.sin(2 * np.pi * 300.0 * np.arange(7680 * 8) / 48000.0).astype(np.float32) * 0.3
    frames = [s.process_one_frame(tone[c * 7680:(c + 1) * 7680]) for c in range(8)]
    assert all(f.shape == (7680,) for f in frames)
    # after the pipeline fills, the pass-through chain reproduces the tone's level (two converters + SOLA crossfades)
    assert abs(rms(frames[-1]) - 0.3 / np.sqrt(2)) < 0.01
    with pytest.raises(ValueError):
        StreamingSession(ora, g, 12, 1.0, 4800, None)          # no converters but model rate != host rate


@pytest.mark.gpu
def test_native_session_matches_python_state_machine_and_oracle():
    # rvc_session_process = the whole process_one_frame with device-resident rings; against the host-side state machine over the
    # same engine kernels (tight) and against the all-CPU restatement (parity tolerance)
    import time
    from oracle import resample_oracle as RO
    from obs_rvc_amd.resample import FftFixedInOut
    from obs_rvc_amd.rvc import RvcInfer
    from obs_rvc_amd.streaming import NativeStreamingSession, StreamingSession
    g = derive(48000, 0.16, 0.07, 2.0, 4800)
    z = zoo("tiny")

    def engine():
        e = RvcInfer(z["data"]); e.load_contentvec(2); e.load_f0(); e.load_model(z["model"]); e.set_noise_seed(3, 0)
        return e
    e1, e2 = engine(), engine()
    ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(2); ora.load_f0(1); ora.load_model(z["model"]); ora.set_noise_seed(3, 0)
    nat = NativeStreamingSession(e1, 48000, 0.16, 0.07, 2.0, 4800, 12, 0.6)
    assert (nat.sample_frame_size, nat.sample_frame_16k, nat.input_buffer_size, nat.input_buffer_16k_size, nat.model_return_length,
            nat.model_return_size, nat.skip_head) == (g.sample_frame_size, g.sample_frame_16k, g.input_buffer_size, g.input_buffer_16k_size,
                                                      g.model_return_length, g.model_return_size, g.skip_head)
    pys = StreamingSession(e2, g, 12, 0.6, 4800, lambda ri, ro, n: FftFixedInOut(e2, ri, ro, n))
    ors = StreamingSession(ora, g, 12, 0.6, 4800, lambda ri, ro, n: RO.FftFixedInOut(ri, ro, n))
    a = np.interp(np.arange(7680 * 16) / 48000.0, np.arange(2560 * 16) / 16000.0, voice_signal(2560 * 16, seed=10)).astype(np.float32)
    t_nat, t_py = [], []
    for c in range(16):
        ch = a[c * 7680:(c + 1) * 7680]
        t0 = time.perf_counter(); fn = nat.process_one_frame(ch); t1 = time.perf_counter(); fp = pys.process_one_frame(ch); t2 = time.perf_counter()
        fo = ors.process_one_frame(ch)
        t_nat.append(t1 - t0); t_py.append(t2 - t1)
        assert fn.shape == (7680,) and nat.last_sola_offset == pys.last_sola_offset
        assert np.abs(fn - fp).max() < 2e-5, (c, float(np.abs(fn - fp).max()))
        assert rms(fn - fo) < 1e-3, (c, rms(fn - fo))
    # fewer host round trips.  Timed in a loop of its own: above, the CPU restatement runs between the chunks and the device idles long enough
    # for whichever session comes first to pay the wake-up (medians: plan construction and the runtime's one-off queue set-up are not compared)
    t_nat, t_py = [], []
    for c in range(12):
        ch = a[(c % 16) * 7680:(c % 16 + 1) * 7680]
        t0 = time.perf_counter(); nat.process_one_frame(ch); t1 = time.perf_counter(); pys.process_one_frame(ch); t2 = time.perf_counter()
        t_nat.append(t1 - t0); t_py.append(t2 - t1)
    # (a sanity bound, not a benchmark: late in a long test process two live engines share hardware queues with whatever earlier tests left in the
    # stream pool, and either session can come out 30 % ahead; alone in a process the native session takes 0.66 ms against 1.03: tests/tools/session_time.py)
    assert np.median(t_nat[2:]) < 2.0 * np.median(t_py[2:])
    with pytest.raises(Exception):
        nat.process_one_frame(a[:100])
    wrong = NativeStreamingSession(e1, 48000, 0.16, 0.07, 2.0, 40000)   # the loaded (tiny) synthesizer runs at 4.8 kHz, not 40 kHz
    with pytest.raises(Exception):
        wrong.process_one_frame(a[:7680])
    assert NativeStreamingSession(e1, 44100, 0.16, 0.07, 2.0, 40000).sample_frame_size == 7056
    sk = NativeStreamingSession(e1, 48000, 0.16, 0.07, 2.0, 4800, 12, 1.0, skip_inference=True)
    tone = (0.3 * np.sin(2 * np.pi * 300.0 * np.arange(7680 * 8) / 48000.0)).astype(np.float32)
    last = [sk.process_one_frame(tone[c * 7680:(c + 1) * 7680]) for c in range(8)][-1]
    assert abs(rms(last) - 0.3 / np.sqrt(2)) < 0.01


@pytest.mark.gpu
def test_native_session_batches_streams():
    # 3 streams through ONE batched session (rings / converter states / SOLA tails with a leading stream axis, one infer_batch per tick)
    # must equal 3 independent single-stream sessions
    from obs_rvc_amd.rvc import RvcInfer
    from obs_rvc_amd.streaming import NativeStreamingSession
    z = zoo("tiny")

    def engine(streams, sid0):
        e = RvcInfer(z["data"]); e.load_contentvec(2); e.load_f0(); e.load_model(z["model"]); e.set_streams(streams); e.set_noise_seed(3, sid0)
        return e
    S = 3
    eb = engine(S, 0)
    batched = NativeStreamingSession(eb, 48000, 0.16, 0.07, 2.0, 4800, 12, 0.6)
    singles = []
    for sidx in range(S):
        e1 = engine(1, sidx)
        singles.append((e1, NativeStreamingSession(e1, 48000, 0.16, 0.07, 2.0, 4800, 12, 0.6)))
    F = batched.sample_frame_size
    audio = [np.interp(np.arange(F * 10) / 48000.0, np.arange(2560 * 10) / 16000.0, voice_signal(2560 * 10, seed=30 + sidx)).astype(np.float32) for sidx in range(S)]
    for c in range(10):
        x = np.stack([a[c * F:(c + 1) * F] for a in audio])
        yb = batched.process_one_frame(x)
        assert yb.shape == (S, F)
        for sidx in range(S):
            y1 = singles[sidx][1].process_one_frame(x[sidx])
            assert np.abs(yb[sidx] - y1).max() < 2e-5, (c, sidx, float(np.abs(yb[sidx] - y1).max()))
            assert batched.last_sola_offsets[sidx] == singles[sidx][1].last_sola_offset
    with pytest.raises(Exception):
        batched.process_one_frame(x[0])               # one stream handed to a 3-stream session


@pytest.mark.gpu
def test_native_session_per_stream_settings():
    # rvc_session_set_params_stream: three streams of one batched session with their own pitch shift and RMS mix rate (one of them 1.0 =
    # no envelope mixing) must equal three single-stream sessions with those settings; settings changed mid-stream
    from obs_rvc_amd.rvc import RvcInfer
    from obs_rvc_amd.streaming import NativeStreamingSession
    z = zoo("tiny")

    def engine(streams, sid0):
        e = RvcInfer(z["data"]); e.load_contentvec(2); e.load_f0(); e.load_model(z["model"]); e.set_streams(streams); e.set_noise_seed(3, sid0)
        return e
    S = 3
    params = [(12, 0.6), (0, 1.0), (-12, 0.25)]
    eb = engine(S, 0)
    batched = NativeStreamingSession(eb, 48000, 0.16, 0.07, 2.0, 4800, 12, 0.6)
    for sidx, (ps, mix) in enumerate(params):
        batched.set_params(ps, mix, stream=sidx)
    singles = []
    for sidx, (ps, mix) in enumerate(params):
        e1 = engine(1, sidx)
        singles.append((e1, NativeStreamingSession(e1, 48000, 0.16, 0.07, 2.0, 4800, ps, mix)))
    F = batched.sample_frame_size
    audio = [np.interp(np.arange(F * 8) / 48000.0, np.arange(2560 * 8) / 16000.0, voice_signal(2560 * 8, seed=60 + sidx)).astype(np.float32) for sidx in range(S)]
    for c in range(8):
        if c == 5:                                    # stream 1 changes its settings mid-stream, the others keep theirs
            batched.set_params(7, 0.5, stream=1); singles[1][1].set_params(7, 0.5)
        x = np.stack([a[c * F:(c + 1) * F] for a in audio])
        yb = batched.process_one_frame(x)
        for sidx in range(S):
            y1 = singles[sidx][1].process_one_frame(x[sidx])
            assert np.abs(yb[sidx] - y1).max() < 2e-5, (c, sidx, float(np.abs(yb[sidx] - y1).max()))
            assert batched.last_sola_offsets[sidx] == singles[sidx][1].last_sola_offset
    with pytest.raises(ValueError):
        batched.set_params(0, 1.0, stream=3)


@pytest.mark.gpu
def test_native_session_at_the_plugin_maximum_settings():
    # 1.5 s chunks: the 72 960-sample downsampler chunk no longer fits the LDS-resident polyphase row (global-memory row fallback);
    # native session vs the host-side state machine over a second engine (tight), and vs the all-CPU chain while the SOLA offsets
    # agree (a near-tie in the 480-lag search may legitimately flip between fp32 summation orders)
    from oracle import resample_oracle as RO
    from obs_rvc_amd.resample import FftFixedInOut
    from obs_rvc_amd.rvc import RvcInfer
    from obs_rvc_amd.streaming import NativeStreamingSession, StreamingSession
    g = derive(48000, 1.5, 0.15, 5.0, 4800)
    z = zoo("tiny")

    def engine():
        e = RvcInfer(z["data"]); e.load_contentvec(2); e.load_f0(); e.load_model(z["model"]); e.set_noise_seed(3, 0)
        return e
    e1, e2 = engine(), engine()
    ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(2); ora.load_f0(1); ora.load_model(z["model"]); ora.set_noise_seed(3, 0)
    nat = NativeStreamingSession(e1, 48000, 1.5, 0.15, 5.0, 4800, 12, 0.6)
    assert (nat.sample_frame_size, nat.input_buffer_size, nat.model_return_length) == (72000, 319680, 155)
    pys = StreamingSession(e2, g, 12, 0.6, 4800, lambda ri, ro, n: FftFixedInOut(e2, ri, ro, n))
    ors = StreamingSession(ora, g, 12, 0.6, 4800, lambda ri, ro, n: RO.FftFixedInOut(ri, ro, n))
    a = np.interp(np.arange(72000 * 3) / 48000.0, np.arange(24000 * 3) / 16000.0, voice_signal(24000 * 3, seed=12)).astype(np.float32)
    report = []
    for c in range(3):
        ch = a[c * 72000:(c + 1) * 72000]
        fn, fp, fo = nat.process_one_frame(ch), pys.process_one_frame(ch), ors.process_one_frame(ch)
        assert fn.shape == fo.shape == (72000,) and nat.last_sola_offset == pys.last_sola_offset
        assert np.abs(fn - fp).max() < 5e-5, (c, float(np.abs(fn - fp).max()))
        assert np.abs(pys.input_buffer_16k - ors.input_buffer_16k).max() < 5e-5
        # Two steps of the chain are discrete decisions: the decode's arg-max over 360 salience bins and the SOLA lag search.  On a
        # near-tie (flat synthetic salience) two fp32 summation orders may pick neighbouring bins; that frame's f0 then moves by one
        # 20-cent bin and stays in the pitch cache.  Every chunk is judged: either all decisions agree and the audio matches to the
        # parity tolerance, or each disagreement is NAMED and must be exactly such a one-bin / voicing-threshold move.
        ce, co = e1.pitch_cache(), ora.pitch_cache()
        flips = np.nonzero(np.abs(ce - co) >= 0.5)[0]
        sola_flip = nat.last_sola_offset != ors.last_sola_offset
        if flips.size == 0 and not sola_flip:
            assert rms(fn - fo) < 1e-3, (c, rms(fn - fo))
            report.append((c, "all decisions agree"))
            continue
        named = []
        for i in flips:
            if ce[i] > 0 and co[i] > 0:
                cents = abs(1200.0 * np.log2(ce[i] / co[i]))
                assert cents < 25.0, "chunk %d: pitch cache[%d] differs by %.1f cents (engine %.3f Hz, oracle %.3f Hz): more than one salience bin" % (c, i, cents, ce[i], co[i])
                named.append("cache[%d] one-bin arg-max flip (%.1f cents)" % (i, cents))
            else:
                named.append("cache[%d] voicing-threshold flip (%.3f vs %.3f Hz)" % (i, ce[i], co[i]))
        if sola_flip:
            named.append("SOLA offset %d vs %d" % (nat.last_sola_offset, ors.last_sola_offset))
        assert flips.size <= 4, "chunk %d: %d pitch-cache entries differ -- not isolated near-ties: %s" % (c, flips.size, named)
        report.append((c, named))
    assert report[0][1] == "all decisions agree", report          # the first 1.5 s chunk has had no earlier decision to inherit
    print("decision report:", report)


@pytest.mark.gpu
def test_native_session_full_size_v2_48k_matches_oracle_chain():
    # rvc_session_process on the FULL model (v2-768 ContentVec + RMVPE + v2-48k synthesizer, BASELINE's 160 ms geometry): the whole
    # plugin-side chain as one native call against the all-CPU chain (oracle infer + numpy/scipy resamplers + reference post-processing)
    from oracle import resample_oracle as RO
    from obs_rvc_amd.rvc import RvcInfer
    from obs_rvc_amd.streaming import NativeStreamingSession, StreamingSession
    g = derive(48000, 0.16, 0.07, 2.0, 48000)
    z = zoo("full")
    eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(); eng.load_model(z["model"]); eng.set_noise_seed(5, 0)
    ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(2); ora.load_f0(1); ora.load_model(z["model"]); ora.set_noise_seed(5, 0)
    nat = NativeStreamingSession(eng, 48000, 0.16, 0.07, 2.0, 48000, 12, 0.75)
    assert (nat.sample_frame_size, nat.input_buffer_16k_size, nat.model_return_length, nat.model_return_size, nat.skip_head) == (7680, 35840, 21, 10080, 200)
    ors = StreamingSession(ora, g, 12, 0.75, 48000, lambda ri, ro, n: RO.FftFixedInOut(ri, ro, n))
    n_ch = 18                         # the 2.24 s ring is full of audio from chunk 14 on
    a = np.interp(np.arange(7680 * n_ch) / 48000.0, np.arange(2560 * n_ch) / 16000.0, voice_signal(2560 * n_ch, seed=21)).astype(np.float32)
    from obs_rvc_amd.rvc_common import RvcInferError
    compared = full = settle = 0
    prev_same = False
    for c in range(n_ch):
        ch = a[c * 7680:(c + 1) * 7680]
        try:
            fn = nat.process_one_frame(ch)
            pe = None
        except RvcInferError as ex:
            pe = ex
        try:
            fo = ors.process_one_frame(ch)
            po = None
        except Exception as ex:
            po = ex
        # a chunk on which the reference would panic (rmvpe.rs:124, flat synthetic salience on a mostly-zero ring) must do so on both sides
        assert (pe is None) == (po is None), (c, pe, po)
        if pe is not None:
            # the reference process dies here and the plugin respawns it with fresh state (obs-rvc/src/lib.rs:716-727); the engine
            # reports RVC_PANIC after the chunk has run, so its pitch cache / chunk counter have moved: restart both sides' state
            assert "Panic" in str(pe)
            eng.reset_state(); ora.reset_state()
            prev_same = False
            settle = 2        # the native chain ran to its end on that chunk (converter overlap, SOLA tail), the CPU chain stopped at infer
            continue
        if settle:
            settle -= 1
            continue
        assert fn.shape == fo.shape == (7680,)
        if np.abs(eng.pitch_cache() - ora.pitch_cache()).max() >= 0.5:
            continue                                   # an arg-max near-tie flipped (see the maximum-settings test): audio not comparable
        # The synthetic synthesizer emits noise-like audio, so the 480-lag SOLA search has no dominant peak and the two fp32 summation
        # orders rarely pick the same lag.  Everything else is compared exactly where it must agree: the frame is
        # output[off .. off + 7680] of the envelope-mixed, upsampled model output, cross-faded over its first sola_buffer_frame_size
        # samples -- behind the cross-fade the two frames are the same signal shifted by the difference of the two offsets.
        oe, oo, xf = nat.last_sola_offset, ors.last_sola_offset, g.sola_buffer_frame_size
        d = oo - oe
        a_e = fn[xf + max(d, 0):7680 + min(d, 0)]
        a_o = fo[xf + max(-d, 0):7680 + min(-d, 0)]
        assert a_e.shape == a_o.shape and a_e.size > 4000
        assert rms(a_e - a_o) < 1e-3, (c, oe, oo, rms(a_e - a_o), rms(a_o))
        compared += 1
        if oe == oo and prev_same:
            assert rms(fn - fo) < 1e-3, (c, rms(fn - fo))         # same lag twice in a row: the cross-faded head agrees too
            full += 1
        prev_same = oe == oo
    assert compared >= 10, (compared, full)
