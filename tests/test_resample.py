"""Row f3: the plugin's two rubato `FftFixedInOut` resamplers (obs-rvc/src/lib.rs:236-242, 675, 747-749).
rubato is absent from /root/reference and the reference has no vector for it (parity unpinned, see
oracle/resample_oracle.py): the CPU tests pin the restatement on the algorithm's own properties, the GPU tests
compare the HIP polyphase kernel with the restatement through the C ABI."""
import numpy as np
import pytest

from common import voice_signal
from oracle import resample_oracle as RO

# (rate_in, rate_out, chunk_size_in): the plugin's down/up-sampler shapes at 48 kHz and 44.1 kHz hosts, 160 / 300 ms chunks
CASES = [(48000, 16000, 7680 + 960), (48000, 48000, 10080), (40000, 48000, 14000), (48000, 44100, 10080),
         (44100, 16000, 7056 + 882), (32000, 48000, 6720), (16000, 48000, 3360),
         (48000, 16000, 72000 + 960), (48000, 44100, 25600)]        # past the LDS-resident row: sample_length 1.5 s; 48k->44.1k at 0.5 s


def test_fft_sizes_follow_the_rate_ratio():
    for ri, ro, ch in CASES:
        fi, fo = RO.fft_sizes(ri, ro, ch)
        assert fi >= ch and fi * ro == fo * ri and fi - ch < ri // np.gcd(ri, ro)
    assert RO.fft_sizes(48000, 16000, 8640) == (8640, 2880)          # lib.rs:236-237 at 160 ms: result[0][160..] has 17 * 160 samples
    assert RO.fft_sizes(44100, 16000, 7938) == (7938, 2880)


def test_sinc_filter_shape():
    h = RO.make_sinc(8640, RO.cutoff(8640, 2880))
    assert abs(float(h.sum()) - 1.0) < 1e-5 and int(np.argmax(h)) == 4320 and abs(float(h[0])) < 1e-10
    assert np.allclose(h[1:], h[1:][::-1], atol=1e-7)                  # symmetric around npoints / 2
    w = RO.blackman_harris(16)
    assert abs(float(w[0]) - 6e-5) < 1e-6 and abs(float(w[8]) - 1.0) < 1e-6


@pytest.mark.parametrize("ri,ro,ch", CASES[:5])
def test_oracle_resamples_a_tone_to_a_delayed_tone(ri, ro, ch):
    r = RO.FftFixedInOut(ri, ro, ch)
    fi, fo = r.input_frames_next(), r.output_frames_max()
    f = 440.0
    x = np.sin(2 * np.pi * f * np.arange(fi * 4) / ri).astype(np.float32)
    y = np.concatenate([r.process(x[i * fi:(i + 1) * fi]) for i in range(4)])
    to = np.arange(y.size) / ro - (fi / 2) / ri                        # the filter is centred on fft_in / 2 input samples
    ref = np.sin(2 * np.pi * f * to)
    assert np.abs(y[fo:] - ref[fo:]).max() < 5e-5
    with pytest.raises(ValueError):
        r.process(x[: fi - 1])


def test_oracle_is_linear_and_stateful_overlap_add():
    r1, r2, r3 = (RO.FftFixedInOut(48000, 16000, 8640) for _ in range(3))
    a, b = voice_signal(8640 * 2, seed=1), voice_signal(8640 * 2, seed=2)
    for i in range(2):
        sl = slice(i * 8640, (i + 1) * 8640)
        ya, yb, yc = r1.process(a[sl]), r2.process(b[sl]), r3.process((a[sl] + 2 * b[sl]).astype(np.float32))
        assert np.abs(yc - (ya + 2 * yb)).max() < 2e-6
    r1.reset()
    z = r1.process(np.zeros(8640, np.float32))
    assert np.all(z == 0)                                              # no state left after reset
    y1 = r2.process(np.zeros(8640, np.float32))
    assert np.abs(y1).max() > 1e-4                                     # the tail of the previous chunk is still in the overlap


def test_downsampler_rejects_out_of_band_content():
    r = RO.FftFixedInOut(48000, 16000, 8640)
    x = np.sin(2 * np.pi * 12000.0 * np.arange(8640 * 3) / 48000).astype(np.float32)    # above the 8 kHz output Nyquist
    y = np.concatenate([r.process(x[i * 8640:(i + 1) * 8640]) for i in range(3)])
    assert np.abs(y[2880:2 * 2880]).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("ri,ro,ch", CASES)
def test_gpu_resampler_matches_the_restatement(ri, ro, ch):
    from common import zoo
    from obs_rvc_amd.resample import FftFixedInOut
    from obs_rvc_amd.rvc import RvcInfer
    from obs_rvc_amd.rvc_common import RvcInferError
    eng = RvcInfer(zoo("tiny")["data"])
    ora, gpu = RO.FftFixedInOut(ri, ro, ch), FftFixedInOut(eng, ri, ro, ch)
    fi, fo = ora.input_frames_next(), ora.output_frames_max()
    assert (gpu.input_frames_next(), gpu.output_frames_max()) == (fi, fo)
    x = voice_signal(fi * 4, seed=5)
    x[fi:fi + 100] += 0.5                                              # a step inside a chunk: exercises the filter tails / overlap
    for i in range(4):
        yo, yg = ora.process(x[i * fi:(i + 1) * fi]), gpu.process(x[i * fi:(i + 1) * fi])
        assert yg.shape == yo.shape == (fo,)
        assert np.abs(yg - yo).max() < 2e-5, (i, float(np.abs(yg - yo).max()))
    gpu.reset(); ora.reset()
    assert np.abs(gpu.process(x[:fi]) - ora.process(x[:fi])).max() < 2e-5
    with pytest.raises(RvcInferError) as ei:
        gpu.process(x[: fi - 1])
    assert ei.value.kind == "NdarrayShapeError"
    buf = np.zeros(fo + 7, np.float32)
    assert gpu.process_into_buffer(x[:fi], buf) == (fi, fo)


@pytest.mark.gpu
def test_gpu_resampler_device_api_and_tone():
    import torch
    from common import zoo
    from obs_rvc_amd.resample import FftFixedInOut
    from obs_rvc_amd.rvc import RvcInfer
    eng = RvcInfer(zoo("tiny")["data"])
    gpu = FftFixedInOut(eng, 48000, 16000, 8640)
    x = np.sin(2 * np.pi * 440.0 * np.arange(8640 * 3) / 48000).astype(np.float32)
    dx, dy = torch.from_numpy(x).cuda(), torch.zeros(3, 2880, device="cuda")
    torch.cuda.synchronize()         # (the fill runs on torch's stream, the engine on its own: the caller orders the two)
    for i in range(3):
        gpu.process_device(dx[i * 8640:].data_ptr(), dy[i].data_ptr(), sync=True)
    y = dy.cpu().numpy().reshape(-1)
    ref = np.sin(2 * np.pi * 440.0 * (np.arange(y.size) / 16000 - 4320 / 48000))
    assert np.abs(y[2880:] - ref[2880:]).max() < 5e-5
