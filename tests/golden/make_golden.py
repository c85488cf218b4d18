"""Regenerates tests/golden/*.  Run in the build container (needs /root/reference):
    python tests/golden/make_golden.py
* ref_*.npy        -- data files held by the reference's own tests (rvc/src/tests/*.npy), copied verbatim
* ref_post_*.npy   -- data files of the reference's post-processing tests (obs-rvc/src/tests/*.npy), copied verbatim
* ref_stft_kat.npy -- the 9x4 torch.stft table quoted in rvc/src/f0/rmvpe.rs:278-288
* tiny_chain.npz   -- oracle outputs (C restatement, tiny synthetic zoo) for a 4-chunk stream; the GPU tests
                      compare the HIP path against these committed vectors as well as against the live oracle
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"

for name in ("input_wav.npy", "input_wav2.npy", "feats.npy"):
    shutil.copyfile(os.path.join(REF, "rvc/src/tests", name), os.path.join(HERE, "ref_" + name))

for name in ("infer_wav", "sola_buffer", "envelop_input_wav", "envelop_infer_wav", "envelop_rms1", "envelop_rms2", "envelop_infer_wav2"):
    shutil.copyfile(os.path.join(REF, "obs-rvc/src/tests", name + ".npy"), os.path.join(HERE, "ref_post_" + name + ".npy"))

np.save(os.path.join(HERE, "ref_stft_kat.npy"), np.array([
    [3.7801e-02, 2.5651e+00, 5.1303e+00, 7.6954e+00], [5.7373e-03, 1.2829e+00, 2.5653e+00, 3.8478e+00],
    [1.4787e-02, 6.7956e-03, 6.7958e-03, 6.7957e-03], [3.2463e-03, 1.6874e-03, 1.6874e-03, 1.6875e-03],
    [2.3478e-03, 6.6042e-04, 6.6042e-04, 6.6054e-04], [1.4494e-03, 3.1195e-04, 3.1202e-04, 3.1184e-04],
    [1.2455e-03, 1.5500e-04, 1.5485e-04, 1.5491e-04], [1.0416e-03, 6.5722e-05, 6.5798e-05, 6.5790e-05],
    [1.0417e-03, 0.0000e+00, 0.0000e+00, 2.3842e-07]], np.float32))

from common import BASELINE_160MS as g, chunk_stream, voice_signal, zoo  # noqa: E402
from oracle import oracle as O  # noqa: E402

z = zoo("tiny")
ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(2); ora.load_f0(1); ora.load_model(z["model"]); ora.set_noise_seed(99, 5)
audio = voice_signal(g.sample_frame_16k * 18, seed=11)
rings = list(chunk_stream(audio, g.input_buffer_16k_size, g.sample_frame_16k))[-4:]
outs = [ora.infer(r, g.sample_frame_16k, 7 if i % 2 else -12, g.skip_head, g.model_return_length) for i, r in enumerate(rings)]
feat = ora.extract_feature(rings[0])
f0 = ora.pitch(rings[0], 12, g.sample_frame_16k)
np.savez_compressed(os.path.join(HERE, "tiny_chain.npz"), audio=audio, outs=np.stack(outs), feat=feat, f0=f0, cache=ora.pitch_cache())
print("golden written:", sorted(os.listdir(HERE)))
