"""Ready-to-run check against REAL model files (none exist in the build image: every test here skips unless $RVC_REAL_DATA points
at a directory laid out like the plugin's data directory).

    $RVC_REAL_DATA/contentvec/vec-768-layer-12.onnx   (or .pth / .safetensors / an already converted .rvcw)
    $RVC_REAL_DATA/f0/rmvpe.onnx | rmvpe.pt
    $RVC_REAL_DATA/model.pth | model.onnx             (any v2 RVC voice model; optional)

The one value-level vector the reference holds on the dense path is rvc/src/tests/hubert.rs:11-19:
`extract_feature(input_wav.npy)` must equal `feats.npy` to abs 2e-3 with the real ContentVec weights.  Both arrays are committed
as tests/golden/ref_input_wav.npy / ref_feats.npy (byte-identical copies of the reference's fixtures)."""
import glob
import os

import numpy as np
import pytest

from common import GOLDEN
from obs_rvc_amd import importers as IM, weights as W

REAL = os.environ.get("RVC_REAL_DATA", "")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not REAL or not os.path.isdir(REAL), reason="RVC_REAL_DATA not set: no real model files in this image")]


def _find(sub, stems):
    for stem in stems:
        for ext in (".rvcw", ".onnx", ".pth", ".pt", ".safetensors"):
            p = os.path.join(REAL, sub, stem + ext)
            if os.path.exists(p):
                return p
    return None


def _converted(kind, src, dst, **kw):
    if src.endswith(".rvcw"):
        return src
    if not os.path.exists(dst):
        IM.convert(kind, src, dst, **kw)
    return dst


def test_real_contentvec_reproduces_the_reference_feats(tmp_path):
    from obs_rvc_amd.rvc import RvcInfer
    src = _find("contentvec", ["vec-768-layer-12", "hubert_base", "checkpoint_best_legacy_500"])
    if not src:
        pytest.skip("no ContentVec file under $RVC_REAL_DATA/contentvec")
    d = tmp_path / "data"
    os.makedirs(d / "contentvec")
    blob = _converted("contentvec", src, str(d / "contentvec" / W.cv_blob_name(2)), version=2)
    if blob != str(d / "contentvec" / W.cv_blob_name(2)):
        os.symlink(blob, str(d / "contentvec" / W.cv_blob_name(2)))
    eng = RvcInfer(str(d)); eng.load_contentvec(2)
    wav = np.load(os.path.join(GOLDEN, "ref_input_wav.npy")).astype(np.float32).reshape(-1)
    ref = np.load(os.path.join(GOLDEN, "ref_feats.npy")).astype(np.float32)
    got = eng.extract_feature(wav)
    assert got.shape == ref.shape, (got.shape, ref.shape)                 # (1, 239, 768)
    assert np.abs(got - ref).max() < 2e-3, np.abs(got - ref).max()        # rvc/src/tests/hubert.rs:18 (abs_diff_eq epsilon 2e-3)


def test_real_rmvpe_tracks_a_known_tone(tmp_path):
    # no expected output is recorded in the reference (rvc/src/tests/pitch.rs only prints): with trained weights a clean 220 Hz
    # harmonic tone must come out as 220 Hz within one 20-cent bin
    from obs_rvc_amd.rvc import RvcInfer
    src = _find("f0", ["rmvpe"])
    if not src:
        pytest.skip("no RMVPE file under $RVC_REAL_DATA/f0")
    d = tmp_path / "data"
    os.makedirs(d / "f0")
    blob = _converted("rmvpe", src, str(d / "f0" / "rmvpe.rvcw"))
    if blob != str(d / "f0" / "rmvpe.rvcw"):
        os.symlink(blob, str(d / "f0" / "rmvpe.rvcw"))
    eng = RvcInfer(str(d)); eng.load_f0(1)
    t = np.arange(35840) / 16000.0
    x = sum(np.sin(2 * np.pi * 220.0 * (h + 1) * t) / (h + 1) for h in range(6)).astype(np.float32) * 0.1
    f0 = eng.pitch(x, 0, 2560)
    voiced = f0[f0 > 0]
    assert voiced.size >= f0.size // 2 and np.abs(1200 * np.log2(np.median(voiced) / 220.0)) < 20.0, f0


def test_real_voice_model_produces_bounded_audio(tmp_path):
    from obs_rvc_amd.rvc import RvcInfer
    cands = [p for p in glob.glob(os.path.join(REAL, "*.pth")) + glob.glob(os.path.join(REAL, "*.onnx")) + glob.glob(os.path.join(REAL, "*.rvcw"))]
    cv, rm = _find("contentvec", ["vec-768-layer-12"]), _find("f0", ["rmvpe"])
    if not cands or not cv or not rm:
        pytest.skip("needs a voice model next to contentvec/ and f0/ under $RVC_REAL_DATA")
    d = tmp_path / "data"
    os.makedirs(d / "contentvec"); os.makedirs(d / "f0")
    a = _converted("contentvec", cv, str(d / "contentvec" / W.cv_blob_name(2)), version=2)
    b = _converted("rmvpe", rm, str(d / "f0" / "rmvpe.rvcw"))
    for src, dst in ((a, str(d / "contentvec" / W.cv_blob_name(2))), (b, str(d / "f0" / "rmvpe.rvcw"))):
        if src != dst:
            os.symlink(src, dst)
    model = _converted("synth", cands[0], str(tmp_path / "voice.rvcw"))
    eng = RvcInfer(str(d)); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(model)
    sr = int(W.read_blob(model)[0]["sr"])
    wav = np.load(os.path.join(GOLDEN, "ref_input_wav2.npy")).astype(np.float32).reshape(-1)
    L = 35840
    x = np.zeros(L, np.float32); x[-min(L, wav.size):] = wav[-min(L, wav.size):]
    y = eng.infer(x, 2560, 0, 200, 21)
    assert y.shape == (21 * sr // 100,) and np.isfinite(y).all() and np.abs(y).max() <= 1.0 + 1e-6 and np.abs(y).max() > 1e-4
