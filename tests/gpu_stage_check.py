"""Diagnostic (not collected by pytest): per-stage GPU-vs-oracle errors.  Usage: python tests/gpu_stage_check.py [tiny|full]"""
import sys
import time

import numpy as np

from common import BASELINE_160MS, chunk_stream, compare_taps, rel_rms, rms, voice_signal, zoo  # noqa

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from obs_rvc_amd.rvc import RvcInfer  # noqa: E402

preset = sys.argv[1] if len(sys.argv) > 1 else "tiny"
g = BASELINE_160MS
z = zoo(preset)
ora = O.OracleRvcInfer(z["data"]); ora.load_contentvec(2); ora.load_f0(1); ora.load_model(z["model"]); ora.enable_taps(True)
eng = RvcInfer(z["data"]); eng.load_contentvec(2); eng.load_f0(1); eng.load_model(z["model"]); eng.enable_taps(True)
ora.set_noise_seed(1234, 0); eng.set_noise_seed(1234, 0)
audio = voice_signal(16000 * 2, seed=0)
rings = list(chunk_stream(audio, g.input_buffer_16k_size, g.sample_frame_16k))
worst = 0.0
for ci, ring in enumerate(rings[-3:]):
    t0 = time.time(); yo = ora.infer(ring, g.sample_frame_16k, 12, g.skip_head, g.model_return_length); to = time.time() - t0
    t0 = time.time(); ye = eng.infer(ring, g.sample_frame_16k, 12, g.skip_head, g.model_return_length); te = time.time() - t0
    err = rms(ye - yo)
    worst = max(worst, err)
    print("chunk %d: out %d samples, oracle %.3fs, engine %.4fs (gpu %.3f ms), rms(out)=%.4f rms err=%.3e max err=%.3e" % (
        ci, len(ye), to, te, eng.last_gpu_ms(), rms(yo), err, float(np.abs(ye - yo).max())))
    if ci == 0:
        R = g.model_return_length
        hints = {"rm.cnn": 32, "rm.gru": 32, "rm.sal": 32, "phone": R}
        for name, e, n in compare_taps(ora, eng, hints):
            print("   %-9s n=%8d rel_rms_err=%.3e" % (name, n, e))
    pc_o, pc_e = ora.pitch_cache(), eng.pitch_cache()
    print("   pitch cache max abs diff %.3e" % float(np.abs(pc_o - pc_e).max()))
print("WORST_RMS_ERR %.3e" % worst)
