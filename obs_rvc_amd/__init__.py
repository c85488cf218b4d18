"""obs_rvc_amd -- MI355X-native RVC streaming inference engine behind the reference's `rvc` API.

Python package name of the `obs-rvc_amd` deliverable (a hyphen is not importable).  Contents:
csrc/ (hand-written gfx950 kernels + the C ABI), rvc.py / rvc_common.py (mirror of the reference's
`rvc` and `rvc-common` crates), weights.py (native weight blob + synthetic model zoo), geometry.py
(the caller's buffer-size formulas), dist.py (stream sharding + RCCL index broadcast)."""
from .rvc_common import PitchAlgorithm, RvcInferError, RvcModelVersion  # noqa: F401


def __getattr__(name):
    if name == "RvcInfer":
        from .rvc import RvcInfer
        return RvcInfer
    raise AttributeError(name)
