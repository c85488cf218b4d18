#!/usr/bin/env python
"""Measurement aid: build a VARIANT of the product library with extra compiler flags next to it (obs_rvc_amd/csrc/librvc_tuning_<name>.so, git-ignored),
for cross-library A/B runs of compile-time experiments on one box:

    python tests/tools/build_variant.py gelu -DRVC_FAST_GELU
    RVC_TUNING=1 RVC_LIB_OVERRIDE=obs_rvc_amd/csrc/librvc_tuning_gelu.so python bench.py --legs streams64 ...

(obs_rvc_amd/_native.py honours RVC_LIB_OVERRIDE only together with RVC_TUNING=1; unlike build_tuning.py this adds neither -DRVC_TUNING nor the probe stamps, so
the variant's timings are product timings.)  Prints the library's path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from obs_rvc_amd import _native  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
so = os.path.join(_native.CSRC, "librvc_tuning_%s.so" % name)
objs = _native.compile_units(extra_flags=flags)
_native.link_library(objs, so, "variant-%s-%s" % (name, _native.source_hash()))
print(so)
