#!/usr/bin/env python
"""Tuning aid: build the TUNING variant of the library next to the product (obs_rvc_amd/csrc/librvc_tuning.so, git-ignored):
-DRVC_TUNING compiles the tuning switches in (tune_env reads the environment; rvc_debug_conv_bench / rvc_debug_conv_probe exist) and
-DRVC_KPROBE adds the per-wave phase stamps.  The product library has none of this.  Tools select it with

    RVC_TUNING=1 RVC_LIB_OVERRIDE=$(python tests/tools/build_tuning.py) python tests/tools/<tool>.py ...

(obs_rvc_amd/_native.py honours RVC_LIB_OVERRIDE only together with RVC_TUNING=1).  Prints the library's path."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from obs_rvc_amd import _native  # noqa: E402

SO = os.path.join(_native.CSRC, "librvc_tuning.so")


def build(extra=()):
    flags = ["-DRVC_TUNING", "-DRVC_KPROBE"] + list(extra)
    objs = _native.compile_units(extra_flags=flags)
    _native.link_library(objs, SO, "tuning-" + _native.source_hash())
    return SO


if __name__ == "__main__":
    print(build(sys.argv[1:]))
