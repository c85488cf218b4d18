#!/usr/bin/env python
"""Tuning aid: per-layer A/B of kernel / tile choices inside ONE process chain on ONE box (figures from
different boxes of the pool differ by +-1.5 %, so every comparison has to be made in one call).

usage: layer_ab.py <streams> NAME[=ENV1=V1,ENV2=V2...] ...      e.g.  layer_ab.py 8 base sq64off=RVC_NO_G32_SQ64=1 bal_off=RVC_NO_BALANCE=1

Builds nothing: needs the tuning library (python tests/tools/build_tuning.py), whose tune_env() switches read the environment.  Every
variant runs tests/tools/op_profile.py (per-launch HIP events, the two front branches one after the other) in its own process; the table
lists, per layer shape, the summed time of its launches and the kernel / tile that ran, variants side by side, the first one as the base."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(streams, envs):
    env = dict(os.environ, RVC_TUNING="1", RVC_LIB_OVERRIDE=os.path.join(ROOT, "obs_rvc_amd", "csrc", "librvc_tuning.so"))
    env.update(envs)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "op_profile.py"), str(streams), "full"], env=env, capture_output=True, text=True).stdout
    agg = collections.OrderedDict()
    wall = ""
    for ln in out.splitlines():
        m = re.match(r"\s*([\d.]+) us\s+([\d.]+) GF\s+([\d.]+) TF\s+([\d.]+)%\s+(\w+) (M=\d+ N=\d+ K=\d+) (.*)", ln)
        if m:
            t = agg.setdefault(m.group(6), [0, 0.0, set()])
            t[0] += 1; t[1] += float(m.group(1))
            tl = re.search(r"tile=(\S+)", m.group(7))
            t[2].add(m.group(5) + ":" + (tl.group(1) if tl else "?"))
        elif ln.startswith("wall"):
            wall = ln
    return agg, wall


if __name__ == "__main__":
    S = int(sys.argv[1])
    variants = []
    for a in sys.argv[2:]:
        name, _, rest = a.partition("=")
        variants.append((name, dict(kv.split("=", 1) for kv in rest.split(",") if kv)))
    res = [(n, ) + run(S, e) for n, e in variants]
    for n, _, wall in res:
        print("%-14s %s" % (n, wall))
    base = res[0][1]
    print("%-28s %4s | %s" % ("layer", "n", " | ".join("%-24s" % n for n, _, _ in res)))
    for k, (n, us, kern) in sorted(base.items(), key=lambda kv: -kv[1][1])[:32]:
        cells = []
        for _, agg, _ in res:
            cells.append("%8.1f %-15s" % (agg[k][1], ",".join(sorted(agg[k][2]))[:15]) if k in agg else "%-24s" % "-")
        print("%-28s %4d | %s" % (k, n, " | ".join(cells)))
