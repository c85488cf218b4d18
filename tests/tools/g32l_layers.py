#!/usr/bin/env python
"""Tuning aid: the table-free 1x1 layers of one chunk on igemm32_kernel (RVC_G32L=0) against igemm32l_kernel, summed per shape.  usage: g32l_layers.py STREAMS"""
import os, re, subprocess, sys
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = sys.argv[1]
res = OrderedDict()
for setting in ("RVC_G32L=0", "RVC_G32L=1"):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests/tools/op_profile.py"), S, "full", setting], capture_output=True, text=True).stdout
    rows = []
    for ln in out.splitlines():
        m = re.match(r"\s*([\d.]+) us\s+([\d.]+) GF\s+([\d.]+) TF\s+[\d.]+%\s+(.*)", ln)
        if m: rows.append((float(m.group(1)), float(m.group(2)), m.group(4).strip()))
    res[setting] = rows
    wall = [ln for ln in out.splitlines() if ln.startswith("wall")]
    print(setting, "launches", len(rows), "sum us %.0f" % sum(r[0] for r in rows), wall[0] if wall else "", flush=True)
a, b = res["RVC_G32L=0"], res["RVC_G32L=1"]
keys = OrderedDict()
if len(a) == len(b):
    for i, (us, gf, d) in enumerate(b):
        if d.startswith("g32l"):
            f = dict(re.findall(r"(\w+)=([\w.x]+)", d))
            keys.setdefault((f["M"], f["N"], f["K"], f["tile"]), []).append(i)
    for k, idx in keys.items():
        ua = sum(a[i][0] for i in idx); ub = sum(b[i][0] for i in idx); gf = sum(b[i][1] for i in idx)
        print("M=%s N=%s K=%s tile=%s n=%d: %7.0f us %5.1f TF -> %7.0f us %5.1f TF" % (k + (len(idx), ua, gf / ua * 1e3, ub, gf / ub * 1e3)))
