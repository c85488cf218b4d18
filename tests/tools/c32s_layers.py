#!/usr/bin/env python
"""Tuning aid: conv32s layers of one chunk, summed per shape, for a list of hook settings.  usage: c32s_layers.py STREAMS "A=1,B=2" "A=3" ...   (one process per setting)"""
import os, re, subprocess, sys
from collections import OrderedDict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S = sys.argv[1]
res = OrderedDict()
for setting in sys.argv[2:]:
    args = [a for a in setting.split(",") if a and a != "-"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests/tools/op_profile.py"), S, "full"] + args, capture_output=True, text=True).stdout
    rows = []
    for ln in out.splitlines():
        m = re.match(r"\s*([\d.]+) us\s+([\d.]+) GF\s+([\d.]+) TF\s+[\d.]+%\s+(.*)", ln)
        if m: rows.append((float(m.group(1)), float(m.group(2)), m.group(4).strip()))
    res[setting] = rows
    wall = [ln for ln in out.splitlines() if ln.startswith("wall")]
    print(setting, "launches", len(rows), "sum us %.0f" % sum(r[0] for r in rows), wall[0] if wall else "", flush=True)
base = list(res.values())[0]
# decoder layers: identified in the first run by M in (32, 64, 128, 256) with a 1-D multi-tap shape in the description of any run
keys = OrderedDict()
for i, (us, gf, d) in enumerate(base):
    f = dict(re.findall(r"(\w+)=([\w.x]+)", d))
    if any(i < len(r) and r[i][2].startswith("c32s") for r in res.values()):
        n = int(f["N"]) // (int(S) if int(f.get("B", "1")) == 1 and d.split()[0] != "c32s" else 1)
        keys.setdefault((f["M"], n, f["K"]), []).append(i)
for k, idx in keys.items():
    line = "M=%s N=%s K=%s n=%d:" % (k[0], k[1], k[2], len(idx))
    for setting, rows in res.items():
        if len(rows) != len(base): line += "  [%s: launch count differs]" % setting; continue
        us = sum(rows[i][0] for i in idx); gf = sum(rows[i][1] for i in idx)
        line += "  %7.0f us %5.1f TF (%s)" % (us, gf / us * 1e3, rows[idx[0]][2].split()[0] + " " + dict(re.findall(r"(\w+)=([\w.x]+)", rows[idx[0]][2])).get("tile", ""))
    print(line)
