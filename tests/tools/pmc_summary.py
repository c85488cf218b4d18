#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (template instantiation) into one JSON.

usage: pmc_summary.py <out.json> <label>=<counter_collection.csv>[,<counter_collection.csv>...] ...
Counters of several passes over the same command are merged per kernel name; derived ratios are added where their inputs exist."""
import collections, csv, json, re, sys

def short(n):
    n = re.sub(r"\(.*", "", n).replace("rvc::", "").replace("void ", "")
    return n.strip()

out = {}
for spec in sys.argv[2:]:
    label, files = spec.split("=", 1)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for fi, f in enumerate(files.split(",")):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if fi == 0 and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); calls[k] += 1
    res = {}
    for k, v in acc.items():
        d = dict(v); d["dispatches"] = calls[k]
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in v: d[c + "_frac_of_wave_cycles"] = round(v[c] / wc, 4)
        if v.get("SQ_INSTS_MFMA"):
            d["valu_per_mfma"] = round(v.get("SQ_INSTS_VALU", 0.0) / v["SQ_INSTS_MFMA"], 3)
        if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
            d["l2_hit_rate"] = round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4)
        res[k] = d
    out[label] = dict(sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)))
json.dump(out, open(sys.argv[1], "w"), indent=1)
for label, res in out.items():
    print("==", label)
    for k, d in list(res.items())[:12]:
        print("  %-62s n=%5d wait_any %.2f wait_inst %.2f active %.2f valu/mfma %s l2hit %s" % (k[:62], d["dispatches"], d.get("SQ_WAIT_ANY_frac_of_wave_cycles", 0), d.get("SQ_WAIT_INST_ANY_frac_of_wave_cycles", 0), d.get("SQ_ACTIVE_INST_ANY_frac_of_wave_cycles", 0), d.get("valu_per_mfma"), d.get("l2_hit_rate")))
