#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection.csv) -> profiles/<round>_pmc_traffic.json.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv|-> <out.json> [note] [config json] [algorithmic bytes per chunk]
config json = the configuration the pass was taken on, e.g. {"streams": 64, "index": false, "version": 2, "preset": "full"}: bench.py uses a
pass only for exactly that configuration (and exactly that build).
Per the MI355X guide's HBM section: FETCH_SIZE is in KiB and reads exactly half of a wide coalesced stream on gfx950, so
HBM read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is left uncalibrated.  A chunk ends at advance_chunk_kernel."""
import csv, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from obs_rvc_amd import _native


def load(path, counter):
    rows = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"]), int(r.get("Grid_Size") or 0) // max(1, int(r.get("Workgroup_Size") or 1))))
    rows.sort()
    return rows


def per_class(rows):
    ends = [i for i, r in enumerate(rows) if "advance_chunk" in r[1]]
    if len(ends) < 3:
        raise SystemExit("not enough chunks in the trace")
    body = rows[ends[1] + 1: ends[-1] + 1]          # skip warm-up chunk(s)
    n_chunks = len(ends) - 2
    acc = collections.defaultdict(lambda: [0, 0.0])
    for _, name, v, _g in body:
        key = "igemm_all_instantiations" if ("igemm_kernel" in name or "igemm_lds_kernel" in name or "igemm2_kernel" in name or "igemm32_kernel" in name or "igemm32l_kernel" in name or "igemm32w_kernel" in name or "conv_tile_kernel" in name or "igemm2w_kernel" in name or "conv32s_kernel" in name or "conv32s_buf_kernel" in name or "rm_block_kernel" in name) else ("knn_scan_select_kernel" if "knn_scan_select_kernel" in name else None)
        if key:
            acc[key][0] += 1; acc[key][1] += v
    return n_chunks, acc


def by_kernel(rows, top=24):
    """every kernel of the profiled chunks by (name, workgroups): where the bytes go"""
    ends = [i for i, r in enumerate(rows) if "advance_chunk" in r[1]]
    body = rows[ends[1] + 1: ends[-1] + 1]
    n_chunks = len(ends) - 2
    acc = collections.defaultdict(lambda: [0, 0.0])
    for _, name, v, g in body:
        short = name.replace("void rvc::", "").replace("rvc::", "").split("(")[0]
        acc[(short, g)][0] += 1; acc[(short, g)][1] += v
    return n_chunks, acc


fetch = load(sys.argv[1], "FETCH_SIZE")
nf, fa = per_class(fetch)
wa = None
if sys.argv[2] != "-":
    nw, wa = per_class(load(sys.argv[2], "WRITE_SIZE"))
out = {"source": (sys.argv[4] if len(sys.argv) > 4 else "") + " rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); FETCH_SIZE is in KiB and reads "
       "exactly half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM section), so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is uncalibrated",
       "chunks_profiled": nf,
       # the library the counters were taken on: bench.py puts these figures into roofline.traffic only when its own library carries the same hash
       "build": _native.binary_hash(),
       "config": (json.loads(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5] else None),
       "kernel_class": "igemm_all_instantiations = igemm_kernel + igemm2_kernel + igemm2w_kernel + igemm_lds_kernel + igemm32_kernel + igemm32l_kernel + conv_tile_kernel + conv32s_kernel + conv32s_buf_kernel + rm_block_kernel (the launches roofline.launches_per_step counts)"}
for key, (n, kib) in fa.items():
    d = {"launches_per_chunk": n / nf, "fetch_size_kib_per_chunk": kib / nf, "hbm_read_bytes_per_chunk": 2 * 1024 * kib / nf,
         "hbm_read_bytes_per_launch": 2 * 1024 * kib / n}
    if wa and key in wa:
        d["write_size_kib_per_chunk"] = wa[key][1] / nw
    if key == "igemm_all_instantiations":
        d["algorithmic_weight_bytes_per_chunk"] = 852778176
        if len(sys.argv) > 6:      # weights once + every layer's input and output once, per step
            d["algorithmic_bytes_per_chunk_estimate"] = float(sys.argv[6])
    else:
        d["algorithmic_bytes_per_launch"] = 307200000
    out[key] = d
# the same passes per (kernel, workgroups): read bytes with the x2 correction, WRITE_SIZE raw KiB -> bytes
nb, fk = by_kernel(fetch)
wk = by_kernel(load(sys.argv[2], "WRITE_SIZE"))[1] if sys.argv[2] != "-" else {}
tot = sum(v[1] for v in fk.values()) or 1.0
out["by_kernel_note"] = "all kernels of a chunk by (kernel, workgroups), largest readers first; read_mb = 2 * FETCH_SIZE KiB * 1024 / 1e6 per launch, write_mb = WRITE_SIZE KiB * 1024 / 1e6 per launch (uncalibrated)"
out["by_kernel"] = [{"kernel": k[0], "workgroups": k[1], "launches_per_chunk": round(n / nb, 2), "read_mb_per_launch": round(2 * 1024 * kib / n / 1e6, 2),
                     "write_mb_per_launch": (round(1024 * wk[k][1] / wk[k][0] / 1e6, 2) if k in wk and wk[k][0] else None), "share_of_reads": round(kib / tot, 4)}
                    for k, (n, kib) in sorted(fk.items(), key=lambda kv: -kv[1][1])[:40]]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
