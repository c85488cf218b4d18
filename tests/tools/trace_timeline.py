#!/usr/bin/env python
"""Summarise a rocprofv3 kernel_trace.csv of bench.py: per-chunk span, per-queue busy time, gaps, top kernels.

usage: trace_timeline.py <kernel_trace.csv> [n_last_chunks]
A chunk boundary is the advance_chunk kernel (last kernel of every chunk's plan).
"""
import csv, sys, collections, re

def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("rvc::", "").replace("void ", "")
    return n[:60]

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
ends = [i for i, r in enumerate(rows) if "advance_chunk" in r[2]]
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 20
chunks = []
for a, b in zip(ends[:-1], ends[1:]):
    chunks.append(rows[a + 1:b + 1])
chunks = chunks[-nl:]
if not chunks:
    raise SystemExit("no chunks found")
span = []; busyq = collections.defaultdict(list); ktime = collections.defaultdict(float); kcnt = collections.Counter()
for c in chunks:
    t0 = c[0][0]; t1 = max(r[1] for r in c)
    span.append((t1 - t0) / 1e3)
    perq = collections.defaultdict(float)
    for s, e, n, q in c:
        perq[q] += (e - s) / 1e3
        ktime[short(n)] += (e - s) / 1e3; kcnt[short(n)] += 1
    for q, v in perq.items():
        busyq[q].append(v)
n = len(chunks)
print("chunks %d  launches/chunk %.0f  span us: mean %.1f min %.1f" % (n, sum(len(c) for c in chunks) / n, sum(span) / n, min(span)))
for q, v in busyq.items():
    print("  queue %s busy %.1f us/chunk" % (q, sum(v) / len(v)))
print("top kernels (us/chunk, launches/chunk, avg us):")
for k, v in sorted(ktime.items(), key=lambda kv: -kv[1])[:25]:
    print("  %8.1f %5.1f %7.2f  %s" % (v / n, kcnt[k] / n, v / kcnt[k], k))
# main-queue timeline of the last chunk: gaps > 3us
c = chunks[-1]; t0 = c[0][0]
mainq = collections.Counter(r[3] for r in c).most_common(1)[0][0]
print("last chunk, per-queue first start / last end (us):")
for q in busyq:
    cc = [r for r in c if r[3] == q]
    if cc:
        print("  queue %s: %.1f .. %.1f (%d kernels)" % (q, (cc[0][0] - t0) / 1e3, (max(r[1] for r in cc) - t0) / 1e3, len(cc)))
prev = None
print("gaps > 4 us on the main queue (at us, gap us, next kernel):")
for s, e, nme, q in c:
    if q != mainq: continue
    if prev is not None and s - prev > 4000:
        print("  %8.1f %6.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, short(nme)))
    prev = e
