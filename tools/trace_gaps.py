#!/usr/bin/env python3
"""Where a mapping context's main stream idles: the gaps between consecutive kernels of the busiest stream of a rocprofv3 --kernel-trace,
grouped by (kernel before, kernel after).  usage: trace_gaps.py <dir> [min_gap_us]"""
import collections, csv, glob, os, sys
fn = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 100e3
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:40], r.get("Stream_Id", "")) for r in csv.DictReader(open(fn[0])))
i0 = next((i for i, r in enumerate(rows) if r[2].startswith("k_sketch")), 0)
rows = rows[i0:]
st = collections.Counter(r[3] for r in rows).most_common(1)[0][0]
iv = [r for r in rows if r[3] == st]
span = iv[-1][1] - iv[0][0]; busy = sum(r[1] - r[0] for r in iv)
agg = collections.defaultdict(lambda: [0, 0]); small = 0
for a, b in zip(iv, iv[1:]):
    g = b[0] - a[1]
    if g > thr: agg[(a[2], b[2])][0] += 1; agg[(a[2], b[2])][1] += g
    elif g > 0: small += g
print("stream %s: %d kernels over %.1f ms, busy %.1f ms (%.0f %%), gaps below %.0f us %.1f ms, above: %.1f ms" % (st, len(iv), span / 1e6, busy / 1e6, 100 * busy / span, thr / 1e3, small / 1e6, sum(v[1] for v in agg.values()) / 1e6))
for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:22]:
    print("%4d gaps %7.1f ms (avg %6.2f ms)  %s -> %s" % (v[0], v[1] / 1e6, v[1] / v[0] / 1e6, k[0], k[1]))
