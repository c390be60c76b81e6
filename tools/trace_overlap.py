#!/usr/bin/env python3
"""Concurrency of the mapping contexts of the drop-in from a rocprofv3 --kernel-trace CSV: busy time of every context's main stream
(the three streams with the most dispatches), their pairwise overlap, the hardware queue each stream landed on, and how long 0 / 1 / 2 / ...
queues were executing a kernel.  usage: trace_overlap.py <dir with *kernel_trace.csv>"""
import collections, csv, glob, os, sys
fn = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], r.get("Stream_Id", ""), r.get("Queue_Id", "")) for r in csv.DictReader(open(fn[0])))
i0 = next((i for i, r in enumerate(rows) if r[2].startswith("k_sketch")), 0)
rows = rows[i0:]; t0 = rows[0][0]; t1 = max(r[1] for r in rows); span = t1 - t0
def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def length(u): return sum(e - s for s, e in u)
def inter(a, b):
    i = j = tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s: tot += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
cnt = collections.Counter(r[3] for r in rows)
mains = [s for s, _ in cnt.most_common(8) if cnt[s] > 0.3 * cnt.most_common(1)[0][1]]
sq = collections.defaultdict(set)
for r in rows: sq[r[3]].add(r[4])
print("span %.1f ms; main streams %s on queues %s" % (span / 1e6, mains, [sorted(sq[s]) for s in mains]))
U = {s: union((r[0], r[1]) for r in rows if r[3] == s) for s in mains}
for s in mains: print("  stream %s: %d dispatches, busy %.1f ms (%.0f %% of the span)" % (s, cnt[s], length(U[s]) / 1e6, 100 * length(U[s]) / span))
for i, a in enumerate(mains):
    for b in mains[i + 1:]: print("  overlap %s & %s: %.1f ms" % (a, b, inter(U[a], U[b]) / 1e6))
ev = []
for s, e, k, st, q in rows: ev.append((s, 1, q)); ev.append((e, -1, q))
ev.sort(); act = collections.Counter(); last = ev[0][0]; hist = collections.Counter()
for t, d, q in ev:
    hist[sum(1 for v in act.values() if v > 0)] += t - last; last = t; act[q] += d
print("queues executing at once: " + ", ".join("%d: %.0f %%" % (k, 100 * v / span) for k, v in sorted(hist.items())))
