#!/usr/bin/env python3
"""GPU occupancy of a process from a rocprofv3 --kernel-trace CSV: time from the first mapping kernel to the last, the union of
the kernel intervals inside it (busy), the sum of the kernel durations, and the kernels by total time."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not f:
    sys.exit("no kernel_trace.csv under " + d)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f[0]))]
rows.sort()
# the mapping phase starts with the first read sketch (index-build kernels come before it)
t_first = next((s for s, e, n in rows if n.startswith("k_sketch")), rows[0][0])
ph = [(s, e, n) for s, e, n in rows if s >= t_first]
t0, t1 = ph[0][0], max(e for s, e, n in ph)
busy, cur_s, cur_e = 0, ph[0][0], ph[0][1]
for s, e, n in ph[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = defaultdict(lambda: [0, 0])
for s, e, n in ph:
    k = n.split("(")[0][:80]; tot[k][0] += 1; tot[k][1] += e - s
ssum = sum(v[1] for v in tot.values())
print("# kernel timeline: %d dispatches in the mapping phase\n" % len(ph))
print("span first->last kernel: %.1f ms; GPU busy (union of kernel intervals): %.1f ms (%.1f %%); sum of kernel durations: %.1f ms (overlap factor %.2f)\n"
      % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), ssum / 1e6, ssum / max(1, busy)))
print("| kernel | calls | total ms | % of sum |\n|---|---|---|---|")
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("| %s | %d | %.2f | %.1f |" % (k, c, t / 1e6, 100.0 * t / ssum))
