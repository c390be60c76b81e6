#!/usr/bin/env python3
"""Timeline of the LAST step of a bench run from a rocprofv3 --kernel-trace CSV: every dispatch longer than a threshold with its start
(ms after the step's k_sketch), duration and stream.   usage: timeline.py <dir> [min_us] [from_ms] [to_ms]"""
import csv, glob, os, sys
fn = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", ""), r.get("Queue_Id", "")) for r in csv.DictReader(open(fn[0])))
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_sketch")]
rows = rows[starts[-1]:]; t0 = rows[0][0]
def short(n):
    import re
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)(<[^>]*>)?", n)
    return ((m.group(1) + (m.group(2) or "")) if m else n)[:58]
for s, e, k, st, q in rows:
    a = (s - t0) / 1e6
    if (e - s) / 1e3 >= min_us and lo <= a <= hi: print("%8.2f  %8.2f ms  st %-3s %s" % (a, (e - s) / 1e6, st, short(k)))
print("step span %.2f ms" % ((max(r[1] for r in rows) - t0) / 1e6))
