#!/usr/bin/env python3
"""Timeline of the kernels of the last few erm_run calls from a rocprofv3 --kernel-trace csv (diagnostics): start of every kernel relative to the call's first one, its
duration and the gap to its predecessor.  usage: python tools/run_timeline.py <kernel_trace.csv> [K]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void erm::", "").replace("erm::", "")[:40]
# the last run_begin_kernel followed by exactly K sweep kernels
idx = [i for i, r in enumerate(rows) if "run_begin_kernel" in r["Kernel_Name"]]
for b in idx[-3:]:
    seg = rows[b:b + K + 3]
    t0 = int(seg[0]["Start_Timestamp"])
    prev_end = None
    print("---- run")
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = "" if prev_end is None else "gap %6.2f" % ((s - prev_end) / 1e3)
        print("%-42s start %8.2f us  dur %7.2f us  %s" % (name(r), (s - t0) / 1e3, (e - s) / 1e3, gap))
        prev_end = e
        if "run_end_kernel" in r["Kernel_Name"]:
            break
