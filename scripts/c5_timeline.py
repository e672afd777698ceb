#!/usr/bin/env python3
"""Timeline of one config-5 pass from a rocprofv3 --kernel-trace CSV: start / end of every continuous-mode kernel
relative to the pass's init kernel (shows which phases overlap).  usage: c5_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rsik::cont_" in r["Kernel_Name"] or "rsik::flags_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
inits = [k for k, r in enumerate(rows) if "cont_init" in r["Kernel_Name"]]
a = inits[-2] if len(inits) >= 2 else inits[-1]
b = inits[-1] if len(inits) >= 2 else len(rows)
t0 = int(rows[a]["Start_Timestamp"])
busy = {}
for r in rows[a:b]:
    name = r["Kernel_Name"].split("::")[1].split("<")[0]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    busy[name] = busy.get(name, 0.0) + (e - s)
    print("%-22s %9.1f -> %9.1f  (%7.1f us)" % (name, s, e, e - s))
print("sum of durations per kernel:", {k: round(v, 1) for k, v in busy.items()})
print("pass span: %.1f us" % ((max(int(r["End_Timestamp"]) for r in rows[a:b]) - t0) / 1e3))
