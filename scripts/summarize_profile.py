#!/usr/bin/env python3
"""Condenses the rocprofv3 output of scripts/profile.sh into a short text summary (kept under profiles/)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KEY = ("solve_kernel", "control_discrete_kernel", "continuous")


def ours(name):
    return "rsik::" in name


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


print(f"# profile summary: {os.path.basename(out)}")
# kernel-trace stats
for r in rows("trace/**/*_kernel_stats.csv"):
    if ours(r["Name"]):
        print(f"stats  {r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):10.0f} min_ns={r['MinNs']} max_ns={r['MaxNs']}")
# per-grid-size durations (separates the timed launches from the input-generation launches)
dur = defaultdict(list)
meta = {}
for r in rows("trace/**/*_kernel_trace.csv"):
    if ours(r["Kernel_Name"]):
        k = (r["Kernel_Name"][:60], r["Grid_Size_X"])
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
for k, v in sorted(dur.items()):
    v = sorted(v)
    print(f"trace  {k[0]:60s} grid={k[1]:>9s} n={len(v):3d} median_ns={v[len(v)//2]:8d} min_ns={v[0]:8d}  vgpr/sgpr/lds/scratch={meta[k]}")
# PMC
pmc = defaultdict(lambda: defaultdict(list))
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_misc"):
    for r in rows(f"{d}/**/*_counter_collection.csv"):
        if ours(r["Kernel_Name"]):
            pmc[(r["Kernel_Name"][:60], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(pmc.items()):
    print(f"pmc    {k[0]} grid={k[1]}")
    for c, v in sorted(cs.items()):
        v = sorted(v)
        print(f"         {c:24s} median={v[len(v)//2]:16.1f} n={len(v)}")
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f = sorted(cs["FETCH_SIZE"])[len(cs["FETCH_SIZE"]) // 2]
        w = sorted(cs["WRITE_SIZE"])[len(cs["WRITE_SIZE"]) // 2]
        # guide: FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream -> x2
        print(f"         HBM traffic per launch (KiB units, FETCH x2 gfx950 correction): fetch={2*f*1024:.0f} B write={w*1024:.0f} B total={(2*f+w)*1024:.0f} B")
