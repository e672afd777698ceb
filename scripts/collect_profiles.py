#!/usr/bin/env python3
"""Copies the judged artefacts of scripts/profile.sh runs (gpurun_out/<prefix>_c{2,3,4,5}) into profiles/<round>/ and
rebuilds profiles/<round>/counters.json: HBM bytes per launch and executed VALU instructions per wave from the PMC
passes, together with the build id of the library they were collected with (bench.py only quotes them for that build).

    python scripts/collect_profiles.py r02_c s1 [r02]
"""
import glob
import json
import os
import re
import shutil
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix, suffix = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r02"
dest = os.path.join(root, "profiles", rnd)
os.makedirs(dest, exist_ok=True)


def parse(path):
    blocks, cur = {}, None
    for line in open(path):
        m = re.match(r"pmc\s+(.*) grid=(\d+)", line)
        if m:
            cur = (m.group(1).strip(), int(m.group(2)))
            blocks[cur] = {}
            continue
        m = re.match(r"\s+(\w+)\s+median=\s*([\d.]+)\s+n=(\d+)", line)
        if m and cur:
            blocks[cur][m.group(1)] = float(m.group(2))
            blocks[cur]["n_" + m.group(1)] = int(m.group(3))
    return blocks


out = {"_comment": "HBM bytes per launch and executed vector instructions per wave from rocprofv3 PMC passes (FETCH_SIZE x2 "
                   "gfx950 correction + WRITE_SIZE, KiB units; SQ_INSTS_VALU / SQ_WAVES), scripts/profile.sh; see "
                   f"config*_{suffix}_summary.txt"}
want = {2: ("solve_kernel<0,", 1048576, "solve_kernel"), 3: ("control_discrete_kernel<false", 262144, "control_discrete_kernel"),
        4: ("solve_kernel<2,", 1048576, "solve_kernel<mixed>"), 5: (None, 4096, "cont_*_kernel pipeline")}
builds = set()
for c, (kn, grid, label) in want.items():
    src = os.path.join(root, "gpurun_out", f"{prefix}{c}")
    if not os.path.isdir(src):
        continue
    shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dest, f"config{c}_{suffix}_summary.txt"))
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dest, f"bench_config{c}_{suffix}.json"))
    try:
        builds.add(json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])["library_build"])
    except (ValueError, KeyError, IndexError):
        pass
    for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
        shutil.copy(f, os.path.join(dest, f"config{c}_{suffix}_kernel_stats.csv"))
    if kn is None:
        # config 5: the pipeline's kernels of one 4096 x 1000 pass together (launches per pass = launches / passes seen, one
        # start-up kernel per pass), per control step like the line's `achieved`
        blocks = {k: v for k, v in parse(os.path.join(src, "summary.txt")).items() if "cont_" in k[0] and "FETCH_SIZE" in v}
        passes = sum(v["n_FETCH_SIZE"] for (name, g), v in blocks.items() if "cont_init" in name)
        if passes:
            per_kernel, total = {}, 0.0
            for (name, g), v in blocks.items():
                b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 * v["n_FETCH_SIZE"] / passes
                key = re.search(r"cont_\w+", name).group(0)
                per_kernel[key] = per_kernel.get(key, 0) + int(b)
                total += b
            out[str(c)] = {"poses_per_gpu": grid, "kernel": label, "bytes": int(total / 1000), "bytes_per_pass": int(total),
                           "bytes_per_pass_by_kernel": per_kernel, "steps_per_pass": 1000}
            print(c, out[str(c)])
        continue
    for (name, g), v in parse(os.path.join(src, "summary.txt")).items():
        if kn in name and g == grid and "FETCH_SIZE" in v:
            out[str(c)] = {"poses_per_gpu": grid, "kernel": label, "bytes": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024),
                           "valu_per_wave": round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1),
                           "lds_instr_per_wave": round(v.get("SQ_INSTS_LDS", 0.0) / v["SQ_WAVES"], 1),
                           "lds_bank_conflict_cycles_per_wave": round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_WAVES"], 1)}
            print(c, out[str(c)])
assert len(builds) == 1, f"the runs were made with different library builds: {builds}"
out["build_id"] = builds.pop()
json.dump(out, open(os.path.join(dest, "counters.json"), "w"), indent=2)
print("build", out["build_id"])
