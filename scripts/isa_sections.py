#!/usr/bin/env python3
"""Vector-instruction count per named section of one kernel, from an -DRSIK_ISA_MARKS -DRSIK_HOT_ONLY assembly:

    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DRSIK_ISA_MARKS -DRSIK_HOT_ONLY -S --cuda-device-only \
        rsik_lib.hip -o /tmp/marks.s
    python scripts/isa_sections.py /tmp/marks.s solve_kernelILb0
"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
m = re.search(r"\n(_Z\w*" + sys.argv[2] + r"\w*):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S)
sec, order = "prologue", ["prologue"]
cnt = collections.defaultdict(collections.Counter)
for line in m.group(2).splitlines():
    line = line.strip()
    mm = re.match(r";\s*RSIK_MARK (\w+)", line)
    if mm:
        sec = mm.group(1)
        if sec not in order:
            order.append(sec)
        continue
    if not line or line.startswith((".", ";", "//")) or line.endswith(":"):
        continue
    op = line.split()[0]
    cnt[sec]["all"] += 1
    if op.startswith("v_"):
        cnt[sec]["valu"] += 1
        if "f64" in op:
            cnt[sec]["f64"] += 1
            if op.startswith(("v_rsq", "v_rcp", "v_sqrt")):
                cnt[sec]["trans"] += 1
    elif op.startswith("ds_"):
        cnt[sec]["lds"] += 1
    elif op.startswith("s_"):
        cnt[sec]["salu"] += 1
print("%-22s %6s %6s %6s %5s %5s" % ("section", "VALU", "f64", "trans", "LDS", "SALU"))
tot = collections.Counter()
for s_ in order:
    c = cnt[s_]
    tot.update(c)
    print("%-22s %6d %6d %6d %5d %5d" % (s_, c["valu"], c["f64"], c["trans"], c["lds"], c["salu"]))
print("%-22s %6d %6d %6d %5d %5d" % ("total", tot["valu"], tot["f64"], tot["trans"], tot["lds"], tot["salu"]))
