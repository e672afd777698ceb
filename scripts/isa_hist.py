#!/usr/bin/env python3
"""Instruction histogram of one kernel from `hipcc -S --cuda-device-only` output.

    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -S --cuda-device-only rsik_lib.hip -o /tmp/rsik.s
    python scripts/isa_hist.py /tmp/rsik.s solve_kernelILb0
"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r"\n(_Z\w*" + pat + r"\w*):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S)
c = collections.Counter()
for line in m.group(2).splitlines():
    line = line.strip()
    if not line or line.startswith((".", ";", "//")) or line.endswith(":"):
        continue
    c[line.split()[0]] += 1
f64 = sum(v for k, v in c.items() if "f64" in k or "b64" in k and k.startswith("v_"))
print(m.group(1))
print("total", sum(c.values()), "f64/b64 VALU", f64, "VALU", sum(v for k, v in c.items() if k.startswith("v_")),
      "SALU", sum(v for k, v in c.items() if k.startswith("s_")))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"{k:28s}{v}")
