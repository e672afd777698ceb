#!/usr/bin/env python3
"""Instruction-mix table of one kernel from `hipcc -S --cuda-device-only` output (static counts; build with
-DRSIK_HOT_ONLY for the common path without the rare branches):

    hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DRSIK_HOT_ONLY -S --cuda-device-only rsik_lib.hip -o /tmp/hot.s
    python scripts/isa_mix.py /tmp/hot.s solve_kernelILi0ELb1

Classes: fp64 arithmetic (add / mul / fma / fmac / min / max / rndne / floor / fract ...), fp64 transcendental seeds
(v_rcp / v_rsq / v_sqrt _f64), fp64 compares, selects (v_cndmask), conversions, 64-bit moves, 32-bit moves, integer /
address arithmetic (everything else on the VALU), LDS, global / buffer memory, SALU (by kind), waits and branches."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
m = re.search(r"\n(_Z\w*" + sys.argv[2] + r"\w*):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S)
ops = collections.Counter()
for line in m.group(2).splitlines():
    line = line.strip()
    if not line or line.startswith((".", ";", "//")) or line.endswith(":"):
        continue
    ops[line.split()[0]] += 1


def klass(op):
    if op.startswith("v_"):
        if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
            return "VALU fp64 transcendental seed (v_rcp / v_rsq / v_sqrt_f64)"
        if op.startswith("v_cmp") and "f64" in op:
            return "VALU fp64 compare"
        if op.startswith("v_cmp"):
            return "VALU integer compare"
        if op.startswith("v_cndmask"):
            return "VALU select (v_cndmask)"
        if op.startswith("v_cvt"):
            return "VALU conversion"
        if "f64" in op:
            return "VALU fp64 arithmetic"
        if op.startswith("v_mov_b64") or op.startswith("v_pk_mov"):
            return "VALU 64-bit move"
        if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_readfirstlane") or op.startswith("v_readlane"):
            return "VALU 32-bit move / lane read"
        return "VALU integer / address / bit arithmetic"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "global / buffer memory"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "SALU wait / nop / barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "SALU branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "scalar memory load"
    if op.startswith("s_mov") or op.startswith("s_movk"):
        return "SALU move (literals for the polynomials, masks)"
    if op.startswith("s_"):
        return "SALU mask / integer arithmetic"
    return "other"


by = collections.defaultdict(collections.Counter)
for op, c in ops.items():
    by[klass(op)][op] += c
tot = sum(ops.values())
valu = sum(c for op, c in ops.items() if op.startswith("v_"))
print(m.group(1))
print(f"static instructions: {tot}   VALU {valu}   SALU {sum(c for op, c in ops.items() if op.startswith('s_'))}   "
      f"LDS {sum(c for op, c in ops.items() if op.startswith('ds_'))}")
for k in sorted(by, key=lambda k: -sum(by[k].values())):
    n = sum(by[k].values())
    top = ", ".join(f"{op} {c}" for op, c in by[k].most_common(6))
    print(f"{n:6d}  {100 * n / tot:5.1f} %  {k:58s} {top}")
