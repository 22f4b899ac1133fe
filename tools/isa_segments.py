#!/usr/bin/env python3
"""Instruction mix of one kernel from hipcc's -S output, split at every s_barrier: isa_segments.py file.s kernel-substring
(v_slow = quarter-rate / 64-bit VALU: integer multiplies, reciprocals, f64)"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and ":" in l and "@" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk"): return "v_pk"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64", "v_rcp", "v_sqrt", "v_rsq", "v_cvt_f64", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_div", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")): return "v_slow"
    if op.startswith(("v_mov", "v_accvgpr")): return "v_mov"
    if op.startswith("v_"): return "v"
    if op.startswith("ds_"): return "ds"
    if op.startswith(("global_", "buffer_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "s"
    return "other"


segs, cur, name = [], collections.Counter(), "entry"
for i in range(start + 1, end):
    l = lines[i].strip()
    if not l or l.startswith((".", ";")) or lines[i].startswith(".LBB"):
        continue
    m = re.match(r"([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    if op == "s_barrier":
        segs.append((name, cur))
        cur, name = collections.Counter(), "after barrier @%d" % i
        continue
    cur[cls(op)] += 1
segs.append((name, cur))
for name, c in segs:
    print("%-26s" % name, " ".join("%s=%d" % kv for kv in sorted(c.items())), " total", sum(c.values()))
