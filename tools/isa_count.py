#!/usr/bin/env python3
"""Instruction mix of one kernel from hipcc's -S output: isa_count.py file.s substring [substring...]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
for pat in sys.argv[2:]:
    start = next((i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0] and l.rstrip().endswith(tuple([":"] + ["EEEv"])) or (l.startswith("_Z") and pat in l and ": " in l and "@" in l)), None)
    if start is None:
        print(pat, "not found")
        continue
    c = collections.Counter()
    for l in lines[start + 1:]:
        l = l.strip()
        if l.startswith("s_endpgm"):
            break
        m = re.match(r"([a-z_0-9]+)\b", l)
        if not m or l.startswith((".", ";")):
            continue
        op = m.group(1)
        if op.startswith("v_pk"): c["v_pk"] += 1
        elif op.startswith("v_permlane"): c["v_permlane"] += 1
        elif op.startswith(("v_mov", "v_accvgpr")): c["v_mov"] += 1
        elif op.startswith("v_mfma"): c["v_mfma"] += 1
        elif op.startswith("v_"): c["v_other"] += 1
        elif op.startswith(("ds_read", "ds_load")): c["ds_read"] += 1
        elif op.startswith(("ds_write", "ds_store")): c["ds_write"] += 1
        elif op.startswith("ds_"): c["ds_other"] += 1
        elif op.startswith("global_load") or op.startswith("buffer_load"): c["gload"] += 1
        elif op.startswith("global_store") or op.startswith("buffer_store"): c["gstore"] += 1
        elif op.startswith("s_waitcnt"): c["s_waitcnt"] += 1
        elif op.startswith("scratch"): c["scratch"] += 1
        elif op.startswith("s_"): c["s_other"] += 1
        else: c[op] += 1
    print(pat, dict(sorted(c.items())), "total", sum(c.values()))
