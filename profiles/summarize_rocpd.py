#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into small text tables that can be committed under profiles/.

usage: summarize_rocpd.py <dir-with-subdirs-of-*_results.db> [kernel-substring] [--last N]
  kernel trace dbs  -> per-kernel calls / avg / min / max duration (us), like `--stats`, plus the MEDIAN and - with --last N - the
                       average and minimum over each kernel's LAST N dispatches: bench.py runs an untimed clock-ramp period and W warm-up
                       steps before its K timed steps, so "last K" is the timed region and is what ms_per_step has to be held against
                       (the all-dispatch average mixes the cold-clock calls in and comes out above the driver-timed step)
  pmc dbs           -> per-kernel per-counter average value per dispatch
"""
import glob
import os
import sqlite3
import sys


def short(name, n=70):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def main():
    argv = list(sys.argv[1:])
    last = 0
    if "--last" in argv:
        i = argv.index("--last")
        last = int(argv[i + 1])
        del argv[i:i + 2]
    root = argv[0]
    filt = argv[1] if len(argv) > 1 else ""
    for db_path in sorted(glob.glob(os.path.join(root, "*", "*_results.db"))):
        db = sqlite3.connect(db_path)
        cur = db.cursor()
        print("== %s" % os.path.relpath(db_path, root))
        rows = list(cur.execute("select counter_name, kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
                                "group by counter_name, kernel_name order by counter_name, avg(value) desc"))
        if rows:
            print("%-28s %-72s %6s %16s %16s %16s" % ("counter", "kernel", "disp", "avg/dispatch", "min", "max"))
            for c, k, n, a, lo, hi in rows:
                if filt in k:
                    print("%-28s %-72s %6d %16.1f %16.1f %16.1f" % (c, short(k), n, a, lo, hi))
        else:
            cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
            order = "start" if "start" in cols else "rowid"
            per = {}
            for k, d in cur.execute("select name, duration from kernels order by %s" % order):
                per.setdefault(k, []).append(d)
            tot = sum(sum(v) for v in per.values()) or 1
            hdr = "%-72s %6s %12s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "median_us", "min_us", "max_us", "pct")
            if last:
                hdr += " %14s %14s" % ("avg_last%d_us" % last, "min_last%d_us" % last)
            print(hdr)
            for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:60]:
                sv = sorted(v)
                med = sv[len(sv) // 2] if len(sv) % 2 else 0.5 * (sv[len(sv) // 2 - 1] + sv[len(sv) // 2])
                line = "%-72s %6d %12.2f %12.2f %12.2f %12.2f %6.1f%%" % (short(k), len(v), sum(v) / len(v) / 1e3, med / 1e3, sv[0] / 1e3, sv[-1] / 1e3, 100.0 * sum(v) / tot)
                if last:
                    tail = v[-last:]
                    line += " %14.2f %14.2f" % (sum(tail) / len(tail) / 1e3, min(tail) / 1e3)
                print(line)
        print()


if __name__ == "__main__":
    main()
