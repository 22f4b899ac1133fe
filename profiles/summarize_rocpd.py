#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into small text tables that can be committed under profiles/.

usage: summarize_rocpd.py <dir-with-subdirs-of-*_results.db> [kernel-substring]
  kernel trace dbs  -> per-kernel calls / avg / min / max duration (us), like `--stats`
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
    root = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
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
            print("%-72s %6s %12s %12s %12s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct"))
            tot = cur.execute("select sum(duration) from kernels").fetchone()[0] or 1
            for k, n, a, lo, hi, s in cur.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration) "
                                                  "from kernels group by name order by sum(duration) desc limit 30"):
                print("%-72s %6d %12.2f %12.2f %12.2f %6.1f%%" % (short(k), n, a / 1e3, lo / 1e3, hi / 1e3, 100.0 * s / tot))
        print()


if __name__ == "__main__":
    main()
