#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) as per-kernel statistics:
calls, total / average / min / max duration, share of GPU time; optionally per-counter means.
Usage: tools/rocpd_summary.py <results.db> [--counters]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-72s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
    for name, calls, tot, avg, mn, mx in rows:
        short = name.split("(")[0][-72:]
        print("%-72s %8d %12.3f %10.2f %10.2f %10.2f %6.1f" % (short, calls, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    if "--counters" in sys.argv:
        try:
            ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            print("\ncounters_collection columns:", ccols)
            q = "select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name order by kernel_name"
            for row in cur.execute(q):
                print("%-60s %-28s n=%6d avg=%16.2f sum=%18.0f" % (row[0].split("(")[0][-60:], row[1], row[2], row[3], row[4]))
        except Exception as e:  # schema differs between rocprofv3 versions
            print("counter summary unavailable:", e)


if __name__ == "__main__":
    main()
