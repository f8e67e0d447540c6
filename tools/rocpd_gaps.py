#!/usr/bin/env python3
"""GPU occupancy over time from a rocprofv3 kernel trace (rocpd database): how much of the wall time between the first and
the last kernel of the busiest stretch had at least one kernel resident, the idle time between dependent launches, and per
kernel name the time it was the ONLY kernel resident. Usage: tools/rocpd_gaps.py <results.db> [last_fraction]
(last_fraction: analyse only the last part of the trace, default 0.35 -- the timed region of bench.py comes last; a value
above 1 is taken as milliseconds before the end of the last kernel)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    if not rows:
        print("no kernels"); return
    t_first, t_last = rows[0][1], max(r[2] for r in rows)
    cut = t_last - (t_last - t_first) * frac if frac <= 1.0 else t_last - frac * 1e6
    rows = [r for r in rows if r[1] >= cut]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    busy, gaps, covered_until = 0, [], t0
    for name, start, end in rows:
        if start > covered_until:
            gaps.append(start - covered_until); covered_until = start
        if end > covered_until:
            busy += end - covered_until; covered_until = end
    wall = t1 - t0
    print("analysed: last %s of the trace, %d kernels, %.3f ms wall" % ("%.0f %%" % (frac * 100) if frac <= 1.0 else "%.1f ms" % frac, len(rows), wall / 1e6))
    print("GPU had a kernel resident %.1f %% of that time; %d idle gaps, %.3f ms in total, median %.1f us, max %.1f us"
          % (100.0 * busy / wall, len(gaps), sum(gaps) / 1e6, (sorted(gaps)[len(gaps) // 2] / 1e3) if gaps else 0.0, (max(gaps) / 1e3) if gaps else 0.0))
    per = {}
    for name, start, end in rows:
        short = name.split("(")[0]
        e = per.setdefault(short, [0, 0]); e[0] += 1; e[1] += end - start
    print("%-60s %8s %12s %8s" % ("kernel", "calls", "total_ms", "% wall"))
    for name, (calls, total) in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
        print("%-60s %8d %12.3f %8.1f" % (name[-60:], calls, total / 1e6, 100.0 * total / wall))


if __name__ == "__main__":
    main()
