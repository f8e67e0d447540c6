#!/usr/bin/env python3
"""The last <ms> milliseconds of a rocprofv3 kernel trace (rocpd database) as a timeline: start (us from the first listed kernel),
duration, gap to the previous kernel's end, name. Usage: tools/rocpd_timeline.py <results.db> [ms = 8] [anchor [after_ms = 1.5]]
With an anchor the window ends after_ms behind the last kernel whose name contains it (instead of at the end of the trace)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1]); ms = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    if not rows:
        print("no kernels"); return
    # memory copies of the same trace (rocprofv3 --memory-copy-trace), where the database has them: listed between the kernels as "copy <direction> <bytes>"
    for (table,) in cur.execute("select name from sqlite_master where type in ('table', 'view') and name like '%memory_cop%'").fetchall():
        ccols = [r[1] for r in cur.execute("pragma table_info(%s)" % table)]
        if "start" in ccols and "end" in ccols:
            label = [c for c in ("name", "direction", "kind") if c in ccols]
            size = "size" if "size" in ccols else ("bytes" if "bytes" in ccols else None)
            query = "select %s, start, end%s from %s" % (label[0] if label else "'copy'", (", " + size) if size else "", table)
            for r in cur.execute(query).fetchall():
                rows.append(("copy %s %s" % (r[0], r[3] if size else ""), r[1], r[2]))
            break
    rows.sort(key=lambda r: r[1])
    t_last = max(r[2] for r in rows)
    if len(sys.argv) > 3:
        anchored = [r[2] for r in rows if sys.argv[3] in r[0]]
        if anchored:
            t_last = max(anchored) + (float(sys.argv[4]) if len(sys.argv) > 4 else 1.5) * 1e6
    rows = [r for r in rows if r[1] >= t_last - ms * 1e6 and r[1] <= t_last]
    t0 = rows[0][1]; prev_end = t0
    for name, start, end in rows:
        print("%10.1f us  %9.1f us  gap %7.1f  %s" % ((start - t0) / 1e3, (end - start) / 1e3, (start - prev_end) / 1e3, name.split("(")[0][-60:]))
        prev_end = max(prev_end, end)


if __name__ == "__main__":
    main()
