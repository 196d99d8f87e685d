"""Per-kernel statistics (calls, total, average, min, max) from a rocprofv3 rocpd SQLite database."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                     "group by %s order by 3 desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-60s %6s %14s %14s %14s %14s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for n, cnt, tot, avg, mn, mx in rows:
        print("%-60s %6d %14d %14.0f %14d %14d %6.2f%%" % (n[:60], cnt, tot, avg, mn, mx, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
