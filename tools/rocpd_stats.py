#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (`*_results.db`) as a per-kernel stats table
(the same content as `rocprofv3 --stats` CSV output).  usage: rocpd_stats.py <db> [out.md]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                     "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    lines = ["| kernel | calls | total_us | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.2f |" % (r[0][:90], r[1], r[2], r[3], r[4], r[5],
                                                                     100 * r[2] / tot))
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
