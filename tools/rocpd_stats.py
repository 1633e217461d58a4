#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (`*_results.db`) as a per-kernel stats table
(the same content as `rocprofv3 --stats` CSV output).  usage: rocpd_stats.py <db> [out.md] [--last FRACTION] [--exclude SUBSTR[,SUBSTR]]
`--last 0.15` restricts the table to dispatches that start in the last 15 % of the traced time span; `--exclude` drops kernels whose
name contains one of the substrings (bench.py's untimed clock warm-up: rocBLAS `Cijk_` and the torch element-wise kernel)."""
import sqlite3
import sys


def main():
    where = ""
    if "--last" in sys.argv:
        i = sys.argv.index("--last")
        frac = float(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    else:
        frac = None
    excl = []
    if "--exclude" in sys.argv:
        i = sys.argv.index("--exclude")
        excl = [e for e in sys.argv[i + 1].split(",") if e]
        del sys.argv[i:i + 2]
    db = sys.argv[1]
    c = sqlite3.connect(db)
    if frac is not None:
        t0, t1 = c.execute("select min(start), max(end) from kernels").fetchone()
        where = " where start >= %d" % int(t1 - frac * (t1 - t0))
    rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                     "max(end-start)/1e3 from kernels" + where + " group by name order by 3 desc").fetchall()
    rows = [r for r in rows if not any(e in r[0] for e in excl)]
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
