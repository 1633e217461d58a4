#!/usr/bin/env python3
"""Per-kernel busy time in the first and in the last `frac` of a rocprofv3 kernel-trace (rocpd sqlite): what grows over a long stream.
usage: rocpd_stats_window.py <db> [frac=0.2] [exclude-substrings,comma-separated]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
    excl = [e for e in (sys.argv[3].split(",") if len(sys.argv) > 3 else []) if e]
    c = sqlite3.connect(db)
    rows = [r for r in c.execute("select start, end, name from kernels order by start").fetchall() if not any(e in r[2] for e in excl)]
    t0, t1 = rows[0][0], rows[-1][1]
    span = t1 - t0
    win = {"first": (t0, t0 + frac * span), "last": (t1 - frac * span, t1)}
    agg = {}
    for s, e, n in rows:
        n = n.replace("(anonymous namespace)::", "").split("(")[0][:60]
        for w, (a, b) in win.items():
            if a <= s < b:
                d = agg.setdefault(n, {"first": [0, 0.0], "last": [0, 0.0]})
                d[w][0] += 1; d[w][1] += (e - s) / 1e3
    print("window = %.0f %% of the run = %.2f s each" % (100 * frac, frac * span / 1e9))
    print("| kernel | first: calls | first: total ms | first: avg us | last: calls | last: total ms | last: avg us | growth of total |")
    print("|---|---|---|---|---|---|---|---|")
    tot = {"first": 0.0, "last": 0.0}
    for n, d in sorted(agg.items(), key=lambda kv: -kv[1]["last"][1])[:28]:
        f, l = d["first"], d["last"]
        print("| `%s` | %d | %.1f | %.1f | %d | %.1f | %.1f | %.2fx |" % (n, f[0], f[1] / 1e3, f[1] / max(f[0], 1), l[0], l[1] / 1e3, l[1] / max(l[0], 1), l[1] / f[1] if f[1] else float("inf")))
    for n, d in agg.items():
        tot["first"] += d["first"][1]; tot["last"] += d["last"][1]
    print("| all kernels | | %.1f | | | %.1f | | %.2fx |" % (tot["first"] / 1e3, tot["last"] / 1e3, tot["last"] / max(tot["first"], 1e-9)))


if __name__ == "__main__":
    main()
