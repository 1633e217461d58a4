#!/usr/bin/env python3
"""Per-kernel average of a PMC counter from a rocprofv3 rocpd sqlite database.  usage: rocpd_pmc.py <db> [name-filter]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    pmc = [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]
    print("tables:", pmc)
    for t in ("counters_collection", "pmc_events"):
        if t in tabs:
            cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
            print(t, cols)
    if "counters_collection" in tabs:
        q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
             "where kernel_name like ? group by kernel_name, counter_name order by 5 desc")
        for r in c.execute(q, ("%" + flt + "%",)).fetchall()[:400]:
            print("| `%s` | %s | %d | %.1f | %.1f |" % (r[0][:80], r[1], r[2], r[3], r[4]))


if __name__ == "__main__":
    main()
