#!/usr/bin/env python3
"""Host-side view of a `rocprofv3 --hip-trace --kernel-trace` rocpd database: where do the frame loop's host threads spend their time?
usage: rocpd_hip_trace.py <db> [--window MS --end MS_BEFORE_END]
Without --window: per thread, the HIP API calls by total time (count, total, mean, max).  With --window: every API call (>= 3 us, or any launch) and every kernel of a
window of MS milliseconds that ends MS_BEFORE_END before the last voxel-update kernel, merged by start time (API rows carry the thread, kernel rows the queue)."""
import sqlite3
import sys


def find_regions(c):
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    for cand in ("regions", "regions_and_samples", "api_calls", "hip_api"):
        if cand in names:
            cols = [r[1] for r in c.execute("pragma table_info(%s)" % cand)]
            if "start" in cols and "end" in cols and "name" in cols:
                return cand, cols, names
    return None, None, names


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tab, cols, names = find_regions(c)
    if tab is None:
        print("no region table found; tables / views:", names)
        return
    tid = "tid" if "tid" in cols else ("thread_id" if "thread_id" in cols else "0")
    if "--window" not in sys.argv:
        print("table %s" % tab)
        rows = c.execute("select %s, name, count(*), sum(end - start), max(end - start) from %s group by 1, 2 order by 4 desc" % (tid, tab)).fetchall()
        per = {}
        for t, n, k, tot, mx in rows:
            per.setdefault(t, []).append((n, k, tot, mx))
        for t, lst in sorted(per.items(), key=lambda kv: -sum(x[2] for x in kv[1]))[:6]:
            print("thread %s: %.1f ms inside HIP calls" % (t, sum(x[2] for x in lst) / 1e6))
            for n, k, tot, mx in lst[:14]:
                print("    %-44s n=%-7d total %8.2f ms  mean %7.1f us  max %8.1f us" % (n[:44], k, tot / 1e6, tot / k / 1e3, mx / 1e3))
        return
    w = float(sys.argv[sys.argv.index("--window") + 1]); e_ms = float(sys.argv[sys.argv.index("--end") + 1]) if "--end" in sys.argv else 6.0
    kcols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in kcols else ("stream_id" if "stream_id" in kcols else "0")
    ks = c.execute("select start, end, name, %s from kernels order by start" % qcol).fetchall()
    upd = [e for _, e, n, _ in ks if "k_update" in n]
    hi = (upd[-1] if upd else ks[-1][1]) - e_ms * 1e6; lo = hi - w * 1e6
    ev = []
    qs, ts = {}, {}
    for s, e, n, q in ks:
        if e >= lo and s <= hi:
            ev.append((s, "K q%-2d %7.1f us  %s" % (qs.setdefault(q, len(qs)), (e - s) / 1e3, n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40])))
    for s, e, n, t in c.execute("select start, end, name, %s from %s where end >= ? and start <= ?" % (tid, tab), (lo, hi)):
        if e - s >= 3000 or "Launch" in n or "Synchronize" in n:
            ev.append((s, "A t%-2d %7.1f us  %s" % (ts.setdefault(t, len(ts)), (e - s) / 1e3, n[:40])))
    for s, txt in sorted(ev):
        print("%9.1f  %s" % ((s - lo) / 1e3, txt))


if __name__ == "__main__":
    main()
