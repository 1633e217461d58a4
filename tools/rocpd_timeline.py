#!/usr/bin/env python3
"""Per-queue busy time / overlap / idle gaps from a rocprofv3 kernel-trace rocpd database.
usage: rocpd_timeline.py <db> [last_fraction] [exclude,names] [--dump MS [--dump-end MS_BEFORE_END]]
--dump MS: instead of the summary, list every kernel of a MS-millisecond window (start relative to the window, duration, queue, name) - the window ends
MS_BEFORE_END milliseconds before the end of the last voxel-update kernel (the last kernel of the run if there is none)."""
import sqlite3
import sys


def main():
    dump = dump_end = None
    if "--dump" in sys.argv:
        i = sys.argv.index("--dump"); dump = float(sys.argv[i + 1]); del sys.argv[i:i + 2]
        dump_end = 3.0
        if "--dump-end" in sys.argv:
            i = sys.argv.index("--dump-end"); dump_end = float(sys.argv[i + 1]); del sys.argv[i:i + 2]
    db = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.8
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = c.execute("select start, end, name, %s from kernels order by start" % (qcol or "0")).fetchall()
    excl = [e for e in (sys.argv[3].split(",") if len(sys.argv) > 3 else []) if e]      # e.g. "Cijk_,at::native": bench.py's untimed clock warm-up
    rows = [r for r in rows if not any(e in r[2] for e in excl)]
    t0, t1 = rows[0][0], rows[-1][1]
    if dump is not None:
        upd = [e for _, e, n, _ in rows if "k_update" in n]       # the frame loop ends with its last voxel update (teardown work may follow much later)
        hi = (upd[-1] if upd else t1) - dump_end * 1e6; lo = hi - dump * 1e6
        qs = {}
        for s, e, n, q in rows:
            if e >= lo and s <= hi:
                short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
                print("%9.1f us  %7.1f us  q%-2s %s" % ((s - lo) / 1e3, (e - s) / 1e3, qs.setdefault(q, len(qs)), short))
        return
    lo = t1 - (t1 - t0) * frac                      # analyse the steady-state tail
    rows = [r for r in rows if r[0] >= lo]
    span = (rows[-1][1] - rows[0][0]) / 1e6
    per_q = {}
    for s, e, n, q in rows:
        per_q.setdefault(q, []).append((s, e, n))
    print("span %.1f ms, %d kernels, %d queues" % (span, len(rows), len(per_q)))
    for q, ks in sorted(per_q.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _ in ks) / 1e6
        gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
        gpos = [g for g in gaps if g > 0]
        big = sorted(gpos)[-5:]
        print("queue %s: %d kernels, busy %.1f ms (%.0f%%), gaps total %.1f ms, median gap %.1f us, top gaps %s us" %
              (q, len(ks), busy, 100 * busy / span, sum(gpos) / 1e6, sorted(gpos)[len(gpos) // 2] / 1e3 if gpos else 0, [round(g / 1e3) for g in big]))
        agg = {}
        for i in range(len(ks) - 1):
            g = ks[i + 1][0] - ks[i][1]
            if g > 0:
                key = ks[i][2][:40] + " -> " + ks[i + 1][2][:40]
                a = agg.setdefault(key, [0, 0]); a[0] += g; a[1] += 1
        for key, (tot, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
            print("    gap %-90s total %.2f ms  n=%d  avg %.1f us" % (key, tot / 1e6, cnt, tot / cnt / 1e3))
    # union busy
    ev = sorted([(s, 1) for s, e, _, _ in rows] + [(e, -1) for s, e, _, _ in rows])
    depth = 0; last = ev[0][0]; any_busy = 0; both = 0
    for t, d in ev:
        if depth >= 1: any_busy += t - last
        if depth >= 2: both += t - last
        depth += d; last = t
    print("GPU busy (any queue) %.1f ms (%.0f%%), >=2 kernels concurrently %.1f ms" % (any_busy / 1e6, 100 * any_busy / 1e6 / span, both / 1e6))


if __name__ == "__main__":
    main()
