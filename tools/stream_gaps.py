"""Idle gaps between consecutive kernels per queue from a rocprofv3 rocpd database: python tools/stream_gaps.py <db> <steps>"""
import sqlite3
import sys

db, steps = sqlite3.connect(sys.argv[1]), float(sys.argv[2])
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else "stream_id"
rows = list(cur.execute("select %s, start, end, name from kernels order by %s, start" % (qcol, qcol)))
by = {}
for q, s, e, n in rows:
    by.setdefault(q, []).append((s, e, n))
for q, ks in by.items():
    busy = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g < 50000]          # < 50 us: launch-to-launch gaps inside a step
    print("queue %s: %d kernels, busy %.2f ms/step, small gaps: n=%d sum %.2f ms/step, median %.1f us, p90 %.1f us" % (
        q, len(ks), busy / steps / 1e6, len(small), sum(small) / steps / 1e6,
        sorted(small)[len(small) // 2] / 1e3 if small else 0, sorted(small)[int(len(small) * 0.9)] / 1e3 if small else 0))
