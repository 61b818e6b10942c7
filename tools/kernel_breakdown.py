"""Per-kernel ms/step from a rocprofv3 rocpd database: python tools/kernel_breakdown.py <db> <steps> [top]"""
import re
import sqlite3
import sys


def short(n):
    m = re.search(r"::(\w+)(<[^>]*>)?\(", n)
    return (m.group(1) + (m.group(2) or "")) if m else n[:46]


db, steps = sqlite3.connect(sys.argv[1]), float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows = list(db.cursor().execute("select name, count(*), sum(end-start) from kernels group by name order by 3 desc"))
for n, c, t in rows[:top]:
    print("%-46s calls/step %6.1f  ms/step %6.2f  avg us %7.1f" % (short(n)[:46], c / steps, t / steps / 1e6, t / c / 1e3))
print("total ms/step %.2f" % (sum(r[2] for r in rows) / steps / 1e6))
