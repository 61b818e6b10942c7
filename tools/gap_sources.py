"""Which launches are followed by idle time on the busiest queue?  python tools/gap_sources.py <rocpd db> [steps]
Aggregates the gaps (< 200 us, i.e. inside a step) between consecutive kernels of the main queue by (previous kernel -> next kernel)."""
import collections
import re
import sqlite3
import sys

db, steps = sqlite3.connect(sys.argv[1]), float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
rows = list(cur.execute("select queue_id, start, end, name from kernels order by start"))
cnt = collections.Counter(r[0] for r in rows)
main = cnt.most_common(1)[0][0]
ks = [r for r in rows if r[0] == main]


def short(n):
    m = re.search(r"(\w+)(<[^>]*>)?\(", n.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else n[:40]


agg = collections.defaultdict(lambda: [0, 0.0])
for a, b in zip(ks, ks[1:]):
    g = b[1] - a[2]
    if 2000 < g < 200000:
        k = (short(a[3]), short(b[3]))
        agg[k][0] += 1
        agg[k][1] += g
tot = sum(v[1] for v in agg.values())
print("main queue %s: gaps of 2-200 us: %.2f ms/step" % (main, tot / 1e6 / steps))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
    print("  %-34s -> %-34s n/step %5.1f  %.3f ms/step  avg %5.1f us" % (k[0], k[1], v[0] / steps, v[1] / 1e6 / steps, v[1] / v[0] / 1e3))
