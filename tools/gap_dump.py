"""Dump the kernels of every queue around the main queue's idle gaps that follow a given kernel:  python tools/gap_dump.py <db> <prev-kernel-substring> [min_gap_us] [count]"""
import re, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel_dispatch" in t][0]
rows = list(cur.execute("select queue_id, start, end, name from %s order by start" % kt))
def short(n):
    m = re.search(r"(\w+)(<[^>]*>)?\(", n.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else n[:40]
main = collections.Counter(r[0] for r in rows).most_common(1)[0][0]
mq = [r for r in rows if r[0] == main]
key, mingap, count = sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 15.0, int(sys.argv[4]) if len(sys.argv) > 4 else 3
shown = 0
for a, b in zip(mq[len(mq) // 2:], mq[len(mq) // 2 + 1:]):
    gap = (b[1] - a[2]) / 1e3
    if key in short(a[3]) and gap >= mingap:
        t0 = a[1] - 60000
        t1 = b[2] + 20000
        print("---- gap %.1f us after %s" % (gap, short(a[3])))
        for r in rows:
            if r[2] >= t0 and r[1] <= t1:
                print("  q%-2d %9.1f .. %9.1f  (%7.1f us)  %s" % (r[0], (r[1] - a[2]) / 1e3, (r[2] - a[2]) / 1e3, (r[2] - r[1]) / 1e3, short(r[3])))
        shown += 1
        if shown >= count:
            break
