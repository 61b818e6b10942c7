"""From a rocpd kernel-trace database of tools/trace_steps.py: take the last `steps` steps, and per queue print busy time, idle time
inside the step window, and for the main queue the idle gaps by (previous kernel -> next kernel).  python tools/step_timeline.py <db> <steps>"""
import collections, re, sqlite3, sys
db, steps = sqlite3.connect(sys.argv[1]), int(sys.argv[2])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel_dispatch" in t][0]
rows = list(cur.execute("select queue_id, start, end, name from %s order by start" % kt))
def short(n):
    m = re.search(r"(\w+)(<[^>]*>)?\(", n.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else n[:40]
# step boundary = the stem conv forward launch (first kernel touching the input): find the cast/stem kernel name that occurs once per step
names = collections.Counter(short(r[3]) for r in rows)
cnt = collections.Counter(r[0] for r in rows)
main = cnt.most_common(1)[0][0]
mark = sys.argv[3] if len(sys.argv) > 3 else "ce_loss"          # a kernel launched once per step / pass
ce = [r for r in rows if short(r[3]).startswith(mark)]
assert len(ce) >= steps + 1, (len(ce), sorted(names.items(), key=lambda kv: -kv[1])[:30])
t0, t1 = ce[-steps - 1][1], ce[-1][1]
win = [r for r in rows if t0 <= r[1] < t1]
print("window: %d steps, %.3f ms/step, %d kernels/step" % (steps, (t1 - t0) / steps / 1e6, len(win) / steps))
for q in sorted(set(r[0] for r in win)):
    ks = [r for r in win if r[0] == q]
    busy = sum(r[2] - r[1] for r in ks)
    print("queue %s: %d kernels/step busy %.2f ms/step" % (q, len(ks) / steps, busy / steps / 1e6))
# union busy (any queue) and idle
ev = sorted((r[1], r[2]) for r in win)
u, cs, ce_ = 0, ev[0][0], ev[0][1]
for s, e in ev[1:]:
    if s > ce_:
        u += ce_ - cs; cs, ce_ = s, e
    else:
        ce_ = max(ce_, e)
u += ce_ - cs
print("any-queue busy %.2f ms/step, GPU idle %.2f ms/step" % (u / steps / 1e6, (t1 - t0 - u) / steps / 1e6))
ks = [r for r in win if r[0] == main]
agg = collections.defaultdict(lambda: [0, 0.0])
for a, b in zip(ks, ks[1:]):
    g = b[1] - a[2]
    if g > 0:
        k = (short(a[3]), short(b[3])); agg[k][0] += 1; agg[k][1] += g
tot = sum(v[1] for v in agg.values())
print("main queue %s idle between consecutive kernels: %.2f ms/step" % (main, tot / steps / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-34s -> %-34s n/step %5.1f  %.3f ms/step  avg %.1f us" % (k[0][:34], k[1][:34], v[0] / steps, v[1] / steps / 1e6, v[1] / v[0] / 1e3))
kt_ = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    kt_[short(r[3])][0] += 1; kt_[short(r[3])][1] += r[2] - r[1]
print("kernels (ms/step):")
for k, v in sorted(kt_.items(), key=lambda kv: -kv[1][1])[:30]:
    print("  %-40s n/step %6.1f  %.3f ms/step" % (k[:40], v[0] / steps, v[1] / steps / 1e6))
