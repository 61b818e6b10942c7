"""Per-dispatch durations of the kernels whose name contains <substr>, for the last step of a tools/trace_steps.py trace:
python tools/kernel_instances.py <db> <substr> [mark]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
sub, mark = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "ce_loss")
gx = [c for c in cols if c in ("grid_x", "grid_size_x", "grid_size")]
q = "select start, end, name%s from kernels order by start" % ("".join(", " + c for c in gx))
rows = list(cur.execute(q))
marks = [r[0] for r in rows if mark in r[2]]
t0, t1 = marks[-2], marks[-1]
prev = None
for r in rows:
    if t0 <= r[0] < t1 and sub in r[2]:
        print("%8.1f us  grid %s  gap-before %6.1f us   prev %s" % ((r[1] - r[0]) / 1e3, r[3:] if gx else "", (r[0] - prev[1]) / 1e3 if prev else 0, re.sub(r"\(.*", "", prev[2])[-40:] if prev else ""))
    if t0 <= r[0] < t1:
        prev = r
print("columns:", cols)
