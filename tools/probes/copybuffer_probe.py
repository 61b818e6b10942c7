import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
c = collections.Counter()
for q, gx, wx in cur.execute("select queue_id, grid_x, workgroup_x from kernels where name like '%copyBuffer%'"):
    c[(q, gx, wx)] += 1
print(c.most_common(12))
rows = list(cur.execute("select queue_id, start, end, name, grid_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "copyBuffer" in r[3]]
i0 = idx[len(idx) // 2]
while "copyBuffer" in rows[i0 - 1][3]: i0 -= 1
for r in rows[i0 - 6:i0 + 4]: print(r[0], r[3][:70], r[4], (r[2] - r[1]) / 1e3)
