import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select queue_id, start, end, name from kernels order by queue_id, start"))
def short(n):
    m = re.search(r"::(\w+)(<[^>]*>)?\(", n); return (m.group(1)) if m else n[:40]
prev = collections.Counter(); nxt = collections.Counter(); gaps_after = collections.Counter(); gsum = collections.Counter()
for i, (q, s, e, n) in enumerate(rows):
    if "copyBuffer" in n:
        if i > 0: prev[short(rows[i-1][3])] += 1
        if i + 1 < len(rows): nxt[short(rows[i+1][3])] += 1
for i in range(len(rows) - 1):
    if rows[i][0] == rows[i+1][0] == 1:
        g = rows[i+1][1] - rows[i][2]
        if 3000 < g < 50000:
            gaps_after[(short(rows[i][3]), short(rows[i+1][3]))] += 1; gsum[(short(rows[i][3]), short(rows[i+1][3]))] += g
print("before copyBuffer:", prev.most_common(6)); print("after copyBuffer:", nxt.most_common(6))
for k, v in sorted(gsum.items(), key=lambda kv: -kv[1])[:14]:
    print("gap pair %-60s n=%4d total %.2f ms avg %.1f us" % (k, gaps_after[k], v / 1e6, v / gaps_after[k] / 1e3))
