import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1])
per = collections.defaultdict(lambda: [0, 0.0])
for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (sys.argv[2],)):
    k = re.sub(r"\(.*", "", name)[:70]
    per[k][0] += 1; per[k][1] += val
for k, (n, v) in sorted(per.items(), key=lambda kv: -kv[1][1])[:5]:
    print(f"{sys.argv[2]} {k}: launches {n}, per launch {v / n:.1f}")
