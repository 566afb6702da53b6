"""One multigrid setup of a bench trace split by HIP stream / queue: busy time, idle time inside the span, kernel count per
stream -- where the main chain waits.  python scripts/prof_streams.py DB"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
src = "kernels" if "kernels" in tabs else next(t for t in tabs if "kernel" in t.lower())
cols = [r[1] for r in c.execute(f"pragma table_info({src})")]
print("columns:", cols)
qcol = next((x for x in ("stream_id", "queue_id", "stream", "queue") if x in cols), None)
rows = list(c.execute(f"select name, grid_x, start, end, {qcol or '0'} from {src} order by start"))
k = [i for i, r in enumerate(rows) if "abs_range_kernel" in r[0]]
e = [i for i, r in enumerate(rows) if "pcg_init_plain_kernel" in r[0]]
a = k[-1]; b = [x for x in e if x > a][0]
seg = rows[a:b]
t0, t1 = seg[0][2], seg[-1][3]
per = collections.defaultdict(list)
for n, g, s, en, q in seg:
    per[q].append((s, en, n))
print(f"setup span {(t1 - t0) / 1e3:.1f} us, {len(seg)} kernels")
for q, v in per.items():
    busy = sum(en - s for s, en, _ in v)
    gaps = sorted(((v[i + 1][0] - v[i][1]) / 1e3, v[i][2].replace("padne::", "").split("(")[0][:40], v[i + 1][2].replace("padne::", "").split("(")[0][:40]) for i in range(len(v) - 1))
    print(f"stream {q}: {len(v)} kernels, busy {busy / 1e3:.1f} us, first {(v[0][0] - t0) / 1e3:.1f} us, last end {(v[-1][1] - t0) / 1e3:.1f} us, idle inside {((v[-1][1] - v[0][0]) - busy) / 1e3:.1f} us")
    print("   largest gaps:", [(round(gp, 1), x, y) for gp, x, y in gaps[-12:]])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for st_, en, nm in v:
        k2 = nm.replace("padne::", "").replace("void ", "").split("(")[0][:52]
        agg[k2][0] += 1; agg[k2][1] += (en - st_) / 1e3
    print("   kernels by time:")
    for k2, (cnt, tt) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"      {k2:54s} n={cnt:4d} {tt:9.1f} us")
    hist = collections.Counter(int(gp // 5) * 5 for gp, _, _ in gaps if gp > 0)
    print("   gap histogram (us bucket: count):", sorted(hist.items())[:14])
