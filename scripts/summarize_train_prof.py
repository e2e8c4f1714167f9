"""Summary of a rocprofv3 kernel trace of scripts/bench_train.py: top kernels, launches per step, busy / idle time."""
import glob
import json
import sqlite3
import sys

out = sys.argv[1]
db = sorted(glob.glob(out + "/trace/**/*.db", recursive=True))[0]
c = sqlite3.connect(db)
top = [dict(name=r[0][:110], calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4]) for r in c.execute("select * from top_kernels limit 16")]
res = {"top_kernels": top}
try:
    rows = list(c.execute("select name, start, end from kernels order by start"))
except Exception as e:  # noqa: BLE001
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    res["schema_error"] = repr(e)
    res["objects"] = names
    rows = []
if rows:
    # steps are delimited by the Fourier-gradient kernel (once per step); take the last 6 whole steps
    marks = [i for i, r in enumerate(rows) if "k_bt_loss" in r[0]]   # the loss kernel runs once per step (one chunk)
    steps = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a:b]
        busy = sum(r[2] - r[1] for r in seg)
        span = seg[-1][2] - seg[0][1] + (rows[b][1] - seg[-1][2])
        gaps = sorted(((seg[i + 1][1] - seg[i][2], seg[i][0][:60], seg[i + 1][0][:60]) for i in range(len(seg) - 1)), reverse=True)
        steps.append(dict(launches=len(seg), busy_us=busy / 1e3, span_us=span / 1e3, idle_frac=1 - busy / span,
                          largest_gaps_us=[(g[0] / 1e3, g[1], g[2]) for g in gaps[:6]]))
    res["steps"] = steps[-4:]
    a, b = marks[-2], marks[-1]
    try:
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        res["kernels_columns"] = cols
        gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
        q = "select name, start, end%s from kernels order by start" % ((", " + gx) if gx else "")
        allrows = list(c.execute(q))
        res["one_step"] = [dict(k=r[0][:48], us=(r[2] - r[1]) / 1e3, grid=(r[3] if gx else None)) for r in allrows[a:b]]
    except Exception as e:  # noqa: BLE001
        res["one_step_error"] = repr(e)
json.dump(res, open(out + "/train_summary.json", "w"), indent=1)
for r in top:
    print("%6d %10.0f %9.1f %5.1f%% %s" % (r["calls"], r["total_us"], r["avg_us"], r["pct"], r["name"]))
for s in res.get("steps", []):
    print(json.dumps(s)[:900])
if "schema_error" in res:
    print(res["schema_error"], res["objects"])
