"""rocpd databases of scripts/profile_c5.sh -> the JSON committed as profiles/r0N_c5_kernels.json.
    python scripts/summarize_c5.py gpurun_out/c5 out.json"""
import glob
import json
import sqlite3
import sys

src, dst = sys.argv[1], sys.argv[2]


def db_of(sub):
    return sqlite3.connect(sorted(glob.glob("%s/%s/**/*.db" % (src, sub), recursive=True))[0])


out = {"command": "rocprofv3 --kernel-trace --stats -- python3 scripts/quick_c5.py 262144 bf16 "
                  "(C5 net, 262144 rows, two sample() calls of 100 Euler steps); PMC: separate passes, 5 steps"}
c = db_of("trace")
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
out["kernels"] = [dict(name=r[0][:100], calls=r[1], total_us=round(r[2], 1), avg_us=round(r[3], 1), min_us=round(r[4], 1),
                       max_us=round(r[5], 1), pct=round(100 * r[2] / tot, 2)) for r in rows[:10]]
seq = c.execute("select name, (end-start)/1e3 from kernels order by start").fetchall()
first = [i for i, r in enumerate(seq) if "_inputs" in r[0] or "k_w_temb" in r[0]][150]
nxt = [i for i, r in enumerate(seq) if "_inputs" in r[0] or "k_w_temb" in r[0]][151]
out["one_solver_step_us"] = [[r[0][:60], round(r[1], 1)] for r in seq[first:nxt]]
t0, t1 = c.execute("select start from kernels order by start").fetchall()[first][0], c.execute("select start from kernels order by start").fetchall()[nxt][0]
out["one_solver_step_span_us"] = round((t1 - t0) / 1e3, 1)
pm = {}
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    q = ("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
         "where kernel_name like '%gemm%' group by 1, 2")
    for k, name, total, n in db_of(sub).execute(q):
        pm.setdefault(k[:80], {})[name] = total / max(1, n)   # per launch
out["pmc_per_launch"] = pm
out["pmc_note"] = "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (FETCH_SIZE x2 on gfx950, MI355X_MICROARCH.md)"
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
