"""Turns the rocprofv3 (rocpd sqlite) outputs of scripts/profile_bench.sh into the JSON summary
committed under profiles/.   python scripts/summarize_prof.py gpurun_out/prof profiles/<name>.json"""
import json
import sqlite3
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {}
c = sqlite3.connect(src + "/trace/bench_results.db")
out["kernel_trace_stats: rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline"] = [
    dict(name=r[0][:140], calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4]) for r in c.execute("select * from top_kernels limit 8")]
pm = {}
for db in ("pmc_sq", "pmc_fetch", "pmc_write"):
    try:
        c = sqlite3.connect("%s/%s/pmc_results.db" % (src, db))
        for r in c.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
            if "k_sample" in r[0]:
                pm[r[1]] = r[2]
    except Exception as e:  # noqa: BLE001
        pm[db + "_error"] = str(e)
out["pmc: separate passes of python3 bench.py --states 131072 --steps 1 --warmup 0 (one launch: 4194304 rows x 100 steps)"] = pm
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(pm, indent=1))
