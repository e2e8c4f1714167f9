"""Turns the rocprofv3 (rocpd sqlite) outputs of scripts/profile_bench.sh into the JSON summary committed under
profiles/ (the file bench.py's pmc_traffic reads).   python scripts/summarize_prof.py gpurun_out/prof out.json ROWS"""
import glob
import json
import sqlite3
import sys

src, dst, rows = sys.argv[1], sys.argv[2], int(sys.argv[3])
out = {}


def db_of(sub):
    found = sorted(glob.glob("%s/%s/**/*.db" % (src, sub), recursive=True))
    if not found:
        raise FileNotFoundError("no rocpd database under %s/%s" % (src, sub))
    return sqlite3.connect(found[0])


c = db_of("trace")
out["kernel_trace_stats"] = {
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-secondary",
    "top_kernels": [dict(name=r[0][:140], calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4])
                    for r in c.execute("select * from top_kernels limit 8")]}
pm = {"rows": rows, "dispatches": {}}
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    try:
        c = db_of(sub)
        q = ("select counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
             "where kernel_name like '%k_sample_bf16%' group by counter_name")
        for name, total, n in c.execute(q):
            pm[name] = total / max(1, n)      # per launch
            pm["dispatches"][name] = n
    except Exception as e:  # noqa: BLE001
        pm[sub + "_error"] = str(e)
if "SQ_VALU_MFMA_BUSY_CYCLES" in pm and "SQ_BUSY_CYCLES" in pm:
    # MFMA_BUSY counts per SIMD-cycle x4 on gfx950 (guide: divide by 4 x CU-busy cycles)
    pm["note_mfma"] = "see DESIGN.md section 6 for how the busy fraction is derived from these"
out["pmc_k_sample_bf16"] = pm
out["pmc_command"] = ("separate rocprofv3 --pmc passes of python3 bench.py --states 131072 --steps 1 --warmup 0 "
                      "--no-cpu-baseline --no-secondary (one launch of %d rows x 100 steps); counters summed over the "
                      "k_sample_bf16 dispatches only, per launch; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them" % rows)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
