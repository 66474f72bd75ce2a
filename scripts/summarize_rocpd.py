"""rocprofv3 (rocpd sqlite output) -> the summaries committed under profiles/:
   <tag>_<cfg>_kernel_stats.csv   per-kernel calls / total / average duration of the sb:: kernels (--kernel-trace pass)
   <tag>_<cfg>_pmc_traffic.json   HBM bytes per launch per kernel from the FETCH_SIZE and WRITE_SIZE passes (separate
                                  runs, MI355X_MICROARCH.md §HBM): hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 —
                                  on gfx950 FETCH_SIZE reports half the bytes of wide coalesced streaming reads; WRITE_SIZE is
                                  used as reported (it matches k_expand's known output byte count 1:1)
Usage: python scripts/summarize_rocpd.py gpurun_out/prof_r02 r02 c1 '{"workload": "C1", ...}' """
import collections
import csv
import json
import os
import sqlite3
import sys

src, tag, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
cfg_desc = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {"workload": cfg}
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def short(name):
    n = name.replace("void ", "")
    n = n[:n.index("(")] if "(" in n else n
    return n.replace("sb::", "")


db = sqlite3.connect(os.path.join(src, cfg + "_kt", "b_results.db"))
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                  "group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
with open(os.path.join(out_dir, "%s_%s_kernel_stats.csv" % (tag, cfg)), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        if "sb::" in r[0] or r[2] * 50 > tot:
            w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / tot), r[4], r[5]])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(src, cfg + "_pmc", c, "b_results.db")
    if not os.path.exists(p):
        continue
    acc = collections.defaultdict(list)
    for name, value in sqlite3.connect(p).execute("select kernel_name, value from counters_collection where counter_name = ? "
                                                  "order by dispatch_id", (c,)):
        if "sb::" in name:
            acc[short(name)].append(float(value))
    for k, v in acc.items():
        v = v[len(v) // 3:]  # drop the untimed first pass / warm-up launches
        res.setdefault(k, {})[c] = sum(v) / len(v)
kern = {}
for k, v in sorted(res.items()):
    fsz, wsz = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    kern[k] = {"fetch_size_kb_raw": round(fsz, 1), "write_size_kb_raw": round(wsz, 1), "hbm_bytes_per_launch": int((2 * fsz + wsz) * 1024)}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as _bench  # noqa: E402  (kernel_source_sha: bench.py reports this file's traffic only for the sources it was collected for)
if not kern:   # (no counter pass for this configuration: no file rather than an empty one)
    print(cfg, "no PMC passes found: no pmc_traffic file written")
    sys.exit(0)
json.dump({"config": cfg_desc, "kernel_source_sha16": _bench.kernel_source_sha(),
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (rocprofv3 --pmc, separate passes)",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads half for wide coalesced streams)",
           "kernels": kern}, open(os.path.join(out_dir, "%s_%s_pmc_traffic.json" % (tag, cfg)), "w"), indent=1)
top = sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:6]
print(cfg, [(k, v["hbm_bytes_per_launch"]) for k, v in top])
