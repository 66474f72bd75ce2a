"""Per-kernel SQ counters from one rocprofv3 --pmc pass (csv): mean per dispatch, plus shares of
SQ_WAVE_CYCLES (WAIT_ANY = parked on s_waitcnt/barrier, WAIT_INST_ANY = issue stalls,
ACTIVE_INST_* = issuing)."""
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "sb::" not in n:
        continue
    k = n.split("(")[0].replace("void ", "").replace("sb::", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    m = {c: sum(v[len(v) // 3:]) / max(1, len(v[len(v) // 3:])) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-36s" % k, " ".join("%s=%.3g(%.0f%%)" % (c.replace("SQ_", ""), v, 100 * v / wc) for c, v in sorted(m.items())))
