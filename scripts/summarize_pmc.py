"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide
prescribes) of `bench.py` into profiles/<tag>_pmc_traffic.json: HBM bytes per launch per kernel.
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE counts 128-byte requests as 64 bytes for
wide coalesced streaming reads -> doubled.  WRITE_SIZE is used as reported (it matches the known
byte count of k_expand's output 1:1)."""
import collections, csv, json, sys

src, out, tag_cfg = sys.argv[1], sys.argv[2], json.loads(sys.argv[3])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("%s/%s/b_counter_collection.csv" % (src, c))):
        n = r["Kernel_Name"]
        if "sb::" in n:
            acc[n.split("(")[0].replace("void ", "").replace("sb::", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v) // 3:]  # drop the untimed first pass / warm-up
        res.setdefault(k, {})[c] = sum(v) / len(v)
kern = {}
for k, v in sorted(res.items()):
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    kern[k] = {"fetch_size_kb_raw": round(f, 1), "write_size_kb_raw": round(w, 1),
               "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump({"config": tag_cfg, "units": "FETCH_SIZE/WRITE_SIZE in KiB per dispatch (rocprofv3 --pmc, separate passes)",
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads half for wide coalesced streams)",
           "kernels": kern}, open(out, "w"), indent=1)
print(json.dumps(kern, indent=1))
