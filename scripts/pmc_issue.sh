#!/bin/bash
# SQ issue counters of the headline's dominant kernel (k_enc_select_runs<8, 2>): is it bound by instruction issue?
# Separate rocprofv3 --pmc passes with kernel tracing only (MI355X_MICROARCH.md: no sys/hip/hsa trace next to --pmc).
# Usage (GPU box, repo root): scripts/pmc_issue.sh r04 ; writes profiles/r04_c2_issue_counters.json
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_issue_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $OUT/avail.txt 2>&1 < /dev/null
grep -o "SQ_[A-Z_0-9]*" $OUT/avail.txt | sort -u > $OUT/sq_names.txt
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM"; do
    name=$(echo $set | tr ' ' '+')
    timeout 400 rocprofv3 --pmc $set -d $OUT/$name -o b -- python $R/bench.py --no-configs --no-cpu-baseline --steps 10 > /dev/null 2> $OUT/$name.err < /dev/null
done
python - <<PYEOF
import glob, json, sqlite3, collections, os
out = {}
for db in glob.glob("$OUT/*/b_results.db"):
    try:
        con = sqlite3.connect(db)
        acc = collections.defaultdict(list)
        for name, cname, value in con.execute("select kernel_name, counter_name, value from counters_collection order by dispatch_id"):
            if "k_enc_select_runs<8, 2>" in name or "k_expand_rle" in name:
                k = "k_enc_select_runs<8, 2>" if "select_runs" in name else "k_expand_rle"
                acc[(k, cname)].append(float(value))
        for (k, c), v in acc.items():
            v = v[len(v) // 3:]
            out.setdefault(k, {})[c] = sum(v) / len(v)
    except Exception as e:
        out.setdefault("errors", []).append("%s: %s" % (os.path.basename(os.path.dirname(db)), e))
import sys
sys.path.insert(0, "$R")
import bench
json.dump({"config": {"workload": "c2 (bench.py --no-configs --steps 10): per-launch averages of the SQ counters, one rocprofv3 --pmc pass per group of three"},
           "kernel_source_sha16": bench.kernel_source_sha(), "counters_per_launch": out}, open("$R/profiles/${TAG}_c2_issue_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PYEOF
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_c2_issue_counters.json $R/gpurun_out/profiles_$TAG/
wc -l $OUT/sq_names.txt
