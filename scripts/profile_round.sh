#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: kernel stats and (separate runs, as MI355X_MICROARCH.md §HBM prescribes) the
# FETCH_SIZE / WRITE_SIZE counters, for the headline bench line and for the other configurations.
# Usage (on the GPU box, from the repo root): scripts/profile_round.sh r03 ; then, per configuration,
#   python scripts/summarize_rocpd.py gpurun_out/prof_r03 r03 <cfg> '<config json>'
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, pmc (0/1), bench args...
    local name=$1; shift
    local pmc=$1; shift
    timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/${name}_kt -o b -- python $R/bench.py "$@" > $OUT/${name}_stdout.txt 2> $OUT/${name}_kt.err < /dev/null
    if [ "$pmc" = "1" ]; then
        for c in FETCH_SIZE WRITE_SIZE; do
            timeout 400 rocprofv3 --pmc $c -d $OUT/${name}_pmc/$c -o b -- python $R/bench.py "$@" > /dev/null 2> $OUT/${name}_pmc_$c.err < /dev/null
        done
    fi
}
run c2 1 --no-configs --no-cpu-baseline --steps 10
run c1 1 --only c1 --no-cpu-baseline
run c3 1 --only c3,c3_lz4 --no-cpu-baseline
run c4 1 --only c4 --no-cpu-baseline
run c5 1 --only c5 --no-cpu-baseline
run one_page 1 --only one_page --no-cpu-baseline
find $OUT -name "*.db" -o -name "*.csv" | head -80 > $OUT/files.txt
for cfg in c2 c1 c3 c4 c5 one_page; do
    timeout 120 python $R/scripts/summarize_rocpd.py $OUT $TAG $cfg "{\"workload\": \"$cfg (bench.py, see profiles/${TAG}_${cfg}_stdout.txt)\"}" > $OUT/${cfg}_summary.txt 2>&1
    cp $OUT/${cfg}_stdout.txt $R/profiles/${TAG}_${cfg}_stdout.txt 2>/dev/null
done
cp $R/profiles/${TAG}_c2_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json 2>/dev/null   # (bench.py's roofline.traffic: the headline's counters)
python - <<PYEOF
import json
p = "$R/profiles/${TAG}_pmc_traffic.json"
try:
    d = json.load(open(p))
    d["config"].update(workload="c2 (bench.py default run)", columns_per_gpu=512, codec="adaptive")
    json.dump(d, open(p, "w"), indent=1)
except OSError:
    pass
PYEOF
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_$TAG/ 2>/dev/null
ls -la $R/gpurun_out/profiles_$TAG
