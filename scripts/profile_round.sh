#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: kernel stats and (separate runs, as MI355X_MICROARCH.md §HBM prescribes) the
# FETCH_SIZE / WRITE_SIZE counters, for the headline bench line and for the C1 / C3 configurations.
# Usage (on the GPU box, from the repo root): scripts/profile_round.sh r02
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d $OUT/${name}_kt -o b -- python $R/bench.py "$@" > $OUT/${name}_stdout.txt 2> $OUT/${name}_kt.err
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c -d $OUT/${name}_pmc/$c -o b -- python $R/bench.py "$@" > /dev/null 2> $OUT/${name}_pmc_$c.err
    done
}
run c2 --no-configs --no-cpu-baseline --steps 10
run c1 --only c1 --no-cpu-baseline
run c3 --only c3,c3_lz4 --no-cpu-baseline
find $OUT -name "*.csv" | head -50 > $OUT/files.txt
