set -u
TAG=r03
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/c5_kt $OUT/c5_pmc
rocprofv3 --kernel-trace --stats -d $OUT/c5_kt -o b -- python $R/bench.py --only c5 --no-cpu-baseline > $OUT/c5_stdout.txt 2> $OUT/c5_kt.err
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $OUT/c5_pmc/$c -o b -- python $R/bench.py --only c5 --no-cpu-baseline > /dev/null 2> $OUT/c5_pmc_$c.err
done
python $R/scripts/summarize_rocpd.py $OUT $TAG c5 "{\"workload\": \"c5 (bench.py, see profiles/${TAG}_c5_stdout.txt)\"}" > $OUT/c5_summary.txt 2>&1
cp $OUT/c5_stdout.txt $R/profiles/${TAG}_c5_stdout.txt
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_c5* $R/gpurun_out/profiles_$TAG/
