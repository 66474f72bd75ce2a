cd $GRAFT_REPO_ROOT
fails=0
for i in $(seq 0 62); do timeout 200 python tests/probes/fuzz_decode.py $i 40 > /tmp/fz_$i.txt 2>&1 || { echo "CASE $i FAILED"; tail -3 /tmp/fz_$i.txt; fails=$((fails+1)); }; done
echo "fuzz_decode: $fails failed cases"
tail -1 /tmp/fz_5.txt /tmp/fz_30.txt /tmp/fz_60.txt
