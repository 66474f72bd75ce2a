cd $GRAFT_REPO_ROOT
for f in c5_i64_32k c5_str_32k; do for b in 1 2048; do echo "== $f blocks $b"; timeout 120 scripts/micro/lz4_probe.bin scripts/micro/$f.raw $b 2>&1 | grep -A12 "zstd encode"; done; done
