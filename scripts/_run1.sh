cd $GRAFT_REPO_ROOT
for p in fuzz_decode fuzz_nested edge_alloc edge_out fuzz_big fuzz_zstd_frames fuzz_zb; do
  echo "== $p"; timeout 900 python tests/probes/$p.py 2>&1 | grep -v amdgpu.ids | tail -2
done
timeout 900 python tests/probes/big_pages.py 3000000 2>&1 | grep -v amdgpu.ids | tail -6
