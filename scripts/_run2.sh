cd $GRAFT_REPO_ROOT
timeout 900 python tests/probes/fuzz_zb.py 150 2>&1 | tail -3
timeout 600 python tests/probes/fuzz_zstd_frames.py 100 2>&1 | tail -2
timeout 600 python tests/probes/fuzz_decode.py 2>&1 | tail -2
timeout 600 python tests/probes/fuzz_nested.py 2>&1 | tail -2
