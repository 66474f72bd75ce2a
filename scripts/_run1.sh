cd $GRAFT_REPO_ROOT
timeout 2400 bash scripts/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1 < /dev/null
tail -30 gpurun_out/profile_round_r04.log
