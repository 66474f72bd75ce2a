cd $GRAFT_REPO_ROOT
timeout 2000 bash scripts/pmc_issue.sh r04 2>&1 | tail -70
