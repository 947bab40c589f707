cd $GRAFT_REPO_ROOT
timeout 300 python tools/time_variants.py base,abl1,abl128,abl256 order 1 2>&1 | tail -4
