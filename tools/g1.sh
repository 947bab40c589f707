cd $GRAFT_REPO_ROOT
timeout 300 python tools/dbg_v4.py 2>&1 | grep -c OK
timeout 300 python tools/dbg_v4.py 2>&1 | grep -v OK | tail -3
timeout 300 python tools/time_variants.py base,abl1 order 1 2>&1 | tail -2
timeout 300 python tools/prof4.py order 1 2>&1 | tail -4
