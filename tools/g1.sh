cd $GRAFT_REPO_ROOT
timeout 120 tools/ubench/mfma_lds
