cd $GRAFT_REPO_ROOT
timeout 1500 bash tools/profile_round.sh r2b_c3 > gpurun_out/r2b_c3.log 2>&1
timeout 1500 bash tools/profile_round.sh r2b_c5 --workload c5shard > gpurun_out/r2b_c5.log 2>&1
tail -n 3 gpurun_out/r2b_c3.log gpurun_out/r2b_c5.log
