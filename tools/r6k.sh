cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6k
for c in "c2 50000" "c1 47890" "c3 2048" "c3 262144"; do set -- $c; timeout 120 python tools/time_warm.py $1 $2 2>&1 | grep warm-up; done
timeout 900 python -m pytest tests/test_gpu_score_topk.py tests/test_gpu_hot_items.py tests/test_gpu_funnel.py tests/test_gpu_full_size.py tests/test_gpu_end_to_end.py -x -q -m gpu 2>&1 | tail -4
timeout 200 python tools/time_huge.py c2 50000 auto 2>&1 | grep "users auto"
timeout 200 python tools/time_huge.py c1 47890 auto 2>&1 | grep "users auto"
timeout 200 python tools/block2048.py c3 2>&1 | tail -2
timeout 300 python tools/time_v4.py c2 50000 1 v4 2>&1 | grep -E "early stop|dense ordered"
timeout 300 python tools/time_v4.py c3 262144 1 v4 2>&1 | grep -E "early stop"
