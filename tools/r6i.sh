cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6i
for wl in "c2 50000" "c1 47890" "c3 2048" "c3 8192" "c3 65536" "c3 262144"; do set -- $wl; timeout 200 python tools/time_funnel.py $1 $2 8 2>&1 | grep -E "schedule|raw head" | cut -c1-200; done
timeout 600 python -m pytest tests/test_gpu_funnel.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/soak_funnel.py 120 2>&1 | tail -4
timeout 600 python tools/warm_tiles_small.py 2>&1 | tee gpurun_out/r6i/warm_tiles_small.txt | tail -40
bash tools/pmc_adam_small.sh r6i/adam_pmc 2>&1 | tail -60
