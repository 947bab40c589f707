mkdir -p gpurun_out/$1
python -m pytest tests/test_gpu_score_topk.py tests/test_gpu_golden.py -x -q > gpurun_out/$1/pytest_topk.log 2>&1; echo "topk rc=$?" > gpurun_out/$1/rc.txt
for v in lds wide lds wide; do
  echo "== geometry $v" >> gpurun_out/$1/ab.txt
  PDA_SCORE_LISTS=$v python tools/time_v4.py c3 262144 1 v4 2>&1 | grep head | grep -v natural >> gpurun_out/$1/ab.txt
done
for v in lds wide; do echo "== c3 131072 $v" >> gpurun_out/$1/ab.txt; PDA_SCORE_LISTS=$v ONLY_ORDER=1 python tools/time_v4.py c3 131072 1 v4 2>&1 | grep head >> gpurun_out/$1/ab.txt; done
for v in lds wide; do echo "== c3 bf16 262144 $v" >> gpurun_out/$1/ab.txt; PDA_SCORE_LISTS=$v ONLY_ORDER=1 python tools/time_v4.py c3 262144 1 v4 bf16 2>&1 | grep head >> gpurun_out/$1/ab.txt; done
tail -4 gpurun_out/$1/pytest_topk.log; cat gpurun_out/$1/rc.txt gpurun_out/$1/ab.txt
