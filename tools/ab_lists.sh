mkdir -p gpurun_out/$1
python -m pytest tests/test_gpu_score_topk.py tests/test_gpu_golden.py -x -q > gpurun_out/$1/pytest_topk.log 2>&1; echo "topk rc=$?" > gpurun_out/$1/rc.txt
for v in lds hbm lds hbm; do
  echo "== lists in $v" >> gpurun_out/$1/ab.txt
  PDA_SCORE_LISTS=$v ONLY_ORDER=1 python tools/time_v4.py c3 262144 1 v4 2>&1 | grep head >> gpurun_out/$1/ab.txt
done
PDA_SCORE_LISTS=lds python tools/prof4.py order 1 262144 > gpurun_out/$1/prof_lds.txt 2>&1
PDA_SCORE_LISTS=hbm python tools/prof4.py order 1 262144 > gpurun_out/$1/prof_hbm.txt 2>&1
cat gpurun_out/$1/rc.txt gpurun_out/$1/ab.txt gpurun_out/$1/prof_lds.txt gpurun_out/$1/prof_hbm.txt
