mkdir -p gpurun_out/r5i
for v in base l3 base; do
  if [ $v = base ]; then unset PDA_HIP_LIB; else export PDA_HIP_LIB=$PWD/pda_amd/csrc/variants/libpda_hip_$v.so; fi
  echo "== $v" >> gpurun_out/r5i/ab.txt
  python tools/time_v4.py c5shard 262144 1 v4 bf16 2>&1 | grep "dense ordered\|early stop" >> gpurun_out/r5i/ab.txt
done
export PDA_HIP_LIB=$PWD/pda_amd/csrc/variants/libpda_hip_l3.so
python -m pytest tests/test_gpu_score_topk.py -x -q -m gpu -k "c5_shard or 256" 2>&1 | tail -2 >> gpurun_out/r5i/ab.txt
cat gpurun_out/r5i/ab.txt
