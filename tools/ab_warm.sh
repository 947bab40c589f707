mkdir -p gpurun_out/r4b
for v in base w4abl2 w4abl6 w4abl8 w4abl16 w4abl24 w4abl26; do
  if [ $v = base ]; then unset PDA_HIP_LIB; else export PDA_HIP_LIB=$PWD/pda_amd/csrc/variants/libpda_hip_$v.so; fi
  echo "== $v" >> gpurun_out/r4b/w4.txt
  python tools/time_warm.py c3 262144 2>&1 | grep warm-up >> gpurun_out/r4b/w4.txt
done
cat gpurun_out/r4b/w4.txt
