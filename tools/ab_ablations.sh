mkdir -p gpurun_out/r3f
for v in base abl1 abl2 abl4 abl6 abl8 abl15 base; do
  if [ $v = base ]; then unset PDA_HIP_LIB; else export PDA_HIP_LIB=$PWD/pda_amd/csrc/variants/libpda_hip_$v.so; fi
  echo "== $v" >> gpurun_out/r3f/abl.txt
  ONLY_ORDER=1 python tools/time_v4.py c3 262144 1 v4 2>&1 | grep head >> gpurun_out/r3f/abl.txt
done
unset PDA_HIP_LIB
python tools/prof4.py order 1 262144 > gpurun_out/r3f/prof.txt 2>&1
cat gpurun_out/r3f/abl.txt gpurun_out/r3f/prof.txt
