# A/B: warm-up lists handed over through the workspace (K .. CAP keys per row) vs through out_keys (exactly K); usage: bash tools/ab_handover.sh <out dir>
out=gpurun_out/$1; mkdir -p $out
for v in base hk; do
  if [ $v = base ]; then unset PDA_HIP_LIB; else export PDA_HIP_LIB=$PWD/pda_amd/csrc/variants/libpda_hip_$v.so; fi
  echo "== $v (hk = hand-over through out_keys)" >> $out/ab.txt
  python tools/time_v4.py c3 262144 1 v4 2>&1 | grep head | grep -v natural >> $out/ab.txt
  python tools/time_v4.py c2 65536 1 v4 2>&1 | grep head | grep -v natural >> $out/ab.txt
done
unset PDA_HIP_LIB
cat $out/ab.txt
