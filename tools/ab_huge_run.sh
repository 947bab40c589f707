for t in base or1 noexit nobar nometa nodma notest bare; do
  if [ $t = base ]; then L=; else L=pda_amd/csrc/ab/libpda_hip_$t.so; fi
  echo "== $t"; PDA_HIP_LIB=$L timeout 120 python tools/time_huge.py c3 262144 huge 2>&1 | tail -1
done
