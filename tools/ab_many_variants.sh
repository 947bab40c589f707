# A/B of build variants on the many-candidates geometry (C3, 262 144 users, raw head by norm + natural order); usage: bash tools/ab_many_variants.sh <out dir> <variant> ...
out=gpurun_out/$1; shift
mkdir -p $out
export PDA_SCORE_LISTS=many
for v in base "$@" base; do
  if [ $v = base ]; then unset PDA_HIP_LIB; else export PDA_HIP_LIB=$PWD/pda_amd/csrc/variants/libpda_hip_$v.so; fi
  echo "== $v" >> $out/ab.txt
  python tools/time_v4.py c3 262144 0 v4 2>&1 | grep "dense ordered" >> $out/ab.txt
  python tools/time_v4.py c3 262144 1 v4 2>&1 | grep "dense natural" >> $out/ab.txt
done
unset PDA_HIP_LIB
cat $out/ab.txt
