# A/B of the many-candidates geometry on smaller blocks; usage: bash tools/ab_many2.sh <out dir>
out=gpurun_out/$1
mkdir -p $out
for v in lds many; do
  echo "== c3 65536 $v" >> $out/ab2.txt
  PDA_SCORE_LISTS=$v python tools/time_v4.py c3 65536 10 v4 2>&1 | grep head | grep -v "early" >> $out/ab2.txt
  echo "== c2 all users $v" >> $out/ab2.txt
  PDA_SCORE_LISTS=$v python tools/time_v4.py c2 65536 10 v4 2>&1 | grep head | grep -v "early" >> $out/ab2.txt
  echo "== c1 all users $v" >> $out/ab2.txt
  PDA_SCORE_LISTS=$v python tools/time_v4.py c1 65536 10 v4 2>&1 | grep head | grep -v "early" >> $out/ab2.txt
done
cat $out/ab2.txt
