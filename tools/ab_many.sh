# A/B: LDS geometry vs the many-candidates geometry on the candidate-heavy sweeps (C3, 262 144 users); usage: bash tools/ab_many.sh <out dir>
out=gpurun_out/$1
mkdir -p $out
for v in lds many; do
  echo "== $v" >> $out/ab.txt
  PDA_SCORE_LISTS=$v python tools/time_v4.py c3 262144 10 v4 2>&1 | grep head >> $out/ab.txt
done
cat $out/ab.txt
