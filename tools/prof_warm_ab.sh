cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d; mkdir -p $O
for v in base hk; do
  if [ $v = base ]; then unset PDA_HIP_LIB; else export PDA_HIP_LIB=$R/pda_amd/csrc/variants/libpda_hip_$v.so; fi
  (cd $R && ONLY_ORDER=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o t -- python tools/time_v4.py c3 262144 1 v4 > $O/$v.log 2>&1)
  echo "== $v" >> $O/stats.txt
  head -6 $O/$v/t_kernel_stats.csv | cut -c1-200 >> $O/stats.txt
done
cat $O/stats.txt
