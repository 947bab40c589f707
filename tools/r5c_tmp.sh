run() { echo "== $3 $4 TUNE=$1 TUNE2=$2"; FUNNEL_TUNE=$1 FUNNEL_TUNE2=$2 python tools/time_funnel.py $3 $4 6 2>&1 | grep "schedule\|funnel:" | sed 's/, kernel.*//'; }
run "" "" c5shard 262144
run "1e-2,4,64,4" "" c5shard 262144
run "3e-2,4,64,4" "" c5shard 262144
run "2e-2,4,64,4" "" c3 262144
run "5e-2,4,64,4" "" c3 262144
run "3e-2,4,64,4" "2,6,30" c3 262144
run "3e-2,3,64,4" "" c3 262144
run "3e-2,4,64,4" "" c3 131072
run "" "" c3 131072
run "3e-2,4,64,4" "" c3 65536
run "" "" c3 65536
