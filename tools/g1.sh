cd $GRAFT_REPO_ROOT
echo "1x1: 262144 users x 200000 items"; timeout 300 python tools/time_v4.py c3 262144 1 v4 f32 262144 200000 2>&1 | grep -E "ordered|early"
echo "1x8: 262144 users x 25000 items"; timeout 300 python tools/time_v4.py c3 262144 1 v4 f32 262144 25000 2>&1 | grep -E "ordered|early"
echo "2x4: 131072 users x 50000 items"; timeout 300 python tools/time_v4.py c3 131072 1 v4 f32 131072 50000 2>&1 | grep -E "ordered|early"
echo "4x2: 65536 users x 100000 items"; timeout 300 python tools/time_v4.py c3 65536 1 v4 f32 131072 100000 2>&1 | grep -E "ordered|early"
echo "8x1: 32768 users x 200000 items"; timeout 300 python tools/time_v4.py c3 32768 1 v4 f32 131072 200000 2>&1 | grep -E "ordered|early"
