cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q -m gpu --timeout 600 2>&1 | tail -2
timeout 900 python bench.py --no-per-config --no-cpu-baseline --no-train 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['traffic'], d['value'])"
