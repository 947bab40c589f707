cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu --timeout 900 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench_r2c.json 2> gpurun_out/bench_r2c.err; tail -c 300 gpurun_out/bench_r2c.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r2c.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_of_measured"), d.get("ordered_sweep",{}).get("value"))
PY
timeout 600 python bench.py --workload c5shard --no-train --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5', d['value'], d['roofline']['frac'], d.get('ordered_sweep',{}).get('value'))"
