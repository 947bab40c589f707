cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu --timeout 900 2>&1 | tail -4
timeout 600 python bench.py > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err; tail -c 600 gpurun_out/bench_r2b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r2b.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_of_measured"), d.get("ordered_sweep",{}).get("value"), d.get("natural_order",{}))
PY
