python -m pytest tests/test_gpu_funnel.py tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-train > gpurun_out/r5x_bench.json 2> gpurun_out/r5x_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5x_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["raw_head"]["ms_per_step"], d["raw_head"]["roofline_frac"])
for c in ("c1","c2"):
    e=d["per_config"][c]["eval"]; print(c, e["ms_per_step"], e["roofline_frac"], e["raw_head"])
PY
