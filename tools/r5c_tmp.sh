python -m pytest tests/test_gpu_funnel.py -x -q 2>&1 | tail -5
python -m pytest tests/test_gpu_full_size.py -x -q -k c5 2>&1 | tail -5
python bench.py --workload c5shard --no-train --no-cpu-baseline --no-per-config > gpurun_out/r5i_c5.json 2> gpurun_out/r5i_c5.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5i_c5.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"])
print({k:d["raw_head"][k] for k in d["raw_head"] if k in ("ms_per_step","roofline_frac","kernel_identity","exact_rescorings_per_user","rows_through_the_exact_fallback")})
PY
