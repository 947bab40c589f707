#!/usr/bin/env python
"""What the driver ingests from bench.py: the LAST non-empty stdout line parses as JSON, stays under 8 kB, has no NaN / Infinity
token and carries roofline + cpu_baseline.  usage: check_contract_line.py <captured stdout file> [--no-cpu]"""
import json
import sys


def check(text, need_cpu=True):
    lines = [x for x in text.splitlines() if x.strip()]
    assert lines, "no stdout"
    last = lines[-1]
    assert len(last) < 8192, "contract line is %d bytes" % len(last)
    assert "NaN" not in last and "Infinity" not in last
    d = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["roofline"]["frac"] > 0
    if need_cpu:
        assert d["cpu_baseline"] and d["cpu_baseline"]["value"] > 0
    return d, len(last)


if __name__ == "__main__":
    d, n = check(open(sys.argv[1]).read(), need_cpu="--no-cpu" not in sys.argv)
    print("contract line ok: %d bytes, value %.4g %s, roofline.frac %.4f" % (n, d["value"], d["unit"], d["roofline"]["frac"]))
