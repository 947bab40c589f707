"""CPU: the projection bench.py prints as its contract line, applied to the largest full record on file (round 5's 21 kB line,
which the driver could not parse: BENCH_r05.parsed == null), and the checker tools/profile_round.sh runs on the box."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_compact_line_of_a_full_c3_record_stays_small():
    """The projection applied to the largest record we have on file (round 5's unparsed 21 kB line)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    full = json.load(open(os.path.join(ROOT, "profiles", "round5d_c3_bench.json")))
    text = json.dumps(b.compact_contract_line(full), allow_nan=False)
    assert len(text) < 4096
    d = json.loads(text)
    assert d["roofline"]["frac"] > 0 and d["cpu_baseline"]["cores"] >= 1 and len(json.dumps(d["summary"])) < 1200
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from check_contract_line import check
    check("noise\n" + text + "\n\n")
