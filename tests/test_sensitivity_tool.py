"""The quantizer-sensitivity sweep (tools/sensitivity_sweep.py; full 2000-query result committed as
profiles/r02_quantizer_sensitivity.json) stays runnable: a 25-query run on the CPU oracle."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sensitivity_sweep_runs_and_the_committed_result_is_small():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "sensitivity_sweep.py"), "25"], timeout=600)
    j = json.loads(out)
    assert len(j["results"]) == 4
    for r in j["results"]:
        assert r["queries"] + r["skipped_qmax_too_high"] == 25
        for k in ("mode0", "qmax_plus_ulp", "qmax_minus_ulp"):
            assert 0.0 <= r[k]["frac_queries_key_set_differs"] <= 0.2
    full = json.load(open(os.path.join(ROOT, "profiles", "r02_quantizer_sensitivity.json")))
    for r in full["results"]:
        assert r["queries"] >= 2000
        for k in ("mode0", "qmax_plus_ulp", "qmax_minus_ulp"):
            assert r[k]["frac_queries_key_set_differs"] <= 0.005      # what DESIGN.md section 6 quotes
