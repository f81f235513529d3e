"""The sensitivity sweep of the parity-unpinned float half (tools/sensitivity_sweep.py; full result committed as
profiles/r03_table_noise_sensitivity.json) stays runnable: a 25-query run on the CPU oracle over ten times less data."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = ("mode0", "qmax_plus_ulp", "qmax_minus_ulp", "qmax_plus_4ulp", "qmax_minus_4ulp", "tables_1ulp", "tables_4ulp")


def test_sensitivity_sweep_runs_and_the_committed_result_is_small():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "sensitivity_sweep.py"), "25", "small"], timeout=600)
    j = json.loads(out)
    assert len(j["results"]) == 6
    for r in j["results"]:
        assert r["queries"] + r["skipped_qmax_too_high"] == 25
        for k in KINDS:
            assert 0.0 <= r[k]["frac_queries_key_set_differs"] <= 0.3
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_table_noise_sensitivity.json")))
    assert len(full["results"]) == 6
    for r in full["results"]:
        assert r["queries"] >= 1000
        for k in KINDS:
            assert r[k]["frac_queries_key_set_differs"] <= 0.02       # what DESIGN.md section 6 quotes (worst case there: < 1 %)
