"""The driver parses the LAST stdout line of bench.py; round 5's line had grown to 28 KB and was not extracted.  The line
builder must turn a full result structure (round 5's own, kept under profiles/) into one JSON line below 4 KB that still
carries the contract's keys, `roofline` and `cpu_baseline`."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _canned():
    with open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_summary_line_is_short_and_complete():
    import bench
    full = _canned()
    assert len(json.dumps(full)) > 20000
    line = bench.summary_line(full, "gpurun_out/bench_detail_n1.json")
    assert "\n" not in line and len(line) < 4096, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "legs", "detail"):
        assert k in d, k
    assert d["value"] == float(f"{full['value']:.7g}")
    assert set(d["roofline"]) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"}
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert d["parity"] == {"checked": full["parity_checked_reads"], "mismatches": full["mismatches"]}
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["legs"]["greedy"]["value"] == float(f"{full['greedy']['value']:.5g}")


def test_summary_line_survives_infinities_long_strings_and_eight_ranks():
    import bench
    full = copy.deepcopy(_canned())
    full["roofline"]["achieved"] = float("inf")
    full["roofline"]["frac"] = float("nan")
    full["config"]["workload"] = "x" * 5000
    full["cpu_baseline"]["sample"] = "y" * 5000
    full["n_gpus"] = 8
    full["verbose"] = {"value": 2.5e6, "unit": "reads/s", "parity": {"checked": 280000, "mismatches": 0}}     # (round 6's leg)
    full["config"]["per_rank_units_per_s"] = [3.9e8 + k for k in range(8)]
    line = bench.summary_line(full, None)
    assert len(line) < 4096
    d = json.loads(line)                                   # strict JSON: no Infinity / NaN tokens
    assert "Infinity" not in line and "NaN" not in line
    assert d["roofline"]["achieved"] is None and d["roofline"]["frac"] is None
    assert len(d["config"]["per_rank_units_per_s"]) == 8
    assert d["legs"]["verbose"]["value"] == 2.5e6


def test_detail_file_holds_the_full_structure(tmp_path, capsys):
    import bench
    full = _canned()
    p = bench.write_detail(full, str(tmp_path), 1)
    assert p is not None
    path = p if os.path.isabs(p) else os.path.join(ROOT, p)
    with open(path) as f:
        back = json.load(f)
    assert back["roofline"]["ops_per_unit"] == full["roofline"]["ops_per_unit"]
    assert "[bench] detail:" in capsys.readouterr().err
