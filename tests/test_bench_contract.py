"""The one JSON line bench.py prints, as the driver reads it: checked on the lines committed under profiles/ (no GPU here
to print a fresh one).  A line that loses a field of the contract - or reports a rate its own step time does not give -
would otherwise only be noticed by the driver at round end."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = sorted(glob.glob(os.path.join(ROOT, "profiles", "r05*bench*.json")))


def _line(path):
    txt = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(txt[-1])


def test_committed_lines_exist():
    assert len(LINES) >= 4


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_bench_line_contract(path):
    d = _line(path)
    for key, typ in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict)):
        assert isinstance(d[key], typ), key
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["metric"].startswith("Groth16 proofs/sec") and d["unit"] == "proofs/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and "workload" in d["config"]
    # value = whole-job proofs / the timed region: proofs per GPU and step x GPUs / step time
    per_step = d["config"]["proofs_per_gpu_per_step"] * d["n_gpus"]
    assert abs(d["value"] - per_step / (d["ms_per_step"] / 1e3)) <= 0.01 * d["value"]
    if "roofline" in d and d["roofline"]:
        r = d["roofline"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in r, key
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
        # achieved = algorithmic bytes per launch / the launch's average duration
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 0.01 * r["achieved"]
    if "cpu_baseline" in d and d["cpu_baseline"]:
        c = d["cpu_baseline"]
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in c, key
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1
