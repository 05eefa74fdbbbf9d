"""`-m gpu`: the contract of bench.py's JSON line at N = 1 on a small batch -- metric / value / unit / config.workload, the
`roofline` and `cpu_baseline` objects (with the calibration against the reference's published figure), and the `box`
record (pqp_box_calibrate before and after the timed region) with `value_normalised` beside `value`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "96", "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "48", "--mpc-steps", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    d = json.loads(lines[0])
    assert d["metric"].startswith("QPs/sec") and d["unit"] == "QPs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["unsolved"] == 0 and d["max_kkt_residual"] <= 1e-9 and d["value"] > 0
    assert abs(d["value"] - 96 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and 0 < roof["frac"] <= 1.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and isinstance(roof["kernel"], str)
    assert roof["kernel_ms"] <= d["ms_per_step"] * 1.001
    cpu = d["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["unit"] == "QPs/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["sample"]
    cal = cpu["calibration"]
    assert cal["published_ms"] == 0.5 and 0.05 < cal["port_ms"] < 20.0 and "benchmark_dense_qp.cpp" in cal["published_on"]
    box = d["box"]
    for when in ("before", "after"):
        c = box[when]
        assert 1000.0 < c["hbm_read_gbs"] < 8000.0 and c["chain_ms"] > 0 and c["valu_ms"] > 0 and "smi" in c
    assert 0.5 < box["slowdown_vs_reference_box"] < 2.0
    assert abs(d["value_normalised"] - d["value"] * box["slowdown_vs_reference_box"]) <= 1e-9 * d["value"]
    assert d["max_abs_delta_vs_cpu"]["cpu_all_solved"] and d["max_abs_delta_vs_cpu"]["x"] <= 1e-9
