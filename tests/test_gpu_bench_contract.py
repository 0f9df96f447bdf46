"""bench.py's one JSON line: the keys the driver reads, consistent with each other (a short run of the real command)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "7", "--warmup", "2", "--no-sampler", "--no-extras",
           "--cpu-sample", "20000"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                        # ONE line on stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 7 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    n = d["config"]["evals_per_step_per_gpu"]
    assert d["value"] == pytest.approx(n / (d["ms_per_step"] * 1e-3), rel=1e-6)          # whole-job rate = evals / step time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.3 < r["frac"] < 1.0
    assert r["achieved"] == pytest.approx(r["bytes_per_eval"] * n / (r["kernel_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02                                     # the kernel fits inside the step
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["bytes_per_eval"] * n
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == d["unit"]
    assert d["parity_max_rel_vs_oracle"] < 1e-10
