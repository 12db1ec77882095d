"""The bench.py contract (one JSON line with the driver's fields, roofline and cpu_baseline) and the
__graft_entry__ smoke run, exercised end to end on the GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_bench(*extra, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "150", "--warmup", "50",
                          "--kernel-timing-launches", "100", *extra], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = _run_bench("--cpu-baseline-seconds", "2")
    assert d["metric"] == "objective-fn evals/sec" and d["unit"] == "evals/s" and d["higher_is_better"] is True
    assert (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["dtype"], d["data"]) == (1, 150, 50, "weak", "f64",
                                                                                          "synthetic")
    assert d["vs_baseline"] is None and d["config"]["workload"] == "de_rosenbrock_n128_p4096"
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-9
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.05 < r["frac"] < 1.0
    assert r["algorithmic_bytes_per_launch"] == 4112 * 4096 and r["traffic"] is not None
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "evals/s" and c["value"] > 0 and c["sample"]
    assert d["value"] > 100 * c["value"]  # north star: >= 10x the host CPU path


def test_bench_other_workload_and_smoke():
    d = _run_bench("--no-cpu-baseline", "--workload", "de_rosenbrock_n1024_p16384")
    assert "cpu_baseline" not in d and d["config"]["dim"] == 1024 and d["roofline"]["frac"] > 0.3
    import __graft_entry__

    __graft_entry__.smoke()
