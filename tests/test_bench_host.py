"""Host-side checks of bench.py (no GPU): it must refuse to run without a HIP device (there is no CPU path to measure), and the
traffic records it quotes must exist with the fields it reads."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_bench_refuses_to_run_without_a_device():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "HIP device" in (r.stdout + r.stderr)
    assert not any(line.startswith("{") for line in r.stdout.splitlines())  # no JSON line without a measurement


def test_bench_help_and_traffic_records():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout
    rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    for key in ("com-amazon-like/N128/valued/plan", "com-amazon-like/N128/valued/plain", "com-amazon-sbm/N128/valued/plan"):
        e = rec[key]
        assert e["bytes_per_launch"] == int(round((2 * e["FETCH_SIZE_KiB"] + e["WRITE_SIZE_KiB"]) * 1024))
        assert 0.0 < e["l2_hit_rate"] < 1.0
        assert os.path.exists(os.path.join(ROOT, e["source"].split(" ")[0])), e["source"]
    # the algorithmic bytes of the BENCH workload against the recorded traffic: 2.7x / 3.1x / 1.5x
    alg = 4 * (334863 + 1) + 8 * 1851744 + 2 * 4 * 334863 * 128
    assert alg == 359053120
    assert 2.5 < rec["com-amazon-like/N128/valued/plan"]["bytes_per_launch"] / alg < 2.9
    assert 1.3 < rec["com-amazon-sbm/N128/valued/plan"]["bytes_per_launch"] / alg < 1.7
