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


def _load_bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.mark.parametrize("log", ["profiles/r03/bench_round3.log", "profiles/r03/bench_rmat24_torchrun1.log",
                                 "profiles/r02/bench_round2.log"])
def test_final_line_stays_small_and_complete(log):
    """the 20 KB line was dropped by the driver. Whatever the full record holds (these are the largest ones
    on file), the stdout line is < 4 KB, parses, and carries the contract's keys with roofline / cpu_baseline intact."""
    bench = _load_bench()
    lines = [ln for ln in open(os.path.join(ROOT, log)).read().splitlines() if ln.startswith("{")]
    full = json.loads(lines[-1])
    # the blocks round 4 adds, at their largest
    full["series"] = {"com-amazon-like": {k: 123456.789012 for k in (
        "kernel_us", "gflops", "frac", "traffic", "l2_hit_rate", "plan_ms", "plain_call_kernel_us", "plain_call_frac",
        "gflops_incl_plan_over_200_launches", "traffic_floor", "ceiling_frac", "achieved_over_ceiling")}}
    full["widths"] = {w: {k: 123456.789012 for k in ("kernel_us", "gflops", "frac", "traffic", "l2_hit_rate")}
                      for w in ("N32", "N512", "N128_plain_call")}
    full["reference_kernel"] = {"what": "x" * 60, "kernel_us": 276.123456, "gflops": 1715.123456, "product_bits_equal": True}
    full["roofline"].update({"ceiling_frac": 0.531234567, "achieved_over_ceiling": 0.851234567, "traffic_floor": 474000000,
                             "ceiling_note": "y" * 200, "launches": 200})
    if "extra" in full:
        full["other_configs"] = bench.other_configs(full["extra"])
    full["extra_file"] = bench.EXTRA_FILE
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_LIMIT == 4096 and "\n" not in line
    rec = json.loads(line)
    for k in REQUIRED:
        assert k in rec, k
    assert "extra" not in rec
    assert "workload" in rec["config"] and "model" not in rec["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rec["roofline"], k
    assert abs(rec["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-4 * full["roofline"]["frac"]
    assert abs(rec["value"] - full["value"]) < 1e-4 * full["value"]
    if full.get("cpu_baseline"):
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in rec["cpu_baseline"], k


def test_self_launch_command_line():
    """`python bench.py --gpus 8` without a launcher re-executes itself under torch.distributed.run (one rank per GPU)."""
    bench = _load_bench()
    argv = bench.self_launch_argv(8, ["--gpus", "8", "--steps", "5"])
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert argv[argv.index("--nproc-per-node") + 1] == "8" and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(argv[argv.index("--master-port") + 1]) < 65536
    assert argv[-5] == os.path.join(ROOT, "bench.py") and argv[-4:] == ["--gpus", "8", "--steps", "5"]
