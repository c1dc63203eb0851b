"""The driver's bench contract, checked without a GPU: the committed round line (profiles/r03_bench.json) carries every field the
contract names, the roofline arithmetic of bench.py is what DESIGN.md 4 says it is, and `value` is frames per second of the whole job."""
import argparse
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r03_bench.json")) as f:
        return json.loads(f.read())


def test_committed_bench_line_has_the_contract_fields():
    j = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "frames/s" and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["data"] == "synthetic" and j["dtype"] == "f32" and j["n_gpus"] == 1
    assert "workload" in j["config"] and "BASELINE configs[2]" in j["config"]["workload"] and "model" not in j["config"]
    # value = frames of the whole job / wall time: global batch / ms_per_step
    assert j["value"] == pytest.approx(j["config"]["global_batch"] / (j["ms_per_step"] * 1e-3), rel=2e-3)
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "family_frac", "step_frac"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == bench.MFMA_F32_PEAK_TFLOPS
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=2e-4)
    assert 0 < r["step_frac"] < r["family_frac"] < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_mb_per_launch"] * 1e6 * 0.9   # PMC bytes >= algorithmic bytes
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["unit"] == "frames/s" and c["cores"] >= 1


def test_traced_roofline_arithmetic():
    """achieved = algorithmic flops of the traced launches / their HIP-event durations; frac = achieved / dense fp32-MFMA peak."""
    args = argparse.Namespace(operand="f32")
    trace = [{"ms": 0.125, "flops": 8.0e9, "bytes": 6.0e7, "windowed": False}, {"ms": 0.135, "flops": 8.2e9, "bytes": 6.1e7, "windowed": False}]
    r = bench._traced_roofline(trace, args, "fwd", "64", "32", pmc=False)
    tf = (8.0e9 + 8.2e9) / (0.260e-3) / 1e12
    assert r["achieved"] == pytest.approx(tf, rel=1e-3) and r["frac"] == pytest.approx(tf / bench.MFMA_F32_PEAK_TFLOPS, abs=1e-4)
    assert r["launches"] == 2 and r["avg_us"] == pytest.approx(130.0, rel=1e-3) and r["traffic"] is None
    assert bench._traced_roofline([], args, "fwd", "64", "32", pmc=False) is None
    # round 6: the bound the conv kernels really run against -- bytes of gathered rows / time (a weight gradient gathers two rows per pair)
    tr = [{"ms": 0.1, "flops": 1.0, "bytes": 1.0, "windowed": False, "pairs": 1_000_000, "ck": 32, "cn": 32, "dir": "dw"}]
    r = bench._traced_roofline(tr, args, "dw", "32", "32", pmc=False)
    assert r["l2_gather_tb_s"] == pytest.approx(1e6 * 256 / 1e-4 / 1e12, rel=1e-3)
    assert r["l2_gather_frac_of_9p9_tb_s"] == pytest.approx(r["l2_gather_tb_s"] / bench.L2_GATHER_CEILING_TBS, abs=1e-3)


def test_every_line_of_the_round_file_parses_and_names_its_workload():
    n = 0
    with open(os.path.join(ROOT, "profiles", "r03_bench_lines.txt")) as f:
        for line in f:
            if not line.startswith("{"):
                continue
            j = json.loads(line)
            n += 1
            assert j["unit"] == "frames/s" and "workload" in j["config"] and "roofline" in j and "cpu_baseline" in j
            assert j["value"] > 0 and j["ms_per_step"] > 0
    assert n >= 10


def test_round5_line_names_the_dominant_kernel_flat_and_the_host_side():
    """the driver's record keeps the LEADING scalar keys of `roofline`; they must be the dominant kernel's (measured
    inside the timed steps), both fractions and the step fraction; `host_enqueue_ms_per_step` is a top-level key."""
    with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:
        j = json.loads(f.read())
    assert j["unit"] == "frames/s" and j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5
    assert j["value"] == pytest.approx(j["config"]["global_batch"] / (j["ms_per_step"] * 1e-3), rel=2e-3)
    assert 0 < j["host_enqueue_ms_per_step"] < 2 * j["ms_per_step"]
    r = j["roofline"]
    lead = list(r.keys())[:12]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "frac_issue_pipe", "step_frac", "family_frac", "kernel_is_dominant"):
        assert k in lead, (k, lead)
    assert r["kernel"].startswith("bwd_weight_kernel") and r["kernel_is_dominant"] is True
    assert r["kernel"] == r["dominant_by_time"]["kernel"]
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=2e-4) and r["peak"] == bench.MFMA_F32_PEAK_TFLOPS
    assert r["frac_issue_pipe"] == pytest.approx(r["achieved"] / (bench.MFMA_16BIT_PEAK_TFLOPS / 6), abs=2e-4)
    assert 0 < r["frac_issue_pipe"] < r["step_frac"] < r["frac"] < r["family_frac"] + 0.1 < 1
    assert r["traffic"] > r["algorithmic_mb_per_launch"] * 1e6 * 0.9 and "r05_traffic_bwd_weight_32_32" in r["traffic_source"]
    assert j["exact_f32_mfma"]["ms_per_step"] > j["ms_per_step"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
