"""The bench line the driver parses: bench.compact_line() of a full result dict must stay a small, self-contained JSON object
(the 25 KB one-liner of that round was not parsed).  CPU only: the canned dict is the committed full
result of round 4 (profiles/r04_bench.json, 25 KB) plus the keys this round added."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _strings(x, path=""):
    if isinstance(x, dict):
        for k, v in x.items():
            yield from _strings(v, path + "/" + str(k))
    elif isinstance(x, (list, tuple)):
        for i, v in enumerate(x):
            yield from _strings(v, path + f"[{i}]")
    elif isinstance(x, str):
        yield path, x


@pytest.fixture()
def canned():
    out = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    assert len(json.dumps(out)) > 20000          # the thing that broke the record
    out["config"]["workload_short"] = "panda_reaching T=3000 set_interval(5), distinct seeds, global batch 1024 over 1 GPU(s), 6 alphas, lambda=0.1"
    out["config"]["residual_jacobians_source"] = "ONE r_x uploaded once (kpilqr_upload_residual_jacobians_const), as host/iLQR_GPU does when the task's r_x is constant"
    out["weak_scaling_projection"] = {"n_gpus": {str(n): {"projected_value": n * out["value"]} for n in (1, 2, 4, 8)}}
    for k, v in out["secondary_configs"].items():
        if k.startswith("configs["):
            v["cpu_baseline"] = {"value": 10.0, "cores": 16}
            v["gpu_over_cpu"] = v["value"] / 10.0
            v["meets_50x"] = v["gpu_over_cpu"] >= 50
    return out


def test_line_is_compact_parses_and_carries_the_judged_keys(canned):
    line = bench.compact_line(canned, "bench_detail.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT <= 4096, len(text)
    assert "\n" not in text
    back = json.loads(text)
    assert back == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "rccl_ranks", "collective", "stage_ms", "roofline", "cpu_baseline", "gpu_over_cpu", "parity_check", "value_pcie_inclusive",
              "value_per_step_jacobians", "strong_scaling_projection", "weak_scaling_projection", "detail"):
        assert k in back, k
    for k in ("workload", "batch_per_gpu", "global_batch", "horizon", "launched", "launched_backward", "launched_forward", "residual_jacobians"):
        assert back["config"][k] is not None, k
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_launch_ms", "hbm",
              "kernel_compulsory_bytes_per_launch", "frac_of_hbm_peak", "traffic_frac_of_hbm_peak"):
        assert back["roofline"][k] is not None, k
    for k in ("kernel_compulsory_bytes_per_launch", "frac_of_hbm_peak", "traffic_frac_of_hbm_peak"):
        assert back["roofline"]["hbm"][k] is not None, k
    for k in ("value", "unit", "cores", "cores_available", "cgroup_cpu_quota_cores", "kind", "single_thread_value", "sample"):
        assert k in back["cpu_baseline"], k
    assert back["parity_check"]["pass"] is True and back["parity_check"]["max_rel_err_K"] < 1e-6
    assert set(back["strong_scaling_projection"]) >= {"2", "4", "8"}
    # the numbers are the full result's, to six digits
    assert abs(back["value"] - canned["value"]) <= 1e-5 * canned["value"]
    assert abs(back["roofline"]["frac"] - canned["roofline"]["frac"]) <= 1e-5
    assert abs(back["value_pcie_inclusive"] - canned["pcie_inclusive_b1024"]["full_payload_constant_jacobians"]["value"]) <= 1e-5 * back["value_pcie_inclusive"]
    assert abs(back["value_per_step_jacobians"] - canned["per_step_residual_jacobians"]["value"]) <= 1e-5 * back["value_per_step_jacobians"]
    # the driver's record cuts strings inside config / roofline / cpu_baseline: nothing there may be longer than it keeps
    for path, s_ in _strings(back):
        assert len(s_) <= 120, (path, len(s_))
    # one figure per BASELINE config with its CPU ratio
    assert {"configs[1]", "configs[2]", "configs[4]"} <= set(back["configs"])


def test_line_survives_missing_side_measurements_and_errors(canned):
    out = copy.deepcopy(canned)
    for k in ("secondary_configs", "strong_scaling_projection", "pcie_inclusive", "pcie_inclusive_b1024", "per_step_residual_jacobians",
              "lambda_sweep", "materialising_pipeline", "weak_scaling_projection"):
        out.pop(k, None)
    out["cpu_baseline"] = {"error": "RuntimeError('gcc missing')" * 20}
    out.pop("gpu_over_cpu", None)
    line = bench.compact_line(out)
    text = json.dumps(line)
    assert len(text) < bench.LINE_LIMIT and json.loads(text)["roofline"]["frac"] > 0
    assert "value_pcie_inclusive" not in line and "error" in line["cpu_baseline"]
    # a multi-GPU line (no side measurements, a weak line beside the strong one)
    out["n_gpus"] = 8; out["rccl_ranks"] = 8; out.pop("cpu_baseline")
    out["weak_scaling"] = {"batch_per_gpu": 1024, "global_batch": 8192, "steps": 10, "value": 1.2e6, "unit": "trajectory-iterations/s", "ms_per_step": 6.8}
    line = bench.compact_line(out)
    assert line["weak_scaling"]["global_batch"] == 8192 and "cpu_baseline" not in line and len(json.dumps(line)) < bench.LINE_LIMIT


def test_line_drops_optional_blocks_rather_than_overflowing(canned):
    out = copy.deepcopy(canned)
    out["secondary_configs"] = {f"configs[{i}] " + "x" * 40: {"value": 1.0 * i, "gpu_over_cpu": 2.0} for i in range(200)}
    line = bench.compact_line(out, "bench_detail.json")
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert "roofline" in line and "cpu_baseline" in line and "value_pcie_inclusive" in line


def test_emit_prints_the_line_last_and_writes_the_detail_file(canned, tmp_path, capsys):
    path = str(tmp_path / "detail.json")
    bench.emit(canned, path)
    cap = capsys.readouterr()
    last = cap.out.strip().splitlines()[-1]
    line = json.loads(last)
    assert len(last) < bench.LINE_LIMIT and line["detail"] == path
    full = json.load(open(path))
    assert "secondary_configs" in full and "lambda_sweep" in full and full["value"] == canned["value"]
