"""bench.py's ONE JSON line stays machine-readable: the driver keeps an 8 126-character tail of stdout, so the line must be well under
that and must carry the contract's keys (round 5's 25 KB line left BENCH_r05.json with parsed = null).  Canned input: the full
document of a real run (profiles/r05_bench_prove2p20.json, prose and per-round tables included), so no GPU is needed."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (importing bench.py touches neither torch nor the GPU)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_prove2p20.json")))


def _check(line):
    assert "\n" not in line
    assert len(line) < bench.LINE_LIMIT < 8126 - 1500, len(line)
    d = json.loads(line)
    for k in REQUIRED:
        assert k in d, k
    assert list(d)[:7] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "work_model_frac"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 * max(1.0, r["frac"])
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # a bounded tail of stdout (what the driver keeps) still holds the whole line
    tail = ("x" * 20000 + "\n" + line + "\n")[-8126:]
    assert json.loads(tail.splitlines()[-1]) == d
    return d


def test_headline_from_a_full_run_fits_and_has_the_contract_keys():
    full = _full()
    assert len(json.dumps(full)) > 20000  # the canned document really is the oversized one
    d = _check(json.dumps(bench.headline(full, "gpurun_out/bench_detail.json")))
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-5
    assert abs(d["ms_per_step"] - full["ms_per_step"]) / full["ms_per_step"] < 1e-5
    assert d["detail"] == "gpurun_out/bench_detail.json"
    assert "_note" not in json.dumps(d) and '"reading"' not in json.dumps(d)


def test_headline_of_an_eight_rank_run_fits():
    full = copy.deepcopy(_full())
    full["n_gpus"] = 8
    full.pop("cpu_baseline")
    full.update({"rccl_ranks": 8, "backend": "nccl", "ms_per_step_ranks": [5.123456789] * 8, "ms_per_step_inproc": 6.54321,
                 "replicas": {"n": 8, "constraints_per_s": 4.4e8, "ms_per_proof_each": [19.1] * 8},
                 "shard_plan": [{"rank": r, "commit_msm_range": [0, 1], "k_msm_range": [0, 1], "extends": True, "pairs": 1} for r in range(8)],
                 "stages_ms_per_step_by_rank": [{"rank": r, "x" * 30: 1.0} for r in range(8)]})
    full["config"]["sharding"] = "y" * 500
    line = json.dumps(bench.headline(full, None))
    assert len(line) < bench.LINE_LIMIT
    d = json.loads(line)
    assert d["rccl_ranks"] == 8 and len(d["ms_per_step_ranks"]) == 8 and "shard_plan" not in d


def test_no_prose_keys_are_built_in_bench_py():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '_note"' not in src and '"reading"' not in src and '"note"' not in src
