"""bench.py --gpus N outside torchrun starts N ranks itself and can not print a line whose n_gpus differs from --gpus
(a plain `python bench.py --gpus 8` used to measure ONE GPU and say so only in `n_gpus`)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=300)


def test_launcher_argv_is_the_drivers_command_shape():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "7", "--warmup", "2"], 4, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


def test_more_ranks_than_gpus_fails_loudly():
    """this container has no GPU: --gpus 2 must exit non-zero with a message and print no JSON line"""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "{" not in r.stdout


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and "{" not in r.stdout
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "{" not in r.stdout


def test_no_fraction_of_a_peak_above_one_and_per_key_pricing():
    """the per-key in-projection launch was priced with the ROW count (frac 2.2 of the matrix peak).  The pricing
    table takes the distinct-key count for such an engine, and the line lists every `*frac*` entry above 1 (none is acceptable)."""
    sys.path.insert(0, ROOT)
    import bench

    class Eng:
        qkv_dedup, dropcorr = False, True
    assert bench.per_key_of(Eng) and not bench.per_key_of(object())
    rows, keys, D = 30800.0, 4600.0, 256
    assert bench.nrms_flops(rows, D, 300, per_key=keys)["qkv_fwd_item"] == 2.0 * keys * D * 3 * D
    assert bench.nrms_flops(rows, D, 300)["qkv_fwd_item"] == 2.0 * rows * D * 3 * D
    line = {"roofline": {"frac": 0.86}, "kernels": {"a": {"frac_of_f32_mfma_peak": 2.19, "in_region": {"frac_of_f32_mfma_peak": 0.4}}},
            "secondary": [{"x": {"frac_of_hbm_peak": 1.01}}]}
    assert bench.fracs_over_one(line) == ["kernels.a.frac_of_f32_mfma_peak=2.19", "secondary[0].x.frac_of_hbm_peak=1.01"]
    assert bench.fracs_over_one({"conv3_fwd": {"frac_of_f32_mfma_peak": 1.016, "mfma_issue_frac": 0.677}}) == []      # Winograd: direct-conv pricing
    assert bench.fracs_over_one({"conv3_fwd": {"frac_of_f32_mfma_peak": 1.6, "mfma_issue_frac": 1.07}}) != []
    st = bench.step_roofline("nrms", {}, 0.98e-3, {"rows": rows, "uniq": keys, "glove": True, "per_key": True}, D, 300, ())
    assert 0.0 < st["issued_frac"] < st["frac"] < 1.0


import pytest  # noqa: E402


@pytest.mark.gpu
def test_the_drivers_command_prints_one_line_with_the_contract_fields():
    """`python bench.py --gpus 1 --steps K --warmup W` (the driver's command shape) on the small world: exactly one JSON line on
    stdout, the contract's fields with the metric's names, the live roofline object of the dominant kernel, f32"""
    import json
    r = _run(["--gpus", "1", "--steps", "6", "--warmup", "2", "--small", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["metric"] == "train impressions/sec on MIND-small NAML" and j["unit"] == "impressions/s"
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 2 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and j["vs_baseline"] is None and "workload" in j["config"]
    assert j["value"] > 0 and abs(j["value"] - 64 * 1e3 / j["ms_per_step"]) < 0.01 * j["value"]
    roof = j["roofline"]
    assert roof["bound"] in ("mfma", "hbm") and roof["unit"] in ("TFLOP/s", "GB/s") and roof["peak"] > 0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and "traffic" in roof
    assert "cpu_baseline" in j and j["cpu_baseline"] is None          # skipped by the flag; the key stays
    assert j["fracs_over_one"] == [], j["fracs_over_one"]
    assert j["value_without_prewarm"] > 0 and j["without_prewarm"]["steps"] == 6
