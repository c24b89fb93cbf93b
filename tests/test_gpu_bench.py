"""GPU test of bench.py's contract: one JSON line with the fields the driver parses, launched the way the driver launches the
N > 1 runs (python -m torch.distributed.run ... bench.py --gpus N).  A single-GPU box cannot host two ranks, so the RCCL process
group is forced on with ONE rank (--force-dist): initialisation over 127.0.0.1, the barriers on both sides of the timed region
and the MAX all-reduce all run through RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, **extra_env):
    env = {k: v for k, v in os.environ.items() if not k.startswith("OZIMMU_")}
    env.update(extra_env)
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_under_torch_distributed_run_with_the_rccl_group():
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "1", "--steps", "3", "--warmup", "1",
                "--force-dist", "--no-cpu", "--no-traffic", "--no-configs"], OZIMMU_BENCH_N="2048")
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1
    assert out["unit"] == "TFLOP/s" and out["higher_is_better"] is True and out["scaling"] == "weak"
    # no published number: vs_baseline is null; the ratios against rocBLAS native DGEMM have their own keys
    assert out["vs_baseline"] is None
    assert out["vs_rocblas_dgemm_alternating"] == out["extra"]["interleaved_vs_rocblas_dgemm"]["ratio"]
    assert out["vs_rocblas_dgemm_sequential"] == out["extra"]["speedup_vs_rocblas_dgemm"]
    assert abs(out["value"] - 2.0 * 2048 ** 3 / (out["ms_per_step"] * 1e-3) / 1e12) < 0.02 * out["value"]
    assert 20.0 < out["value"] < 200.0                      # fp64_int8_9 at 2048^3: ~54 TFLOP/s
    assert out["roofline"]["bound"] == "mfma" and 0.1 < out["roofline"]["frac"] < 1.0
    assert out["extra"]["relative_residual"] < 1e-15


def test_bench_default_line_has_the_contract_fields():
    out = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-configs"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["config"]["workload"].startswith("fp64_int8_9, M=8192 N=8192 K=8192")
    rf = out["roofline"]
    assert rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None or rf["traffic"] > rf["traffic_algorithmic_bytes"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert out["extra"]["interleaved_vs_rocblas_dgemm"]["ratio"] > 0.5
    assert out["vs_baseline"] is None and out["vs_rocblas_dgemm_alternating"] == out["extra"]["interleaved_vs_rocblas_dgemm"]["ratio"]
    assert 0.5 < rf["frac_of_measured_mfma_ceiling"] < 1.05 and rf["measured_mfma_ceiling"] < rf["peak"]
    assert set(out["extra"]["short_k"]) == {"8192x8192x256", "8192x8192x512"}
    assert "1536" in out["extra"]["square_sizes_tflops"]
