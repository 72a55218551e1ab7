"""bench.py end to end on a small configuration (2^16 rows, a handful of instances): the one JSON line, its objects, the host-fed figures —
and the N > 1 launch (torch.distributed.run, one rank per GPU, communicator set-up, per-step gather) BEFORE the driver's scaling run meets
it: on a box with two GPUs the gather is the RCCL one behind the C ABI, on a one-GPU box the two ranks share the device and the line says so."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(*args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_one_gpu_line_carries_roofline_and_host_fed_figures(zk):
    d = run_bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "6", "--log2-rows", "16", "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["unit"] == "constraints/s"
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["kernel"] == "zke::k_witness_loop" and 0 < r["frac"] < 1 and r["avg_launch_ms"] > 0
    assert d["config"]["commitment_gather"].startswith("zk_cs_gather_commitments")
    for key in ("value_including_host_pack", "value_states_from_witness"):
        h = d[key]
        assert h is not None and "error" not in h, h
        assert h["value"] > 0 and h["pack_ms_per_instance_one_core"] > 0 and h["h2d_GBps"] > 0 and h["host_cores_per_gpu_to_sustain_value"] > 0
    # device_seeds stages 117 of the 360 rows, states_from_witness all of them
    assert d["value_states_from_witness"]["staged_bytes_per_window"] > 2.5 * d["value_including_host_pack"]["staged_bytes_per_window"]
    assert d["distinct_commitments"] == 6


def test_two_rank_launch_preflight(zk):
    import zkgl
    d = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--log2-rows", "16", "--no-cpu-baseline", "--headline-only")
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["config"]["per_rank_ms_per_step"]) == 2
    if zkgl.device_count() >= 2:
        assert d["config"]["commitment_gather"].startswith("zk_cs_gather_commitments"), d["config"]["commitment_gather"]
    else:
        assert "ranks share one GPU" in d["config"]["commitment_gather"]
    assert d["distinct_commitments"] == 8
