"""The -m gpu parity tests, run HERE on the emulated device (tests/emu/README.md, tests/emu/dev/).

The product's device source — every kernel of zkgl_device.hip, with the launchers and the host side around them — is compiled as host C++ over a
stand-in <hip/hip_runtime.h>; work-items are fibers, 64 consecutive ones a wavefront; ballots, readfirstlane / readlane, shuffles, DPP moves, the
wave barrier and __syncthreads are rendezvous points.  The library is loaded through ZKGL_LIB exactly like a variant build, so the parity tests
run unchanged: same C ABI, same oracle comparisons.  This is TEST INFRASTRUCTURE: it says nothing about time, coalescing or registers, and
the product has no path to it (tests/test_abi.py: no GPU -> loud failure).

A subset runs here on every CPU run (a few minutes of CPU in parallel workers); `tools/emulated_gpu_suite.sh` runs the whole -m gpu suite (minus
the 2^20-row tests) — profiles/r5_emulated_device.md holds the round's record."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = os.path.join(ROOT, "tests", "emu", "dev")

SUBSET = [
    "tests/test_gpu_primitives.py",                                              # K1..K4 + the encodings, the grand-product scans (shuffles)
    "tests/test_gpu_cs.py::test_ram_fixture_trace_bit_exact",                     # the reference's ram_permutation fixture
    "tests/test_gpu_cs.py::test_ram_batch_of_instances",
    "tests/test_gpu_cs.py::test_ram_unsatisfied_witnesses_are_rejected_like_the_oracle",
    "tests/test_gpu_cs.py::test_copy_constraint_failures_name_the_pair",
    "tests/test_gpu_cs.py::test_storage_validity_gpu_equals_oracle",
    "tests/test_gpu_cs.py::test_demux_log_queue_gpu",
    "tests/test_gpu_cs.py::test_sort_decommittment_requests_gpu",
    "tests/test_gpu_cs.py::test_sha256_round_function_fsm_gpu",                   # macro-op SHA256_ROUNDS + the two-wavefront FSM seeder
    "tests/test_queue_seed.py",                                                   # scan seeders: shuffles, DPP rows, ballots
    "tests/test_ntt.py", "tests/test_copy_permutation.py",                        # K11 (LDS passes up to 2^22), K12 (scans)
    "tests/test_gpu_main_vm.py::test_native_seeding_with_several_walkers_per_wavefront",   # k_vm_walk<2> / <4>: the forms a full batch's seeding pass takes
    "tests/test_gpu_main_vm.py::test_main_vm_gpu_bit_exact",                      # k_witness_loop on whole wavefronts: flag planes, gated permutations, wave-aggregated multiplicities
    # round 6: the loop scope over the NARROW store (k_witness_loop_narrow, k_check_prog_t<true>, links over address words, k_widen_last / k_widen_store, columns read from
    # one-byte slots) on a queue circuit; the main_vm cases of that file and the other full-size configurations run in tools/emulated_gpu_suite.sh
    "tests/test_zz_round6_narrow_store.py::test_ram_permutation_over_the_narrow_store",
    "tests/test_gpu_full_size.py::test_c1_ram_permutation_2_16_rows",             # BASELINE's C1 at full size
    # the macro-op backends with kernels of their own (XMACROS): the reference's 4-bit-chunk SHA compression, default recording of its table set
    "tests/test_zz_round5_gpu.py::test_gpu_equals_oracle_with_the_reference_tables",
    # zk_comm_create / zk_cs_gather_commitments with a world of TWO processes (the stand-in collective of tests/emu/dev/rccl/rccl.h: comm.cpp's ranks, counts, buffers — not RCCL)
    "tests/test_multi.py::test_rccl_gather_behind_the_c_abi_on_two_gpus",
]


def build(variant="", defs=()):
    env = dict(os.environ)
    if variant:
        env["EMU_VARIANT"] = variant
    r = subprocess.run(["bash", os.path.join(DEV, "build.sh"), *defs], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return os.path.join(ROOT, "tests", "emu", "_gen", "dev" + ("_" + variant if variant else ""), "libzkgl.so")


def run_gpu_tests(lib, nodes, jobs=6, timeout=1500):
    env = dict(os.environ, ZKGL_LIB=lib)
    env.pop("PYTEST_CURRENT_TEST", None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-n", str(jobs), "-p", "no:cacheprovider", *nodes], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    tail = r.stdout[-4000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout and " error" not in r.stdout, tail
    return r.stdout


def test_generator_rewrites_every_launch_and_nothing_else(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(DEV, "gen_dev.py"), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = open(os.path.join(ROOT, "era-zkevm_circuits_amd", "csrc", "zkgl_device.hip")).read()
    gen = open(os.path.join(tmp_path, "src", "zkgl_device.cpp")).read()
    assert "<<<" not in gen and gen.count("emu::launch(") == src.count("<<<") >= 70
    # apart from the launches the device translation unit is the product's, line for line
    a = [l for l in src.splitlines() if "<<<" not in l and ">>>" not in l]
    b = [l for l in gen.splitlines()[1:] if "emu::launch(" not in l]
    assert len(set(a) - set(b)) <= 8, sorted(set(a) - set(b))[:10]   # (continuation lines of multi-line launches)


def test_gpu_parity_subset_on_the_emulated_device():
    out = run_gpu_tests(build(), SUBSET)
    n = int(out.strip().splitlines()[-1].split(" passed")[0].split()[-1])
    assert n >= 83, out[-500:]


def test_bench_py_end_to_end_on_the_emulated_device():
    """bench.py — the driver's instrument — with EVERY leg (stored mode, deferred intermediates, resident inputs, witness columns, the realistic fixture, both
    host-fed forms, the narrow store's labelled mode) on a tiny configuration: one JSON line, no leg reports an error, the contract's keys are there, the legs
    that compare commitments agree.  Several of these legs were written while no GPU call was accepted; until this runner nothing had executed them
    (tests/emu/bench_on_emulator.py: torch.cuda patched from outside, host tensors as device tensors; the times mean nothing)."""
    import json
    lib = build()
    env = dict(os.environ, ZKGL_LIB=lib)
    env.pop("PYTEST_CURRENT_TEST", None)
    base = [sys.executable, os.path.join(ROOT, "tests", "emu", "bench_on_emulator.py"), "--batch", "2", "--log2-rows", "15", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    for extra, kernel in ((["--with-narrow-store-mode"], "zke::k_witness_loop"), (["--narrow-store", "--headline-only"], "zke::k_witness_loop_narrow")):
        r = subprocess.run(base + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in d, key
        assert d["unit"] == "constraints/s" and d["value"] > 0 and d["n_gpus"] == 1 and d["steps"] == 2 and d["vs_baseline"] is None
        rf = d["roofline"]
        assert rf["bound"] == "hbm" and rf["kernel"] == kernel and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and rf["achieved"] > 0
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
        assert "workload" in d["config"] and "model" not in d["config"]
        assert [k for k, v in d.items() if isinstance(v, dict) and "error" in v] == []
        if "--headline-only" in extra:
            continue
        for key in ("value_realistic_fixture", "value_including_host_pack", "value_states_from_witness", "mode_p2_intermediates_deferred", "mode_narrow_store"):
            assert isinstance(d[key], dict) and d[key], key
        assert d["mode_p2_intermediates_deferred"]["commitments_equal_native_restatement"] is True
        for label in ("plain", "p2_deferred"):
            leg = d["mode_narrow_store"][label]
            assert leg["commitments_equal_native_restatement"] is True and leg["steps_repeated_over_the_ordinary_store"] == 0
        assert 0.70 < d["mode_narrow_store"]["bytes_ratio"] < 0.80
        assert d["witness_rows_materialised_per_s"] > 0
    # the N > 1 launch in the driver's own form (one process per rank through torch.distributed.run, rendezvous on 127.0.0.1, max over ranks, rank 0 prints the line);
    # the ranks get the torch.cuda stand-ins through tests/emu/site/sitecustomize.py.  EMU_DEVICES=2: rank r binds "device" r and the step's gather is the product's
    # (zk_cs_gather_commitments on a world-2 communicator made from rank 0's unique id; the collective underneath is the stand-in of tests/emu/dev/rccl/rccl.h).
    env2 = dict(env, ZKGL_EMU_TORCH="1", EMU_DEVICES="2", PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "emu", "site")] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else [])))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--log2-rows", "15", "--no-cpu-baseline", "--headline-only"],
                       cwd=ROOT, env=env2, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and len(d["config"]["per_rank_ms_per_step"]) == 2
    assert d["ms_per_step"] >= max(d["config"]["per_rank_ms_per_step"]) - 2e-3          # the MAX over ranks (the per-rank figures are rounded to 1 us)
    assert d["distinct_commitments"] == 4 and d["config"]["commitment_gather"].startswith("zk_cs_gather_commitments per step")


def test_product_library_is_not_the_emulated_one():
    import zkgl
    assert not __import__("helpers").emulated_device()


SANITIZED = ["tests/test_gpu_primitives.py", "tests/test_gpu_cs.py::test_ram_fixture_trace_bit_exact", "tests/test_gpu_cs.py::test_storage_validity_gpu_equals_oracle",
             "tests/test_queue_seed.py", "tests/test_copy_permutation.py"]


@pytest.mark.skipif(not os.environ.get("ZKGL_EMU_SANITIZERS"), reason="two more builds of the emulated device (~2.5 min): set ZKGL_EMU_SANITIZERS=1; the round's full record is profiles/r5_emulated_device.md")
@pytest.mark.parametrize("tool", ["emulated_race_check.sh", "emulated_bounds_check.sh"])
def test_kernels_under_the_race_detector_and_the_bounds_check(tool, tmp_path):
    """tools/emulated_race_check.sh (ThreadSanitizer fibers, the device's happens-before edges) / tools/emulated_bounds_check.sh (AddressSanitizer): no report"""
    env = dict(os.environ, OUT=str(tmp_path), JOBS="6")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", tool), *SANITIZED], cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0 and " passed" in r.stdout and " failed" not in r.stdout, r.stdout[-3000:]
    assert "reported: 0" in r.stdout, r.stdout[-3000:]
