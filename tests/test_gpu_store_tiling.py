"""-m gpu: the lane tiling of the loop scope's variable store (csrc/store_geom.hpp) is a layout choice, not a semantic one.
The default is one 64-lane tile per wavefront; ZKGL_STORE_TILE_LOG2 = 7..12 widens the loop scope's tiles (an A/B switch: the bare
store pattern likes tiles of 4 096 lanes, the real kernel does not care, profiles/r3_loop_probe.md).  Every parity property the other GPU tests establish at 64-lane tiles must hold for a forced
tiling too: whole trace bit-exact against the oracle interpreter, multiplicities, commitments, fault injection naming the place,
seeding, stream windows, fused and stored verdicts, the hash-circuit (strand / multiplicity-pass) kernels, trace columns.
Lane counts here are NOT multiples of the tile (592, 2 048, 70 lanes): partial tiles are the common case."""
import numpy as np
import pytest

import test_fused_check as tfc
import test_gpu_cs as tgc
import test_gpu_main_vm as tvm

pytestmark = pytest.mark.gpu

TILINGS = [7, 9, 12]


@pytest.fixture(scope="module")
def batch():
    d, D = tvm.vp.defs()
    cs = tvm.vp.vm_cs(tvm.LIMIT)
    outer, loop, commits, info = tvm.vp.mixed_batch(cs, D, tvm.LIMIT, 64)
    return cs, D, outer, loop, commits, info


@pytest.mark.parametrize("tile_log2", TILINGS)
def test_ram_permutation_under_a_forced_tiling(zk, monkeypatch, tile_log2):
    monkeypatch.setenv("ZKGL_STORE_TILE_LOG2", str(tile_log2))
    tgc.test_ram_fixture_trace_bit_exact(zk)
    tgc.test_ram_batch_of_instances(zk)
    tgc.test_ram_unsatisfied_witnesses_are_rejected_like_the_oracle(zk)
    tgc.test_fault_injection_reports_place(zk)
    tgc.test_copy_constraint_failures_name_the_pair(zk)
    tgc.test_all_ops_circuit_gpu_equals_oracle(zk)


@pytest.mark.parametrize("tile_log2", TILINGS)
def test_main_vm_under_a_forced_tiling(zk, batch, monkeypatch, tile_log2):
    monkeypatch.setenv("ZKGL_STORE_TILE_LOG2", str(tile_log2))
    tvm.test_main_vm_gpu_bit_exact(zk, batch)
    tvm.test_main_vm_gpu_reports_tampered_witness(zk, batch)
    tvm.test_main_vm_gpu_stream_seeding_and_windows(zk, batch)
    tvm.test_main_vm_hook_compare_witness(zk)


@pytest.mark.parametrize("tile_log2", [12])
def test_checkers_and_hash_circuits_under_a_forced_tiling(zk, monkeypatch, tile_log2):
    monkeypatch.setenv("ZKGL_STORE_TILE_LOG2", str(tile_log2))
    for verify_stored in (False, True):
        tfc.test_fused_and_stored_verdicts_agree(zk, monkeypatch, verify_stored)
    tgc.test_storage_validity_gpu_equals_oracle(zk)
    tgc.test_sha256_round_function_fsm_gpu(zk)
    tgc.test_keccak256_round_function_fsm_gpu(zk)
    tgc.test_eip4844_gpu(zk)


def test_forced_tiling_with_full_and_partial_tiles(zk, monkeypatch):
    """4 800 lanes under tiles of 4 096: one full tile + a partial one; the default stays 64-lane tiles; same trace and commitments"""
    from helpers import oracle_run, ram_cs, random_instances
    from oracle import ram_native as rn

    limit, n = 16, 300
    cs = ram_cs(limit)
    insts = random_instances(23, n, 9, limit)
    outer, loop = rn.pack_streams(insts, limit)
    monkeypatch.delenv("ZKGL_STORE_TILE_LOG2", raising=False)
    cs.set_batch(n)
    assert cs.stats()["loop_store_tile_lanes"] == 64
    monkeypatch.setenv("ZKGL_STORE_TILE_LOG2", "12")
    keep = tgc.gpu_run(zk, cs, outer, loop, n)
    assert cs.stats()["loop_store_tile_lanes"] == 4096
    run = oracle_run(cs, outer, loop, n)
    tgc.assert_trace_equal(cs, run)
    ok, f = cs.check_if_satisfied()
    assert ok, f
    for i in (0, 255, 256, n - 1):
        assert cs.public_inputs(i) == insts[i]["commitment"]
    del keep
