"""Regenerates the committed golden fixtures.  Run from the repo root: python tests/golden/make_golden.py

  poseidon_rc_known.json  independently known published Poseidon-Goldilocks round constants
                          (plonky2 `ALL_ROUND_CONSTANTS`, re-used by boojum): first 14 and the last.
                          These are DATA typed from public knowledge, not produced by this repo.
  ram_fixture.json        the 3+3 MemoryQuery fixture of the reference's only ram_permutation test
                          (/root/reference/src/ram_permutation/mod.rs:559-634), transcribed as data.
  poseidon2_vectors.json  SELF-REFERENTIAL vectors (produced by oracle/libzko.so): they pin the GPU
                          path and future refactors to today's oracle, not to boojum (parity unpinned).
  ram_commitments.json    oracle-native input commitments for the fixture / seeded random instances.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ram_native as rn  # noqa: E402
from oracle import zko  # noqa: E402

known = {
    "source": "plonky2 poseidon_goldilocks ALL_ROUND_CONSTANTS (public), indices into the 12x30 table",
    "first": ["0xb585f766f2144405", "0x7746a55f43921ad7", "0xb2fb0d31cee799b4", "0x0f6760a4803427d7",
              "0xe10d666650f4e012", "0x8cae14cb07d09bf1", "0xd438539c95f63e9f", "0xef781c7ce35b4c3d",
              "0xcdc4a239b0c44426", "0x277fa208bf337bff", "0xe17653a29da578a1", "0xc54302f225db2c76",
              "0x86287821f722c881", "0x59cd1a8a41c18e55"],
    "last": "0xbc8dfb627fe558fc",
}
json.dump(known, open(os.path.join(HERE, "poseidon_rc_known.json"), "w"), indent=1)

X = 1125899906842626
fixture = {
    "source": "/root/reference/src/ram_permutation/mod.rs:559-634 (test_ram_permutation_inner), limit 16 (:538), "
              "num_nondeterministic_writes starts at 1 (:529-535); BOOTLOADER_HEAP_PAGE is a zkevm_opcode_defs constant [EXT]",
    "fields": ["timestamp", "memory_page", "index", "rw_flag", "is_ptr", "value"],
    "unsorted": [[1025, 30, 0, 0, 0, X], [1024, 30, 0, 1, 0, X], [0, "BOOTLOADER_HEAP_PAGE", 695, 1, 0, 12345678]],
    "sorted": [[0, "BOOTLOADER_HEAP_PAGE", 695, 1, 0, 12345678], [1024, 30, 0, 1, 0, X], [1025, 30, 0, 0, 0, X]],
    "limit": 16,
}
json.dump(fixture, open(os.path.join(HERE, "ram_fixture.json"), "w"), indent=1)

rng = np.random.default_rng(0x7051)
vecs = []
for case in ([0] * 12, list(range(12)), [zko.P - 1] * 12):
    vecs.append({"in": [hex(x) for x in case], "out": [hex(x) for x in zko.poseidon2_permute(case)]})
for _ in range(5):
    s = [int(x) % zko.P for x in rng.integers(0, 2**63, size=12, dtype=np.uint64) * 2 + rng.integers(0, 2, size=12, dtype=np.uint64)]
    vecs.append({"in": [hex(x) for x in s], "out": [hex(x) for x in zko.poseidon2_permute(s)]})
commits = []
for L in (0, 1, 7, 8, 9, 18, 51, 69):
    v = [int(x) % zko.P for x in rng.integers(0, 2**63, size=L, dtype=np.uint64)]
    commits.append({"in": [hex(x) for x in v], "out": [hex(x) for x in zko.commit_encoding(v)]})
json.dump({"note": "self-referential (oracle-generated); parity with boojum unpinned", "permute": vecs, "commit_encoding": commits},
          open(os.path.join(HERE, "poseidon2_vectors.json"), "w"), indent=1)

def items(lst):
    return [rn.mq(a, rn.BOOTLOADER_HEAP_PAGE if b == "BOOTLOADER_HEAP_PAGE" else b, c, d, e, f) for a, b, c, d, e, f in lst]
inst = rn.instance(items(fixture["unsorted"]), items(fixture["sorted"]), 16, 1)
out = {"note": "self-referential: oracle/ram_native.py on the reference fixture and on seeded random witnesses",
       "fixture_limit16": [hex(x) for x in inst["commitment"]], "random": []}
for seed, n, limit in ((1, 5, 8), (2, 8, 8), (3, 12, 16)):
    r = np.random.default_rng(seed)
    u, s, nd = rn.random_ram_witness(r, n)
    i2 = rn.instance(u, s, limit, nd)
    assert i2["satisfiable"] and i2["completed"]
    out["random"].append({"seed": seed, "n": n, "limit": limit, "commitment": [hex(x) for x in i2["commitment"]]})
json.dump(out, open(os.path.join(HERE, "ram_commitments.json"), "w"), indent=1)
print("golden fixtures written")
