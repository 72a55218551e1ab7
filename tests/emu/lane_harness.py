"""tests/emu/lane_harness.py — Python face of the LANE HARNESS (tests/emu/README.md): the product's witness interpreter source compiled for the host,
run one lane at a time on a recorded circuit's device programs.  Test infrastructure; the product has no path to it."""
import ctypes as C
import os
import subprocess

import numpy as np

import zkgl

HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def lib(variant: str = "", defs=()):
    """build (once per process) and load libzkgl_emu[_variant].so; `variant` names a side-by-side library loaded through ZKGL_LIB"""
    key = variant
    if key not in _libs:
        if not os.environ.get("EMU_NO_BUILD"):
            subprocess.run(["bash", os.path.join(HERE, "build.sh"), variant, *defs], check=True, capture_output=True)
        zkgl.lib()   # libzkgl (or the variant ZKGL_LIB names) first: the harness links against it
        _libs[key] = C.CDLL(os.path.join(HERE, "_gen", f"libzkgl_emu{'_' + variant if variant else ''}.so"))
    return _libs[key]


class Result:
    pass


def resolve(cs, outer: np.ndarray, loop: np.ndarray, batch: int, strands: bool = False, variant: str = "", defs=()) -> Result:
    """resolve `batch` instances on the lane harness -> traces shaped like ConstraintSystem.trace(), public inputs, fused failure words, multiplicities"""
    L = lib(variant, defs)
    sz = (C.c_uint64 * 6)()
    L.zk_emu_sizes(cs._h, C.c_uint32(batch), sz)
    oc = np.zeros((sz[0], sz[1]), dtype=np.uint64); lc = np.zeros((sz[2], max(sz[3], 1) if sz[2] else 0), dtype=np.uint64)
    pub = np.zeros((batch, sz[4]), dtype=np.uint64)
    fail = (C.c_ulonglong * 16)()
    mult = np.zeros((batch, max(sz[5], 1)), dtype=np.uint32)
    outer = np.ascontiguousarray(outer, dtype=np.uint64); loop = np.ascontiguousarray(loop, dtype=np.uint64)
    rc = L.zk_emu_resolve(cs._h, outer.ctypes.data_as(C.c_void_p), loop.ctypes.data_as(C.c_void_p), C.c_uint32(batch), C.c_int(int(strands)),
                          oc.ctypes.data_as(C.c_void_p), lc.ctypes.data_as(C.c_void_p), pub.ctypes.data_as(C.c_void_p), fail, mult.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(zkgl.lib().zk_last_error().decode())
    r = Result()
    r.oc, r.lc, r.public, r.mult = oc, lc, pub, mult
    r.fail = [int(x) for x in fail]
    r.fused_failure = any(x != 0xFFFFFFFFFFFFFFFF for x in r.fail[:6])
    return r


def check(cs, stored: bool, variant: str = ""):
    """the step's checkers (the product's check-kernel source, lane by lane) on the stored values of the LAST resolve(): mode fused (what the fused
    step leaves to the store, merged with the witness kernels' own failure flags) or stored (every relation) -> (accepted, failing lane or None)"""
    L = lib(variant)
    fail = (C.c_ulonglong * 6)()
    rc = L.zk_emu_check(cs._h, C.c_int(1 if stored else 0), fail)
    if rc < 0:
        raise RuntimeError(zkgl.lib().zk_last_error().decode())
    keys = [int(x) for x in fail if int(x) != 0xFFFFFFFFFFFFFFFF]
    return rc == 0, (min(keys) >> 32) if keys else None
