"""tests/emu/dev/gen_dev.py — the product's sources, re-written for the EMULATED DEVICE (tests/emu/README.md; TEST INFRASTRUCTURE).

Every file of era-zkevm_circuits_amd/csrc is copied to <out>/src with four mechanical edits (each counted; an unexpected count is an error):
  1. kernel launches      K<<<grid, block, lds, stream>>>(args);   ->  emu::launch(grid, block, lds, stream, [&] { K(args); });
  2. dynamic LDS          extern __shared__ T name[];              ->  T* name = (T*)emu::dyn_lds;
  3. AMD inline assembly  the s_waitcnt of the strand barriers is dropped (the emulated __syncthreads is the whole fence); the hand-scheduled
                          destination walk of the round-1 interpreter (kernels_engine.hpp) is switched off — the generic loop right behind it
                          does the same stores
  4. duplicate lanes      behind every `lane = active ? lane : n - 1;` (the witness / check kernels let the lanes beyond the batch redo the last
                          valid lane's work) a hook tells the emulator that this work-item is a duplicate: the race-detector build
                          (EMU_TSAN) ignores its accesses — it stores and reads the very cells of the lane it copies.  Empty otherwise.
  5. nothing else.
usage: python tests/emu/dev/gen_dev.py <out_dir>"""
import os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CSRC = os.path.join(ROOT, "era-zkevm_circuits_amd", "csrc")


def match_forward(s, i, open_c="(", close_c=")"):
    assert s[i] == open_c, s[i - 20:i + 20]
    depth = 0
    while True:
        c = s[i]
        if c == open_c: depth += 1
        elif c == close_c:
            depth -= 1
            if depth == 0: return i
        i += 1


def rewrite_launches(s, name):
    out, pos, n = [], 0, 0
    while True:
        i = s.find("<<<", pos)
        if i < 0: break
        j = s.find(">>>", i)
        cfg = s[i + 3:j]
        # the kernel expression, backwards: identifier[::identifier] [<template arguments>]
        k = i
        while s[k - 1].isspace(): k -= 1
        if s[k - 1] == ">":
            depth, k = 0, k - 1
            while True:
                if s[k] == ">": depth += 1
                elif s[k] == "<":
                    depth -= 1
                    if depth == 0: break
                k -= 1
        while s[k - 1].isalnum() or s[k - 1] in "_:": k -= 1
        kern = s[k:i].strip()
        a0 = j + 3
        while s[a0].isspace(): a0 += 1
        a1 = match_forward(s, a0)
        semi = a1 + 1
        while s[semi].isspace(): semi += 1
        if s[semi] != ";": raise SystemExit(f"gen_dev.py: {name}: a launch that is not a statement near {s[i - 40:i + 40]!r}")
        out.append(s[pos:k])
        out.append(f"emu::launch({cfg}, [&] {{ {kern}{s[a0:a1 + 1]}; }});")
        pos = semi + 1
        n += 1
    out.append(s[pos:])
    return "".join(out), n


def main(out_dir):
    src_out = os.path.join(out_dir, "src")
    counts = {"launches": 0, "dyn_lds": 0, "waitcnt": 0, "walk": 0, "duplicates": 0, "mirrored": 0}
    for dirpath, _, files in os.walk(CSRC):
        rel = os.path.relpath(dirpath, CSRC)
        os.makedirs(os.path.join(src_out, rel), exist_ok=True)
        for f in files:
            if not f.endswith((".hpp", ".cpp", ".hip", ".h")): continue
            s = open(os.path.join(dirpath, f)).read()
            name = os.path.join(rel, f)
            if "<<<" in s:
                s, n = rewrite_launches(s, name); counts["launches"] += n
            s, n = re.subn(r"extern\s+__shared__\s+([\w:]+)\s+(\w+)\s*\[\s*\]\s*;", r"\1* \2 = (\1*)emu::dyn_lds;", s); counts["dyn_lds"] += n
            s = re.sub(r"__shared__\s+(alignas\(\d+\))", r"\1 __shared__", s)   # (C++ wants the alignment before the storage class the shim maps __shared__ to)
            s, n = re.subn(r"((?:lane|inst) = active \? (?:lane|inst) : [^;]+;)", r"\1 EMU_DUPLICATE_LANE(!active);", s); counts["duplicates"] += n
            # (the recorded-cone seeders fold the lanes of a wavefront onto lpb instances and clamp the surplus to the last instance — "identical values, benign
            #  duplicate stores", kernels_engine2.hpp — by design: the race-detector build does not look at them)
            s, n = re.subn(r"(const uint32_t inst = min\(blockIdx\.x \* lpb \+ l, n_instances - 1\);)", r"\1 EMU_DUPLICATE_LANE(true);", s); counts["mirrored"] += n
            w = 'asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");'
            counts["waitcnt"] += s.count(w); s = s.replace(w, "/* (s_waitcnt: device only) */")
            if f == "kernels_engine.hpp":
                a = s.find("asm volatile(\n")
                if a < 0: raise SystemExit("gen_dev.py: the destination walk of kernels_engine.hpp was not found")
                b = match_forward(s, s.find("(", a))
                while s[b] != ";": b += 1
                s = s[:a] + "wlast = 0; addr = 0; (void)addr; /* (hand-scheduled walk: device only) */" + s[b + 1:]
                if s.count("if (off < 64) {") != 1: raise SystemExit("gen_dev.py: expected one `if (off < 64) {` in kernels_engine.hpp")
                s = s.replace("if (off < 64) {", "if (false && off < 64) {"); counts["walk"] += 1
            if os.environ.get("EMU_SABOTAGE") == "fsm_lockstep" and f == "kernels_fsm_seed.hpp":   # self-test of the race detector: take out the fence this harness asked for
                k = s.find("wave_sync();   // every lane has read the state of cycle c")
                if k < 0: raise SystemExit("gen_dev.py: the walker's wave_sync was not found")
                s = s[:k] + "/* sabotaged */" + s[k + len("wave_sync();"):]
            s = s.replace('#include "../../include/', f'#include "{os.path.join(ROOT, "include")}/')
            s = s.replace('#include "../../../include/', f'#include "{os.path.join(ROOT, "include")}/')
            p = os.path.join(src_out, rel, f if not f.endswith(".hip") else f[:-4] + ".cpp")
            new = f"// GENERATED by tests/emu/dev/gen_dev.py from era-zkevm_circuits_amd/csrc/{name} — do not edit\n" + s
            if not os.path.exists(p) or open(p).read() != new: open(p, "w").write(new)
    if counts["duplicates"] != 5 or counts["mirrored"] != 3 or counts["walk"] != 1 or counts["waitcnt"] < 2 or counts["launches"] < 70:
        raise SystemExit(f"gen_dev.py: unexpected edit counts {counts}")
    print("gen_dev.py:", counts)


if __name__ == "__main__":
    main(sys.argv[1])
