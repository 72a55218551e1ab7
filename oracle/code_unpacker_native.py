"""oracle/code_unpacker_native.py — CPU ORACLE (test infrastructure): native restatement of
unpack_code_into_memory_entry_point / _inner (/root/reference/src/code_unpacker_sha256/mod.rs:33-442).

[EXT] ContractCodeSha256::VERSION_BYTE = 0x01 (zkevm_opcode_defs); pinned by the reference fixture."""
from __future__ import annotations

import hashlib

from . import zko
from .decommit_native import ZERO, dq, encode
from .ram_native import mq

IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
OUTER_WORDS, LOOP_WORDS, CARRIED = 125, 101, 74


def versioned_hash(code_words):
    """0x01 00 | length in words (u16 BE) | sha256(code)[4..32]"""
    code = b"".join(w.to_bytes(32, "big") for w in code_words)
    return int.from_bytes(b"\x01\x00" + len(code_words).to_bytes(2, "big") + hashlib.sha256(code).digest()[4:], "big")


def empty_fsm():
    return dict(state=[0] * 8, hash=[0] * 8, index=0, page=0, timestamp=0, rounds_left=0, length_in_bits=0, get=0, decommit=0, finished=0,
                req=[0] * 25, mem=[0] * 25)


def flatten_fsm(f):
    return list(f["state"]) + list(f["hash"]) + [f["index"], f["page"], f["timestamp"], f["rounds_left"], f["length_in_bits"], f["get"],
                                                 f["decommit"], f["finished"]] + list(f["req"]) + list(f["mem"])


def limbs(x):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def instance(requests, limit, start_flag=True, fsm_in=None, obs=None, pending=None):
    """`requests`: list of (DecommitQuery 11 words, code words) still in the queue; `pending`: words left of the code in flight"""
    if start_flag:
        tail = [0] * 12
        for q, _ in requests:
            tail = zko.queue_full_push(tail, encode(q))
        obs_req = [0] * 12 + tail + [len(requests)]
        obs_mem = [0] * 25
        fsm_in = empty_fsm()
        f = dict(empty_fsm(), get=1, req=list(obs_req), mem=list(obs_mem))
    else:
        obs_req, obs_mem = obs
        f = {k: list(v) if isinstance(v, list) else v for k, v in fsm_in.items()}
    outer = [int(start_flag)] + list(obs_req) + list(obs_mem) + flatten_fsm(fsm_in)
    assert len(outer) == OUTER_WORDS
    requests, pending = list(requests), list(pending or [])
    state, hash_cmp = list(f["state"]), list(f["hash"])
    index, page, timestamp, rounds_left, length_in_bits = f["index"], f["page"], f["timestamp"], f["rounds_left"], f["length_in_bits"]
    get, decommit, finished = f["get"], f["decommit"], f["finished"]
    req_head, req_tail, req_len = f["req"][0:12], f["req"][12:24], f["req"][24]
    mem_head, mem_tail, mem_len = f["mem"][0:12], f["mem"][12:24], f["mem"][24]
    ok, rows, pushed = True, [], []

    def push(q):
        nonlocal mem_tail, mem_len
        pushed.append(q)
        mem_tail = zko.queue_full_push(mem_tail, zko.memory_query_encode(q))
        mem_len += 1

    for _ in range(limit):
        carried = [(w >> (8 * k)) & 0xFF for w in state for k in range(4)] + hash_cmp + [index, page, timestamp, rounds_left, length_in_bits,
                                                                                           get, decommit, finished] + req_head + [req_len] + mem_tail + [mem_len]
        assert len(carried) == CARRIED
        req = list(ZERO)
        if get:
            if req_len == 0:
                ok = False
            else:
                req, pending = requests.pop(0)
                pending = list(pending)
                req_head = zko.queue_full_push(req_head, encode(req))
                req_len -= 1
            top = req[7]
            if (top >> 16) != 0x0100:
                ok = False
            length_in_words = top & 0xFFFF
            if length_in_words % 2 == 0:
                ok = False   # (length + 1) / 2 is not a u16
            rounds_left = (length_in_words + 1) // 2
            length_in_bits = length_in_words * 256
            timestamp, page = req[10], req[8]
            hash_cmp = req[0:7] + [0]
            index = 0
            state = list(IV)
        decommit = 1 if (decommit or get) else 0
        get = 0
        if decommit:
            rounds_left -= 1
        last_round = rounds_left == 0
        finalize = last_round and decommit
        second = (not last_round) and decommit
        w0 = pending.pop(0) if (decommit and pending) else 0
        w1 = pending.pop(0) if (second and pending) else 0
        if decommit:
            push(mq(timestamp, page, index, 1, 0, w0))
            index += 1
        if second:
            push(mq(timestamp, page, index, 1, 0, w1))
            index += 1
        block = bytearray(w0.to_bytes(32, "big") + w1.to_bytes(32, "big"))
        if finalize:
            block[32:] = b"\x80" + bytes(27) + length_in_bits.to_bytes(4, "big")
        new_state = zko.sha256_compress(state, bytes(block))
        if decommit:
            state = new_state
        if finalize and [new_state[7 - i] for i in range(7)] + [0] != hash_cmp:
            ok = False
        is_empty = req_len == 0
        finished = 1 if (finished or (is_empty and finalize)) else 0
        get = 1 if ((not is_empty) and finalize) else 0
        decommit = 1 if second else 0
        rows.append(carried + list(req) + limbs(w0) + limbs(w1))
    if req_len == 0 and req_head != req_tail:
        ok = False
    fsm_out = dict(state=state, hash=hash_cmp, index=index, page=page, timestamp=timestamp, rounds_left=rounds_left, length_in_bits=length_in_bits,
                   get=get, decommit=decommit, finished=finished, req=req_head + req_tail + [req_len], mem=mem_head + mem_tail + [mem_len])
    done = finished
    z4 = [0] * 4
    # observable input in CodeDecommitterInputData's field order (input.rs:80-83): memory queue state, then the requests queue state
    compact = [int(start_flag), done] + zko.commit_encoding(list(obs_mem) + list(obs_req)) + \
        (zko.commit_encoding(fsm_out["mem"]) if done else z4) + \
        (z4 if start_flag else zko.commit_encoding(flatten_fsm(fsm_in))) + \
        (z4 if done else zko.commit_encoding(flatten_fsm(fsm_out)))
    return dict(outer=outer, rows=rows, fsm_out=fsm_out, satisfiable=ok, pushed=pushed, rest=(requests, pending), obs=(obs_req, obs_mem),
                public_input=zko.commit_encoding(compact), memory_state=fsm_out["mem"])
