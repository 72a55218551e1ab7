"""oracle/linear_hasher_native.py — CPU ORACLE (test infrastructure): native restatement of linear_hasher_entry_point
(/root/reference/src/linear_hasher/mod.rs:35-212) and LogQuery::into_bytes (src/base_structures/log_query/mod.rs:645-686)."""
from __future__ import annotations

from . import zko
from .storage_native import ZERO_QUERY, encode, fields

RATE, MSG, PERIOD = 136, 88, 17
OUTER_WORDS, LOOP_WORDS, CARRIED = 10, 818, 206


def into_bytes(q) -> bytes:
    f = fields(q)
    be = lambda limbs: b"".join(l.to_bytes(4, "big") for l in reversed(limbs))
    return bytes([f["shard"], f["is_service"]]) + f["tx"].to_bytes(4, "big")[2:] + be(f["address"]) + be(f["key"]) + be(f["written"])


def absorb(state, block):
    s = list(state)
    for j in range(RATE):
        s[j // 8] ^= block[j] << (8 * (j % 8))
    return zko.keccak_f1600(s)


def instance(queries, limit):
    assert limit % PERIOD == 0
    tail = [0] * 4
    for q in queries:
        tail = zko.queue_tail4_push20(tail, encode(q))
    obs = [0] * 4 + tail + [len(queries)]
    outer = [1] + obs
    queries = list(queries)
    head, length = [0] * 4, len(queries)
    no_work = int(length == 0)
    done, state = no_work, [0] * 25
    ok = all(fields(q)["tx"] < 65536 for q in queries)
    rows, buffer = [], b""
    for it in range(limit // PERIOD):
        row = [(lane >> (8 * k)) & 0xFF for lane in state for k in range(8)] + head + [length, done]
        for c in range(PERIOD):
            q = list(ZERO_QUERY)
            should_pop = length != 0
            if should_pop:
                q = queries.pop(0)
                head = zko.queue_tail4_push20(head, encode(q))
                length -= 1
            is_last = should_pop and length == 0
            row += list(q)
            buffer += into_bytes(q)
            cont = not done
            if len(buffer) >= RATE:
                block, buffer = buffer[:RATE], buffer[RATE:]
                if cont:
                    state = absorb(state, block)
            if cont and is_last:
                last = bytearray(buffer.ljust(RATE, b"\0"))
                if len(buffer) == RATE - 1:
                    last[len(buffer)] = 0x81
                else:
                    last[len(buffer)] = 0x01
                    last[RATE - 1] = 0x80
                state = absorb(state, bytes(last))
            done = 1 if (done or is_last) else 0
        assert buffer == b""
        rows.append(row)
    completed = length == 0
    if not completed or head != tail:
        ok = False
    digest = zko.keccak256(b"") if no_work else b"".join(state[i].to_bytes(8, "little") for i in range(4))
    z4 = [0] * 4
    compact = [1, int(completed)] + zko.commit_encoding(obs) + (zko.commit_encoding(list(digest)) if completed else z4) + z4 + z4
    return dict(outer=outer, rows=rows, satisfiable=ok, digest=digest, public_input=zko.commit_encoding(compact))
