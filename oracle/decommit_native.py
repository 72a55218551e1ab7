"""oracle/decommit_native.py — CPU ORACLE (test infrastructure): native restatement of
sort_and_deduplicate_code_decommittments_entry_point / _inner (/root/reference/src/sort_decommittment_requests/mod.rs:40-372)
and DecommitQuery::encode (src/base_structures/decommit_query/mod.rs:33-113)."""
from __future__ import annotations

from . import zko

P = zko.P
REPS, ENC, NCH = 2, 8, 9
OUTER_WORDS, LOOP_WORDS, CARRIED = 151, 87, 65


def dq(code_hash, page, is_first, timestamp):
    """flattened 11 words: code_hash limbs, page, is_first, timestamp (decommit_query/mod.rs:137-155)"""
    return [(code_hash >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [page, int(is_first), timestamp]


ZERO = [0] * 11


def encode(q):
    h, page, first, ts = q[0:8], q[8], q[9], q[10]
    pb = [(page >> (8 * k)) & 0xFF for k in range(4)]
    tb = [(ts >> (8 * k)) & 0xFF for k in range(4)]
    return [h[0] + (pb[0] << 32) + (pb[1] << 40) + (pb[2] << 48), h[1] + (pb[3] << 32) + (tb[0] << 40) + (tb[1] << 48),
            h[2] + (tb[2] << 32) + (tb[3] << 40) + (first << 48)] + h[3:8]


def queue_simulate(items):
    tail = [0] * 12
    for it in items:
        tail = zko.queue_full_push(tail, encode(it))
    return tail


def empty_fsm():
    return dict(initial=[0] * 25, sorted=[0] * 25, final=[0] * 25, lhs=[0, 0], rhs=[0, 0], prev_key=[0] * 9, first_ts=0, prev_record=[0] * 11)


def flatten_fsm(f):
    return list(f["initial"]) + list(f["sorted"]) + list(f["final"]) + list(f["lhs"]) + list(f["rhs"]) + list(f["prev_key"]) + \
        [f["first_ts"]] + list(f["prev_record"])


def key_of(q):
    return q[10] | sum(q[i] << (32 * (i + 1)) for i in range(8))   # timestamp = least significant limb


def instance(unsorted, sorted_items, limit, start_flag=True, fsm_in=None, obs=None, total=None):
    """`unsorted` / `sorted_items`: what the two queues still hold; start instances derive the observable states from them."""
    if start_flag:
        n = len(unsorted)
        assert len(sorted_items) == n
        obs_initial = [0] * 12 + queue_simulate(unsorted) + [n]
        obs_sorted = [0] * 12 + queue_simulate(sorted_items) + [n]
        fsm_in = empty_fsm()
        f = dict(empty_fsm(), initial=list(obs_initial), sorted=list(obs_sorted), lhs=[1, 1], rhs=[1, 1])
    else:
        obs_initial, obs_sorted = obs
        f = {k: list(v) if isinstance(v, list) else v for k, v in fsm_in.items()}
    ch = zko.fs_challenges(obs_initial[12:25] + obs_sorted[12:25], REPS, NCH)
    outer = [int(start_flag)] + list(obs_initial) + list(obs_sorted) + flatten_fsm(fsm_in)
    assert len(outer) == OUTER_WORDS
    unsorted, sorted_items = list(unsorted), list(sorted_items)
    o_head, o_tail, o_len = f["initial"][0:12], f["initial"][12:24], f["initial"][24]
    s_head, s_tail, s_len = f["sorted"][0:12], f["sorted"][12:24], f["sorted"][24]
    r_head, r_tail, r_len = f["final"][0:12], f["final"][12:24], f["final"][24]
    lhs, rhs = list(f["lhs"]), list(f["rhs"])
    prev_key, first_ts, prev_record = list(f["prev_key"]), f["first_ts"], list(f["prev_record"])
    ok = o_len == s_len
    prev_trivial = 1 if (o_len == 0 or start_flag) else 0
    rows, result = [], []

    def push_result(rec, ts):
        nonlocal r_tail, r_len
        q = rec[0:9] + [1, ts]
        result.append(q)
        r_tail = zko.queue_full_push(r_tail, encode(q))
        r_len += 1

    for _ in range(limit):
        should_pop = o_len != 0
        if (o_len == 0) != (s_len == 0):
            ok = False
        uq = unsorted.pop(0) if should_pop else ZERO
        sq = sorted_items.pop(0) if should_pop and sorted_items else ZERO
        rows.append([prev_trivial] + lhs + rhs + o_head + [o_len] + s_head + [s_len] + r_tail + [r_len] + prev_key + [first_ts] + prev_record +
                    list(uq) + list(sq))
        ue, se = encode(uq), encode(sq)
        if should_pop:
            o_head = zko.queue_full_push(o_head, ue); o_len -= 1
            s_head = zko.queue_full_push(s_head, se); s_len -= 1
            for r in range(REPS):
                lc = rc = ch[r][ENC]
                for i in range(ENC):
                    lc = (lc + ue[i] * ch[r][i]) % P
                    rc = (rc + se[i] * ch[r][i]) % P
                lhs[r] = lhs[r] * lc % P
                rhs[r] = rhs[r] * rc % P
        packed = [sq[10]] + sq[0:8]
        if should_pop and not key_of(sq) > (prev_key[0] | sum(prev_key[1 + i] << (32 * (i + 1)) for i in range(8))):
            ok = False
        same_hash = prev_record[0:8] == sq[0:8]
        if (not same_hash) and should_pop and not sq[9]:
            ok = False
        if same_hash and not prev_trivial and sq[8] != prev_record[8]:
            ok = False
        if (not prev_trivial) and not same_hash:
            push_result(prev_record, first_ts)
        prev_trivial = 0 if should_pop else 1
        if not same_hash:
            first_ts = sq[10]
        prev_record, prev_key = list(sq), packed
    completed = int(o_len == 0)
    if (o_len == 0) != (s_len == 0):
        ok = False
    if (not prev_trivial) and completed:
        push_result(prev_record, first_ts)
    if o_len == 0 and o_head != o_tail:
        ok = False
    if s_len == 0 and s_head != s_tail:
        ok = False
    if completed and lhs != rhs:
        ok = False
    fsm_out = dict(initial=o_head + o_tail + [o_len], sorted=s_head + s_tail + [s_len], final=r_head + r_tail + [r_len], lhs=lhs, rhs=rhs,
                   prev_key=prev_key, first_ts=first_ts, prev_record=prev_record)
    obs_out = fsm_out["final"] if completed else [0] * 25
    z4 = [0] * 4
    compact = [int(start_flag), completed] + zko.commit_encoding(list(obs_initial) + list(obs_sorted)) + \
        (zko.commit_encoding(obs_out) if completed else z4) + \
        (z4 if start_flag else zko.commit_encoding(flatten_fsm(fsm_in))) + \
        (z4 if completed else zko.commit_encoding(flatten_fsm(fsm_out)))
    return dict(outer=outer, rows=rows, fsm_out=fsm_out, completed=completed, satisfiable=ok, result=result, rest=(unsorted, sorted_items),
                obs=(obs_initial, obs_sorted), public_input=zko.commit_encoding(compact))


def random_decommits(rng, n_hashes, max_repeats=4):
    """decommitment requests: every code hash requested 1..max_repeats times at increasing timestamps (first one is_first)"""
    items, ts = [], 1
    hashes = [int.from_bytes(rng.bytes(32), "little") for _ in range(n_hashes)]
    seen = {}
    order = [h for h in hashes for _ in range(int(rng.integers(1, max_repeats + 1)))]
    order = [order[i] for i in rng.permutation(len(order))]
    for h in order:
        page = seen.setdefault(h, 2048 + 8 * len(seen))
        items.append(dq(h, page, h not in {x[1] for x in items}, ts))
        items[-1] = (items[-1], h)
        ts += int(rng.integers(1, 50))
    items = [dq(h, seen[h], first, q[10]) for (q, h), first in zip(items, _first_flags([h for _, h in items]))]
    return items, sorted(items, key=key_of)


def _first_flags(hs):
    s, out = set(), []
    for h in hs:
        out.append(h not in s)
        s.add(h)
    return out
