"""oracle/log_sorter_native.py — CPU ORACLE (test infrastructure): native restatement of
sort_and_deduplicate_events_entry_point / repack_and_prove_events_rollbacks_inner
(/root/reference/src/log_sorter/mod.rs:34-441)."""
from __future__ import annotations

import numpy as np

from . import zko
from .storage_native import ENC, NCH, P, REPS, ZERO_QUERY, encode, fields, log_query, queue4_simulate  # noqa: F401


def empty_fsm():
    return dict(lhs=[0, 0], rhs=[0, 0], unsorted=[0] * 9, sorted=[0] * 9, result=[0] * 9, prev_key=0, prev_item=[0] * 36)


def flatten_fsm(f):
    return list(f["lhs"]) + list(f["rhs"]) + list(f["unsorted"]) + list(f["sorted"]) + list(f["result"]) + [f["prev_key"]] + list(f["prev_item"])


def cleaned(prev):
    """query_to_add (mod.rs:372-386): read_value, rw, aux, rollback, timestamp zeroed"""
    f = fields(prev)
    return f["address"] + f["key"] + [0] * 8 + f["written"] + [0, 0, 0, f["is_service"], f["shard"], f["tx"], 0]


def instance(unsorted, sorted_items, limit):
    n = len(unsorted)
    assert len(sorted_items) == n and n <= limit
    ub, utail = queue4_simulate([encode(q) for q in unsorted])
    sb, stail = queue4_simulate([encode(q) for q in sorted_items])
    obs_unsorted, obs_sorted = [0] * 4 + utail + [n], [0] * 4 + stail + [n]
    ch = zko.fs_challenges(utail + [n] + stail + [n], REPS, NCH)
    fsm_in = empty_fsm()
    ok = True
    lhs, rhs = [1, 1], [1, 1]
    u_head, s_head, u_len, s_len = [0] * 4, [0] * 4, n, n
    r_tail, r_len = [0] * 4, 0
    prev_trivial, prev_key, prev_item = 1, 0, list(ZERO_QUERY)
    result_items, rows = [], []

    def push_result(item):
        nonlocal r_tail, r_len
        q = cleaned(item)
        result_items.append(q)
        r_tail = zko.queue_tail4_push20(r_tail, encode(q))
        r_len += 1

    for k in range(limit):
        uq = unsorted[k] if k < n else ZERO_QUERY
        sq = sorted_items[k] if k < n else ZERO_QUERY
        rows.append([prev_trivial] + lhs + rhs + u_head + [u_len] + s_head + [s_len] + r_tail + [r_len] + [prev_key] + prev_item + list(uq) + list(sq))
        should_pop = u_len != 0
        trivial = not should_pop
        ue, se = encode(uq), encode(sq)
        if should_pop:
            u_head = zko.queue_tail4_push20(u_head, ue)
            s_head = zko.queue_tail4_push20(s_head, se)
            u_len -= 1; s_len -= 1
            if not fields(uq)["rw"] or not fields(sq)["rw"]:
                ok = False
            for r in range(REPS):
                lc = rc = ch[r][ENC]
                for i in range(ENC):
                    lc = (lc + ue[i] * ch[r][i]) % P
                    rc = (rc + se[i] * ch[r][i]) % P
                lhs[r] = lhs[r] * lc % P
                rhs[r] = rhs[r] * rc % P
        f, pf = fields(sq), fields(prev_item)
        sorting_key = f["ts"]
        same_log = sorting_key == prev_key
        if should_pop and sorting_key < prev_key:
            ok = False
        if should_pop and not same_log and f["rollback"]:
            ok = False
        if should_pop and same_log and not f["rollback"]:
            ok = False
        same_body = f["key"] == pf["key"] and f["written"] == pf["written"]
        if same_log and not prev_trivial and not same_body:
            ok = False
        if (not prev_trivial) and ((not same_log) or trivial) and not pf["rollback"]:
            push_result(prev_item)
        prev_trivial, prev_item, prev_key = int(trivial), list(sq), sorting_key
    if (not prev_trivial) and (not fields(prev_item)["rollback"]) and u_len == 0:
        push_result(prev_item)
    completed = u_len == 0
    if u_len == 0 and u_head != utail: ok = False
    if s_len == 0 and s_head != stail: ok = False
    if (u_len == 0) != (s_len == 0): ok = False
    permutation_ok = lhs == rhs
    if completed and not permutation_ok:
        ok = False
    fsm_out = dict(lhs=lhs, rhs=rhs, unsorted=u_head + utail + [u_len], sorted=s_head + stail + [s_len], result=[0] * 4 + r_tail + [r_len],
                   prev_key=prev_key, prev_item=prev_item)
    obs_in = obs_unsorted + obs_sorted
    obs_out = ([0] * 4 + r_tail + [r_len]) if completed else [0] * 9
    c_obs_in, c_obs_out = zko.commit_encoding(obs_in), zko.commit_encoding(obs_out)
    c_fsm_in, c_fsm_out = zko.commit_encoding(flatten_fsm(fsm_in)), zko.commit_encoding(flatten_fsm(fsm_out))
    z4 = [0] * 4
    compact = [1, int(completed)] + c_obs_in + (c_obs_out if completed else z4) + z4 + (z4 if completed else c_fsm_out)
    outer = [1] + obs_in + flatten_fsm(fsm_in)
    assert len(outer) == 87 and all(len(r) == 129 for r in rows)
    return dict(outer=outer, loop=rows, fsm_out=fsm_out, completed=completed, commitment=zko.commit_encoding(compact), satisfiable=ok,
                permutation_ok=permutation_ok, result_items=result_items)


def pack_streams(instances, limit):
    B = len(instances)
    outer = np.array([i["outer"] for i in instances], dtype=np.uint64).T.copy()
    loop = np.array([row for i in instances for row in i["loop"]], dtype=np.uint64).T.copy()
    assert outer.shape == (87, B) and loop.shape == (129, B * limit)
    return outer, loop


def random_events(rng, n_events, rollback_frac=0.3):
    """events with unique timestamps; a fraction is later rolled back (a twin with rollback=1, same timestamp/body)"""
    fwd = []
    for t in range(n_events):
        fwd.append(dict(address=int(rng.integers(1, 1 << 40)), key=int.from_bytes(rng.bytes(32), "little"),
                        written_value=int.from_bytes(rng.bytes(32), "little"), rw_flag=1, aux_byte=int(rng.integers(0, 3)),
                        is_service=int(rng.integers(0, 2)), shard_id=0, tx_number_in_block=int(rng.integers(0, 100)), timestamp=10 + 3 * t))
    unsorted = [log_query(**e) for e in fwd]
    rolled = [e for e in fwd if rng.random() < rollback_frac]
    for e in reversed(rolled):
        unsorted.append(log_query(rollback=1, **e))
    order = sorted(range(len(unsorted)), key=lambda i: (unsorted[i][35], unsorted[i][31]))
    return unsorted, [unsorted[i] for i in order]
