"""Transcribes the reference's log_sorter fixture (data only) into tests/golden/log_sorter_fixture.json.
source: /root/reference/src/log_sorter/mod.rs:638-815 (witness_input_unsorted / witness_input_sorted)."""
import json, os, re
src = open('/root/reference/src/log_sorter/mod.rs').read()
def parse_val(tok):
    tok = tok.strip().rstrip(',')
    if tok == 'bool_false': return 0
    if tok == 'bool_true': return 1
    if tok in ('zero_8', 'zero_32'): return 0
    if tok == 'one_8': return 1
    for pat in (r'from_low_u64_le\((\d+)\)', r'from_dec_str\(\s*"(\d+)"', r'allocated_constant\(cs,\s*(\d+)\)'):
        m = re.search(pat, tok, re.S)
        if m: return int(m.group(1))
    raise ValueError(tok)
F = ['address', 'key', 'read_value', 'written_value', 'rw_flag', 'aux_byte', 'rollback', 'is_service', 'shard_id', 'tx_number_in_block', 'timestamp']
def parse_queries(body):
    out = []
    for m in re.finditer(r'LogQuery::<F>\s*\{(.*?)\n\s*\}[;,]', body, re.S):
        blk, d = m.group(1), {}
        for f in F:
            mm = re.search(r'\b' + f + r':\s*(.*?)(?=,\n\s*(?:' + '|'.join(F) + r'):|\s*$)', blk, re.S)
            d[f] = str(parse_val(mm.group(1)))
        out.append(d)
    return out
a, b = src.index('fn witness_input_unsorted'), src.index('fn witness_input_sorted')
uns, srt = parse_queries(src[a:b]), parse_queries(src[b:])
assert len(uns) == 4 and len(srt) == 4, (len(uns), len(srt))
json.dump({"source": "/root/reference/src/log_sorter/mod.rs:638-815, limit 16 (:617); address = argument of Address::from_low_u64_le",
           "limit": 16, "unsorted": uns, "sorted": srt}, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'log_sorter_fixture.json'), 'w'), indent=0)
print([(d['timestamp'], d['rollback']) for d in uns], [(d['timestamp'], d['rollback']) for d in srt])
