"""Transcribes the reference's demux_log_queue fixture (data only) into tests/golden/demux_fixture.json.
source: /root/reference/src/demux_log_queue/mod.rs:595-923 (witness_input_unsorted), limit 16 (:587)."""
import json, os, re
src = open('/root/reference/src/demux_log_queue/mod.rs').read()
def parse_val(tok):
    tok = tok.strip().rstrip(',')
    if tok == 'bool_false': return 0
    if tok == 'bool_true': return 1
    if tok in ('zero_8', 'zero_32'): return 0
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
uns = parse_queries(src[src.index('fn witness_input_unsorted'):])
assert len(uns) == 16, len(uns)
json.dump({"source": "/root/reference/src/demux_log_queue/mod.rs:595-923, limit 16 (:587); address = argument of Address::from_low_u64_le",
           "limit": 16, "unsorted": uns}, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'demux_fixture.json'), 'w'), indent=0)
print([(d['address'], d['timestamp'], d['aux_byte']) for d in uns])
