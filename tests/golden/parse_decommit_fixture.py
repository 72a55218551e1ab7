"""Transcribes the reference's sort_decommittment_requests fixture (data only) into tests/golden/decommit_fixture.json.
source: /root/reference/src/sort_decommittment_requests/mod.rs:565-1390 (witness_input_unsorted / witness_input_sorted), limit 16 (:540)."""
import json, os, re
src = open('/root/reference/src/sort_decommittment_requests/mod.rs').read()
def parse_val(tok):
    tok = tok.strip().rstrip(',')
    if tok == 'bool_false': return 0
    if tok == 'bool_true': return 1
    for pat in (r'from_dec_str\(\s*"(\d+)"', r'allocated_constant\(cs,\s*(\d+)\)'):
        m = re.search(pat, tok, re.S)
        if m: return int(m.group(1))
    raise ValueError(tok)
F = ['code_hash', 'page', 'is_first', 'timestamp']
def parse_queries(body):
    out = []
    for m in re.finditer(r'DecommitQuery::<F>\s*\{(.*?)\n\s*\}[;,]', body, re.S):
        blk, d = m.group(1), {}
        for f in F:
            mm = re.search(r'\b' + f + r':\s*(.*?)(?=,\n\s*(?:' + '|'.join(F) + r'):|\s*$)', blk, re.S)
            d[f] = str(parse_val(mm.group(1)))
        out.append(d)
    return out
a, b = src.index('fn witness_input_unsorted'), src.index('fn witness_input_sorted')
uns, srt = parse_queries(src[a:b]), parse_queries(src[b:])
assert len(uns) == 29 and len(srt) == 29, (len(uns), len(srt))
json.dump({"source": "/root/reference/src/sort_decommittment_requests/mod.rs:565-1390, limit 16 (:540)", "limit": 16, "unsorted": uns, "sorted": srt},
          open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'decommit_fixture.json'), 'w'), indent=0)
key = lambda d: (int(d['code_hash']), int(d['timestamp']))
print('sorted side ordered:', [key(d) for d in srt] == sorted(key(d) for d in srt), 'permutation:', sorted(map(key, uns)) == sorted(map(key, srt)))
print([(d['code_hash'][-4:], d['page'], d['is_first'], d['timestamp']) for d in srt])
