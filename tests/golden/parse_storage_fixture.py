import re, json, sys
src=open('/root/reference/src/storage_validity_by_grand_product/test_input.rs').read()
def parse_val(tok):
    tok=tok.strip().rstrip(',')
    if tok in ('bool_false',): return 0
    if tok in ('bool_true',): return 1
    if tok in ('zero_8','zero_32'): return 0
    m=re.search(r'from_low_u64_le\((\d+)\)',tok)
    if m: return int(m.group(1))
    m=re.search(r'from_dec_str\(\s*"(\d+)"',tok,re.S)
    if m: return int(m.group(1))
    m=re.search(r'allocated_constant\(cs,\s*(\d+)\)',tok)
    if m: return int(m.group(1))
    raise ValueError(tok)
def parse_queries(body):
    out=[]
    for m in re.finditer(r'LogQuery::<F>\s*\{(.*?)\n\s*\}[;,]', body, re.S):
        blk=m.group(1)
        d={}
        for f in ['address','key','read_value','written_value','rw_flag','aux_byte','rollback','is_service','shard_id','tx_number_in_block','timestamp']:
            mm=re.search(r'\b'+f+r':\s*(.*?)(?=,\n\s*(?:address|key|read_value|written_value|rw_flag|aux_byte|rollback|is_service|shard_id|tx_number_in_block|timestamp):|\s*$)', blk, re.S)
            d[f]=parse_val(mm.group(1))
        out.append(d)
    return out
i=src.index('pub fn generate_test_input_sorted')
uns=parse_queries(src[:i])
srt_body=src[i:]
srt=parse_queries(srt_body)
# wrapper timestamps
wts=[int(x) for x in re.findall(r"TimestampedStorageLogRecord::<F>\s*\{\s*timestamp:\s*UInt32::allocated_constant\(cs,\s*(\d+)\)", srt_body)]
print(len(uns),len(srt),len(wts))
print(uns[0]); print(srt[0], wts[:4])
json.dump({"source":"/root/reference/src/storage_validity_by_grand_product/test_input.rs:12-632 (generate_test_input_unsorted / _sorted), transcribed as data; address = argument of Address::from_low_u64_le",
 "limit":16,"unsorted":[{k:str(v) for k,v in d.items()} for d in uns],"sorted":[dict({k:str(v) for k,v in d.items()}, record_timestamp=str(t)) for d,t in zip(srt,wts)]}, open('/root/repo/tests/golden/storage_fixture.json','w'), indent=0)
