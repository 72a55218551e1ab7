"""Transcribes the reference's code_unpacker_sha256 fixture (data only) into tests/golden/code_unpacker_fixture.json.
source: /root/reference/src/code_unpacker_sha256/mod.rs:618-718 (request page 2368 / timestamp 40973, code hash, 33 bytecode words), limit 40 (:582)."""
import json, os, re
src = open('/root/reference/src/code_unpacker_sha256/mod.rs').read()
i, j = src.index('fn get_code_hash_witness'), src.index('fn get_byte_code_witness')
code_hash = re.findall(r'"(\d+)"', src[i:j])
words = re.findall(r'"(\d+)"', src[j:])
assert len(code_hash) == 1 and len(words) == 33
json.dump({"source": "/root/reference/src/code_unpacker_sha256/mod.rs:618-718", "limit": 40, "page": 2368, "timestamp": 40973,
           "code_hash": code_hash[0], "code_words": words},
          open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'code_unpacker_fixture.json'), 'w'), indent=0)
print(code_hash[0][:20], len(words))
