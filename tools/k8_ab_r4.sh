# GPU box, repo root.  (1) device parity of the ByteBuffer macro-op and of the strand-form flag planes, (2) C3 keccak step A/B on ONE box:
# product / ZKGL_BYTEBUF_MACRO=1 / ZKGL_STRAND_PLANES=1 / both.  -> gpurun_out/k8_ab.txt
mkdir -p gpurun_out
export ZKGL_TEST_UNMEASURED=1
timeout 600 python -m pytest tests/test_bytebuf_macro.py -m gpu -x -q > gpurun_out/t_bytebuf.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/t_bytebuf.log | tail -4
ZKGL_STRAND_PLANES=1 timeout 900 python -m pytest tests/test_gpu_cs.py tests/test_gpu_fsm_seed.py tests/test_queue_seed.py tests/test_fuzz_programs.py tests/test_fused_check.py -m gpu -x -q > gpurun_out/t_splanes.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/t_splanes.log | tail -4
ZKGL_STRAND_PLANES=1 ZKGL_BYTEBUF_MACRO=1 timeout 600 python -m pytest tests/test_bytebuf_macro.py -m gpu -x -q > gpurun_out/t_both.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/t_both.log | tail -4
for v in "A=0" "ZKGL_BYTEBUF_MACRO=1" "ZKGL_STRAND_PLANES=1" "ZKGL_BYTEBUF_MACRO=1 ZKGL_STRAND_PLANES=1" "A=0" "ZKGL_BYTEBUF_MACRO=1"; do
  env $v CONFIGS=C3k timeout 400 python tests/config_timings.py 2>/dev/null | grep "^{" | cut -c1-400 | sed "s/^/[$v] /"
done
