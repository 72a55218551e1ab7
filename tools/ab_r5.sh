# GPU box, repo root, after tools/variants_r5.sh in the container: parity + A/B of every opt-in device path against the product on ONE box.
# -> gpurun_out/r5_ab.txt (decisions: promote with the numbers committed, or delete)
mkdir -p gpurun_out
P=$PWD/era-zkevm_circuits_amd
OUT=gpurun_out/r5_ab.txt; : > $OUT
t() { local name=$1; shift; env "$@" > gpurun_out/t_$name.log 2>&1; echo "[$name] $(grep -E 'passed|failed|rror' gpurun_out/t_$name.log | tail -2 | tr '\n' ' ')" | tee -a $OUT; }
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', 'loop_ms', round(r['avg_launch_ms'],2), 'clock', round(r['shader_clock_mhz']), 'value', round(d['value']/1e9,1), 'G checksum', d['commitment_checksum'])"; }
vm() { local name=$1; shift; env "$@" timeout 300 python bench.py --headline-only --steps 5 --warmup 2 2>/dev/null | line "[$name default fixture]" | tee -a $OUT
       env "$@" timeout 300 python bench.py --headline-only --steps 5 --warmup 2 --fixture realistic 2>/dev/null | line "[$name realistic fixture]" | tee -a $OUT; }
# ---- (1) batched inversions: parity (whole-trace main_vm tests, fused differential on a sample), then k_witness_loop on both fixtures, product first and last
if [ -f $P/libzkgl_binv.so ]; then
  t binv ZKGL_LIB=$P/libzkgl_binv.so timeout 1200 python -m pytest tests/test_gpu_main_vm.py tests/test_fused_check.py tests/test_fuzz_programs.py tests/test_gpu_cs.py -m gpu -x -q
  vm product A=0; vm binv ZKGL_LIB=$P/libzkgl_binv.so; vm product A=0; vm binv ZKGL_LIB=$P/libzkgl_binv.so
fi
# ---- (1b) merged gated permutations (profiles/r5_iszero_stats.json: 15.0 -> 7.9 permutations per wavefront on the default fixture), alone and with (1)
for v in p2m p2m_binv; do
  if [ -f $P/libzkgl_$v.so ]; then
    t $v ZKGL_LIB=$P/libzkgl_$v.so timeout 1200 python -m pytest tests/test_gpu_main_vm.py tests/test_gpu_full_size.py tests/test_fused_check.py tests/test_fuzz_programs.py -m gpu -x -q
    vm product A=0; vm $v ZKGL_LIB=$P/libzkgl_$v.so; vm product A=0; vm $v ZKGL_LIB=$P/libzkgl_$v.so
  fi
done
# VALU instructions and busy cycles of the loop kernel: product vs both levers (the reduction the harness statistics predict: profiles/r5_iszero_stats.json)
if [ -f $P/libzkgl_p2m_binv.so ]; then
  for v in product p2m_binv; do
    L=$P/libzkgl.so; [ $v = product ] || L=$P/libzkgl_$v.so
    export PMC_CMD="env ZKGL_LIB=$L python $PWD/bench.py --batch 384 --seed-windows 2 --steps 2 --warmup 0 --no-cpu-baseline --headline-only"
    tools/pmc_pass.sh ab_valu_$v SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES > /dev/null 2>&1
    grep "k_witness_loop " gpurun_out/pmc_ab_valu_$v.txt | sed "s/^/[$v] /" | tee -a $OUT
    unset PMC_CMD
  done
fi
# ---- (2) mux chains
if [ -f $P/libzkgl_chains.so ]; then
  t chains ZKGL_LIB=$P/libzkgl_chains.so ZKGL_SELECT_CHAINS=1 timeout 900 python -m pytest tests/test_gpu_main_vm.py tests/test_fused_check.py tests/test_fuzz_programs.py -m gpu -x -q
  vm chains_lib_off ZKGL_LIB=$P/libzkgl_chains.so ZKGL_SELECT_CHAINS=0; vm chains ZKGL_LIB=$P/libzkgl_chains.so ZKGL_SELECT_CHAINS=1; vm order_only ZKGL_SELECT_CHAINS=1
fi
# ---- (3) K8: ByteBuffer macro-op, strand-form flag planes (keccak / sha256 / eip_4844 steps)
if [ -f $P/libzkgl_k8.so ]; then
  t bytebuf ZKGL_LIB=$P/libzkgl_k8.so timeout 600 python -m pytest tests/test_zz_round5_gpu.py -m gpu -x -q -k bytebuf
  t splanes ZKGL_LIB=$P/libzkgl_k8.so ZKGL_STRAND_PLANES=1 timeout 1200 python -m pytest tests/test_gpu_cs.py tests/test_gpu_fsm_seed.py tests/test_queue_seed.py tests/test_fuzz_programs.py tests/test_fused_check.py tests/test_zz_round5_gpu.py -m gpu -k "forged or sha4" -x -q
  t both ZKGL_LIB=$P/libzkgl_k8.so ZKGL_STRAND_PLANES=1 ZKGL_BYTEBUF_MACRO=1 timeout 600 python -m pytest tests/test_zz_round5_gpu.py -m gpu -x -q -k bytebuf
  for v in "A=0" "ZKGL_LIB=$P/libzkgl_k8.so" "ZKGL_LIB=$P/libzkgl_k8.so ZKGL_BYTEBUF_MACRO=1" "ZKGL_LIB=$P/libzkgl_k8.so ZKGL_STRAND_PLANES=1" "ZKGL_LIB=$P/libzkgl_k8.so ZKGL_BYTEBUF_MACRO=1 ZKGL_STRAND_PLANES=1" "A=0"; do
    env $v CONFIGS=C3k,C3s,C5 timeout 500 python tests/config_timings.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('[$v]', d['config'], 'step_ms', d['step_ms'], 'loop_ms', d['k_witness_loop_ms'], 'seed_s', d['seed_s'])" | sed "s#$P/##g" | tee -a $OUT
  done
fi
# ---- (4) a16: the reference's SHA table set as a macro-op (ZK_OP_SHA256_ROUNDS a = 1): parity, then the C3 sha256 step under that table set
if [ -f $P/libzkgl_sha4.so ]; then
  t sha4 ZKGL_LIB=$P/libzkgl_sha4.so timeout 900 python -m pytest tests/test_zz_round5_gpu.py -m gpu -x -q -k sha4
  t sha4_tables ZKGL_LIB=$P/libzkgl_sha4.so ZKGL_SHA4_MACRO=1 timeout 900 python -m pytest tests/test_sha256_reference_tables.py -m gpu -x -q
  t sha4_forged ZKGL_LIB=$P/libzkgl_sha4.so ZKGL_SHA4_MACRO=1 timeout 900 python -m pytest tests/test_zz_round5_gpu.py -m gpu -k forged -x -q
  for v in "A=0" "ZKGL_LIB=$P/libzkgl_sha4.so ZKGL_SHA4_MACRO=1" "A=0" "ZKGL_LIB=$P/libzkgl_sha4.so ZKGL_SHA4_MACRO=1"; do
    env $v CONFIGS=C3s,C3s4 timeout 600 python tests/config_timings.py 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('[$v]', d['config'][:110], 'step_ms', d['step_ms'], 'loop_ms', d['k_witness_loop_ms'], 'rows', d['rows_per_instance'])" | sed "s#$P/##g" | tee -a $OUT
  done
fi
cat $OUT
