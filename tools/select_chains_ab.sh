# Mux chains (ZKGL_SELECT_CHAINS=1 in a -DZKGL_SELECT_CHAINS_KERNEL build): parity, then A/B of k_witness_loop on ONE box.
# Before `gpurun -- bash tools/select_chains_ab.sh`, build the variant library in the container (it travels with the snapshot):
#   (cd era-zkevm_circuits_amd && ZKGL_DEFS=-DZKGL_SELECT_CHAINS_KERNEL ZKGL_OUT=../libzkgl_chains.so ZKGL_BUILD_DIR=../build/var/chains ./build.sh)
L=$PWD/era-zkevm_circuits_amd/libzkgl_chains.so
[ -f "$L" ] || { echo "build $L first (see the header)"; exit 1; }
ZKGL_LIB=$L ZKGL_SELECT_CHAINS=1 timeout 900 python -m pytest tests/test_gpu_main_vm.py tests/test_fused_check.py tests/test_fuzz_programs.py -m gpu -x -q > gpurun_out/t_chains.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/t_chains.log | tail -4
run() { env "$@" timeout 250 python bench.py --headline-only --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$*', round(r['avg_launch_ms'],2), round(r['shader_clock_mhz']), d['value'], d['commitment_checksum'])"; }
run A=product
run ZKGL_LIB=$L ZKGL_SELECT_CHAINS=0
run ZKGL_LIB=$L ZKGL_SELECT_CHAINS=1
run ZKGL_SELECT_CHAINS=1
run A=product
run ZKGL_LIB=$L ZKGL_SELECT_CHAINS=1
