# GPU box, repo root, call B of round 6 (after call A is green): the NARROW STORE against the ordinary store on ONE box, the read-counter calibration,
# the elimination probes of both kernels.  -> gpurun_out/r6b_*  (what is judged is copied into profiles/)
mkdir -p gpurun_out
OUT=gpurun_out/r6b_ab.txt; : > $OUT
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$1', r['kernel'], 'loop_ms', round(r['avg_launch_ms'],2), 'clock', round(r['shader_clock_mhz']), 'step_ms', round(d['ms_per_step'],2), 'value', round(d['value']/1e9,1), 'G frac', round(r['frac'],3), 'bytes/cycle', r.get('bytes_written_per_cycle'), 'checksum', d['commitment_checksum'])"; }
ab() { timeout 400 python bench.py --headline-only --steps 5 --warmup 2 --fixture $1 $2 2>gpurun_out/r6b_err.txt | line "[$1 ${2:-ordinary}]" | tee -a $OUT; }
# ---- (1) parity first (the same tests the suite of call A ran; a red one stops the A/B)
timeout 900 python -m pytest tests/test_zz_round6_narrow_store.py -m "gpu or not gpu" -x -q -p no:cacheprovider > gpurun_out/r6b_parity.log 2>&1; tail -2 gpurun_out/r6b_parity.log | tee -a $OUT
grep -q " passed" gpurun_out/r6b_parity.log && ! grep -q "failed" gpurun_out/r6b_parity.log || { echo "narrow-store parity is not green: no A/B" | tee -a $OUT; exit 1; }
# ---- (2) k_witness_loop vs k_witness_loop_narrow, both fixtures, alternating, twice
for rep in 1 2; do for fx in default realistic; do ab $fx ""; ab $fx "--narrow-store"; done; done
# ---- (3) the full line with mode_narrow_store (plain + deferred on top) and the widening cost
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-feed --with-narrow-store-mode > gpurun_out/r6b_bench_modes.json 2> gpurun_out/r6b_bench_modes.err
python -c "import json; d=json.loads(open('gpurun_out/r6b_bench_modes.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('mode_narrow_store'), indent=1)); print('deferred', json.dumps(d.get('mode_p2_intermediates_deferred'), indent=1))" | tee -a $OUT
# ---- (4) the narrow kernel's own profile: kernel trace + FETCH / WRITE / SQ passes + the driver's bench command line, all with --narrow-store
#          -> r6n_kernel_trace.md, pmc_r6n_*.txt, r6n_bench.json, pmc_r6n.json (tools/pmc_json.py for zke::k_witness_loop_narrow)
TAG=r6n EXTRA_BENCH_ARGS=--narrow-store BENCH_ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline" KT_STEPS=5 timeout 1500 bash tools/profile_tag.sh > gpurun_out/r6n_profile.log 2>&1; tail -6 gpurun_out/r6n_profile.log | tee -a $OUT
grep -h "k_witness_loop_narrow \|k_check_prog_t<true>" gpurun_out/pmc_r6n_*.txt | tee -a $OUT
# ---- (5) read-counter calibration in this kernel's access patterns (tools/rprobe.hip)
hipcc --offload-arch=gfx950 -O3 tools/rprobe.hip -o gpurun_out/rprobe 2>/dev/null && gpurun_out/rprobe > gpurun_out/rprobe.json && cat gpurun_out/rprobe.json | tee -a $OUT
PMC_CMD="$PWD/gpurun_out/rprobe" tools/pmc_pass.sh rprobe FETCH_SIZE | tee -a $OUT
python tools/pmc_json.py r6n zke::k_witness_loop_narrow > gpurun_out/pmc_r6n.json 2> gpurun_out/pmc_json_r6n.err || tail -2 gpurun_out/pmc_json_r6n.err
[ -f gpurun_out/r6_bench.json ] && python tools/pmc_json.py r6 > gpurun_out/pmc_r6.json 2>/dev/null   # (again, now with the measured FETCH_SIZE factor beside the guide's)
# ---- (6) elimination probes of both kernels (tools/probe_variants.sh in the container built libzkgl_var_{S,L,SL,ALLV}.so): what binds the narrow kernel?
for t in S L SL ALLV; do
  lib=$PWD/era-zkevm_circuits_amd/libzkgl_var_$t.so; [ -f $lib ] || continue
  for nar in "" "--narrow-store"; do
    ZKGL_STUB_RUN=1 ZKGL_LIB=$lib timeout 300 python bench.py --headline-only --steps 3 --warmup 1 $nar 2>/dev/null | line "[probe -$t ${nar:-ordinary}]" | tee -a $OUT
  done
done
cat $OUT
