#!/bin/bash
# GPU box, repo root: the round-3 evidence set -> gpurun_out/r3_* (copy what is judged into profiles/).
#  1. rocprofv3 --kernel-trace --stats of the default bench command            -> r3_kernel_trace.md, r3_bench_under_rocprof.json
#  2. PMC passes (their own runs): FETCH_SIZE, WRITE_SIZE at the default batch   -> pmc_r3_fetch.txt, pmc_r3_write.txt
#  3. the bench itself, no profiler                                             -> r3_bench.json
set -u
ROOT=$(pwd); mkdir -p "$ROOT/gpurun_out"
B=${B:-384}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_r3
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_r3 -o kt -- python "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline < /dev/null > "$ROOT/gpurun_out/r3_bench_under_rocprof.json" 2> /tmp/kt_r3.err
db=$(find /tmp/kt_r3 -name "*_results.db" | head -1)
[ -n "$db" ] && python "$ROOT/profiles/summarize_rocpd.py" "$db" > "$ROOT/gpurun_out/r3_kernel_trace.md"
cd "$ROOT"
export PMC_CMD="python $ROOT/bench.py --batch $B --seed-windows 2 --steps 2 --warmup 0 --no-cpu-baseline"
tools/pmc_pass.sh r3_fetch FETCH_SIZE > /dev/null
tools/pmc_pass.sh r3_write WRITE_SIZE > /dev/null
unset PMC_CMD
timeout 900 python bench.py ${BENCH_ARGS:-} < /dev/null > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err || tail -3 gpurun_out/r3_bench.err
head -24 gpurun_out/r3_kernel_trace.md
grep -E "k_witness_loop|k_check_prog|k_check_p2|k_vm_" gpurun_out/pmc_r3_fetch.txt gpurun_out/pmc_r3_write.txt
