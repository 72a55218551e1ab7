#!/bin/bash
# GPU box, repo root: the round-4 evidence set -> gpurun_out/r4_* (copy what is judged into profiles/).
#  1. rocprofv3 --kernel-trace --stats of the default bench command            -> r4_kernel_trace.md, r4_bench_under_rocprof.json
#  2. PMC passes (their own runs): FETCH_SIZE, WRITE_SIZE at the default batch   -> pmc_r4_fetch.txt, pmc_r4_write.txt
#  3. the bench itself, no profiler                                             -> r4_bench.json
set -u
ROOT=$(pwd); mkdir -p "$ROOT/gpurun_out"
B=${B:-384}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_r4
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_r4 -o kt -- python "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --headline-only < /dev/null > "$ROOT/gpurun_out/r4_bench_under_rocprof.json" 2> /tmp/kt_r4.err
db=$(find /tmp/kt_r4 -name "*_results.db" | head -1)
[ -n "$db" ] && python "$ROOT/profiles/summarize_rocpd.py" "$db" > "$ROOT/gpurun_out/r4_kernel_trace.md"
cd "$ROOT"
export PMC_CMD="python $ROOT/bench.py --batch $B --seed-windows 2 --steps 2 --warmup 0 --no-cpu-baseline --headline-only"
tools/pmc_pass.sh r4_fetch FETCH_SIZE > /dev/null
tools/pmc_pass.sh r4_write WRITE_SIZE > /dev/null
unset PMC_CMD
timeout 900 python bench.py ${BENCH_ARGS:-} < /dev/null > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err || tail -3 gpurun_out/r4_bench.err
head -24 gpurun_out/r4_kernel_trace.md
grep -E "k_witness_loop|k_check_prog|k_check_p2|k_vm_" gpurun_out/pmc_r4_fetch.txt gpurun_out/pmc_r4_write.txt
