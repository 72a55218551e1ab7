#!/bin/bash
# GPU box, repo root: the round-2 evidence set -> gpurun_out/r2_* (copy what is judged into profiles/).
#  1. rocprofv3 --kernel-trace --stats of the default bench command            -> r2_kernel_trace.md, r2_bench_under_rocprof.json
#  2. PMC passes (their own runs): FETCH_SIZE, WRITE_SIZE at the default batch   -> r2_pmc_fetch.txt, r2_pmc_write.txt
#  3. bench at a few batch sizes                                                -> r2_bench_b*.json
#  4. the other BASELINE configurations                                         -> r2_config_timings.jsonl
set -u
ROOT=$(pwd); mkdir -p "$ROOT/gpurun_out"
B=${B:-384}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_r2
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_r2 -o kt -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline < /dev/null > "$ROOT/gpurun_out/r2_bench_under_rocprof.json" 2> /tmp/kt_r2.err
db=$(find /tmp/kt_r2 -name "*_results.db" | head -1)
[ -n "$db" ] && python "$ROOT/profiles/summarize_rocpd.py" "$db" > "$ROOT/gpurun_out/r2_kernel_trace.md"
cd "$ROOT"
export PMC_CMD="python $ROOT/bench.py --batch $B --seed-windows 1 --steps 2 --warmup 0 --no-cpu-baseline"
tools/pmc_pass.sh r2_fetch FETCH_SIZE > /dev/null
tools/pmc_pass.sh r2_write WRITE_SIZE > /dev/null
tools/pmc_pass.sh r2_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY > /dev/null
unset PMC_CMD
for b in ${BATCHES:-64 256 384}; do
  timeout 900 python bench.py --steps 4 --warmup 1 --batch $b --no-cpu-baseline < /dev/null > gpurun_out/r2_bench_b$b.json 2> gpurun_out/r2_bench_b$b.err || tail -2 gpurun_out/r2_bench_b$b.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2_bench_b$b.json"))
    print("B=$b step %.2f ms loop %.2f gates %.2f outer %.2f seed %.2f s  %.1f G constraints/s, from raw %.1f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["other_kernels_ms"]["k_check_gates_loop"], d["roofline"]["other_kernels_ms"]["outer_post_and_checks_overlapped"], d["config"]["input_seeding_s"], d["value"]/1e9, d["value_from_raw_witness"]/1e9))
except Exception as e:
    print("B=$b failed", e)
PY
done
python tests/config_timings.py 2>/dev/null | grep '"config"' > gpurun_out/r2_config_timings.jsonl
cut -c1-100 gpurun_out/r2_config_timings.jsonl
head -16 gpurun_out/r2_kernel_trace.md
grep -E "k_witness_loop|k_check_prog" gpurun_out/pmc_r2_fetch.txt gpurun_out/pmc_r2_write.txt
