#!/bin/bash
# GPU box: quick parity run + bench at a few batch sizes (usage: bash tools/ab_bench.sh "64 128" [pytest args])
cd "$(dirname "$0")/.."
BATCHES=${1:-"64 128"}
python -m pytest ${2:-tests/test_gpu_main_vm.py tests/test_gpu_cs.py tests/test_gpu_full_size.py} -x -q -m gpu 2>&1 | tail -4
for b in $BATCHES; do
  python bench.py --steps 4 --batch $b ${BENCH_ARGS:-} --no-cpu-baseline > gpurun_out/ab_b$b.json 2> gpurun_out/ab_b$b.err || tail -3 gpurun_out/ab_b$b.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab_b$b.json"))
    print("B=$b", "step", round(d["ms_per_step"],2), "loop", round(d["roofline"]["avg_launch_ms"],2), "gates", round(d["roofline"]["other_kernels_ms"]["k_check_gates_loop"],2), "outer", round(d["roofline"]["other_kernels_ms"]["outer_post_and_checks_overlapped"],2), "seed", d["config"]["input_seeding_s"], d["config"]["commitments_equal_native_restatement"], "G constraints/s", round(d["value"]/1e9,2), "from raw", round(d["value_from_raw_witness"]/1e9,2), "stream", d["config"]["seeded_stream_instances_per_gpu"])
except Exception as e:
    print("B=$b failed", e)
PY
done
