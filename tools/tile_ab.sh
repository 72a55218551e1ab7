#!/bin/bash
# GPU box: A/B of the loop store's lane tiling (ZKGL_STORE_TILE_LOG2) over fresh processes, with the clocks / power sampled beside it.
# One line per run: T, loop-kernel ms, resident step ms.  -> gpurun_out/tile_ab.txt, gpurun_out/tile_ab_smi.txt
mkdir -p gpurun_out
: > gpurun_out/tile_ab.txt
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' '|'; echo; sleep 1; done ) > gpurun_out/tile_ab_smi.txt &
SMI=$!
for rep in 1 2 3 4; do
  for t in ${TILES:-6 9 12}; do
    ZKGL_STORE_TILE_LOG2=$t timeout 600 python bench.py --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err
    python - "$t" >> gpurun_out/tile_ab.txt <<'PY'
import json, sys
d = json.load(open("/tmp/b.json"))
print("T", sys.argv[1], "loop_ms", round(d["roofline"]["avg_launch_ms"], 2), "resident_ms", round(d["config"]["ms_per_step_inputs_resident"], 2), "step_ms", round(d["ms_per_step"], 2))
PY
  done
done
kill $SMI
cat gpurun_out/tile_ab.txt
sort gpurun_out/tile_ab_smi.txt | uniq -c | sort -rn | head -12
