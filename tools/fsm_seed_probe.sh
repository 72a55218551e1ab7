#!/bin/bash
# GPU box, repo root: where the time of k_fsm_seed goes (ZKGL_FSM_SEED_DEBUG: 1 = no hashing, 2 = no walking, 3 = IO only)
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 3; do
  rm -rf /tmp/kt$d
  ZKGL_FSM_SEED_DEBUG=$d CONFIGS=${CONFIGS:-C3k,C3s} timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt$d -o kt -- python $ROOT/tests/config_timings.py > /tmp/o$d.txt 2>/tmp/e$d.txt
  db=$(find /tmp/kt$d -name "*_results.db" | head -1)
  echo "debug=$d"; python $ROOT/profiles/summarize_rocpd.py $db | grep -E "k_fsm_seed|k_storage|k_logsort|k_eip"
done
