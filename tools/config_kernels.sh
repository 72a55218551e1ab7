#!/bin/bash
# GPU box, repo root: per-kernel tables of the other BASELINE configurations -> gpurun_out/r3_config_kernels.md (OUT_NAME overrides)
# (rocprofv3 --kernel-trace --stats of `CONFIGS=<c> python tests/config_timings.py`, one run per configuration)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${OUT_NAME:-r3_config_kernels.md}; : > $OUT
cd /tmp && export TMPDIR=/tmp
for c in ${CFGS:-C1 C3k C3s C4s C4l C5}; do
  rm -rf /tmp/ck_$c
  CONFIGS=$c timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ck_$c -o kt -- python "$ROOT/tests/config_timings.py" < /dev/null > /tmp/ck_$c.out 2> /tmp/ck_$c.err
  db=$(find /tmp/ck_$c -name "*_results.db" | head -1)
  echo "### CONFIGS=$c python tests/config_timings.py under rocprofv3 --kernel-trace --stats" >> $OUT
  grep '"config"' /tmp/ck_$c.out | cut -c1-400 >> $OUT
  [ -n "$db" ] && python "$ROOT/profiles/summarize_rocpd.py" "$db" | head -14 >> $OUT
  echo >> $OUT
done
grep -c "^###" $OUT
