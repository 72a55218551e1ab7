#!/bin/bash
# GPU box: per-kernel averages (rocprofv3 --kernel-trace) of the step for the real library, env-switched variants and side-by-side
# libraries, on ONE box.  usage: B=384 bash tools/kt_ab.sh "TAG:ENV=VAL ..." ...   (TAG:ZKGL_LIB=path for another library)
ROOT=$(pwd); B=${B:-384}
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  rm -rf /tmp/kt_$tag
  env $envs timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt -- python "$ROOT/bench.py" --steps 3 --warmup 1 --batch $B --seed-windows 1 --no-cpu-baseline < /dev/null > /tmp/kt_$tag.json 2> /tmp/kt_$tag.err
  db=$(find /tmp/kt_$tag -name "*_results.db" | head -1)
  echo "== $tag ($envs): $(python -c "import json;d=json.load(open('/tmp/kt_$tag.json'));print('step %.2f ms' % d['ms_per_step'])" 2>/dev/null)"
  grep -h "loop store at" /tmp/kt_$tag.err | tail -1
  [ -n "$db" ] && python "$ROOT/profiles/summarize_rocpd.py" "$db" | grep -E "k_witness_loop|k_check_prog|k_check_p2|k_witness_outer" | cut -d'|' -f2-8
done
