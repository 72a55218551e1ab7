#!/bin/bash
# GPU box, repo root: the evidence set of a round (TAG=r5) -> gpurun_out/${TAG}_* (copy what is judged into profiles/).
#  1. rocprofv3 --kernel-trace --stats of the default bench command            -> ${TAG}_kernel_trace.md, ${TAG}_bench_under_rocprof.json
#  2. PMC passes (their own runs): FETCH_SIZE, WRITE_SIZE at the default batch   -> pmc_${TAG}_fetch.txt, pmc_${TAG}_write.txt
#  3. the bench itself, no profiler                                             -> ${TAG}_bench.json
# EXTRA_BENCH_ARGS (e.g. --narrow-store) goes to every bench.py invocation above: the profile of another mode of the same step (TAG=r6n)
set -u
X=${EXTRA_BENCH_ARGS:-}
TAG=${TAG:-r5}
ROOT=$(pwd); mkdir -p "$ROOT/gpurun_out"
B=${B:-384}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_${TAG}
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_${TAG} -o kt -- python "$ROOT/bench.py" --steps ${KT_STEPS:-5} --warmup 1 --no-cpu-baseline --headline-only $X < /dev/null > "$ROOT/gpurun_out/${TAG}_bench_under_rocprof.json" 2> /tmp/kt_${TAG}.err
db=$(find /tmp/kt_${TAG} -name "*_results.db" | head -1)
[ -n "$db" ] && python "$ROOT/profiles/summarize_rocpd.py" "$db" > "$ROOT/gpurun_out/${TAG}_kernel_trace.md"
cd "$ROOT"
export PMC_CMD="python $ROOT/bench.py --batch $B --seed-windows 2 --steps 2 --warmup 0 --no-cpu-baseline --headline-only $X"
tools/pmc_pass.sh ${TAG}_fetch FETCH_SIZE > /dev/null
tools/pmc_pass.sh ${TAG}_write WRITE_SIZE > /dev/null
# VALU-issue side (its own passes): instructions and busy cycles of the vector ALUs against the GPU's active cycles
tools/pmc_pass.sh ${TAG}_valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > /dev/null
tools/pmc_pass.sh ${TAG}_salu SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS > /dev/null
unset PMC_CMD
timeout 900 python bench.py ${BENCH_ARGS:-} $X < /dev/null > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || tail -3 gpurun_out/${TAG}_bench.err
head -24 gpurun_out/${TAG}_kernel_trace.md
grep -E "k_witness_loop|k_check_prog|k_check_p2|k_vm_" gpurun_out/pmc_${TAG}_fetch.txt gpurun_out/pmc_${TAG}_write.txt
