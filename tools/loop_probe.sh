#!/bin/bash
# GPU box, repo root: what bounds zke::k_witness_loop at the bench batch?  -> gpurun_out/loop_probe_*.txt
#  1. elimination: stubbed-store / stubbed-load / no-multiplicity variants of the kernel (WRONG results, timing only; built by
#     tools/variants.sh S=.. L=.. SL=.. SLM=.. M=..)
#  2. SQ counters of the real kernel: VALU / SALU / VMEM instruction counts and busy cycles against GRBM_GUI_ACTIVE
#  3. the effective shader clock of the launch: GRBM_GUI_ACTIVE / kernel duration
set -u
ROOT=$(pwd); mkdir -p gpurun_out
B=${B:-384}
one() {
  ZKGL_STUB_RUN=1 ZKGL_LIB=$2 timeout 600 python bench.py --steps 5 --warmup 1 --batch $B --no-cpu-baseline 2>gpurun_out/stub_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-6s step(resident) %.2f ms  k_witness_loop %.2f ms @ %.0f MHz  loop check %.2f  outer %.2f' % ('$1', d['config']['ms_per_step_inputs_resident'], d['roofline']['avg_launch_ms'], d['roofline']['shader_clock_mhz'], d['roofline']['other_kernels_ms']['k_check_gates_loop'], d['roofline']['other_kernels_ms']['outer_post_and_checks_overlapped']))"
}
{
one full "$ROOT/era-zkevm_circuits_amd/libzkgl.so"
for t in ${VARIANTS:-M S L SL SLM INV P2 P2LIN FMA FIND ALLV}; do [ -f era-zkevm_circuits_amd/libzkgl_var_$t.so ] && one "-$t" "$ROOT/era-zkevm_circuits_amd/libzkgl_var_$t.so"; done
one full "$ROOT/era-zkevm_circuits_amd/libzkgl.so"
} > gpurun_out/loop_probe_stubs.txt 2>&1
cat gpurun_out/loop_probe_stubs.txt
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ|GRBM|TCC|TCP|TA|TD)_[A-Z0-9_]+" | sort -u | tr '\n' ' ') > gpurun_out/loop_probe_counters_avail.txt
[ -n "${NO_PMC:-}" ] && exit 0
export PMC_CMD="python $ROOT/bench.py --batch $B --seed-windows 2 --steps 2 --warmup 0 --no-cpu-baseline"
tools/pmc_pass.sh lp_a SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE > /dev/null
tools/pmc_pass.sh lp_b SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVE_CYCLES > /dev/null
tools/pmc_pass.sh lp_c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM > /dev/null
tools/pmc_pass.sh lp_d SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY > /dev/null
grep -h "k_witness_loop" gpurun_out/pmc_lp_a.txt gpurun_out/pmc_lp_b.txt gpurun_out/pmc_lp_c.txt gpurun_out/pmc_lp_d.txt > gpurun_out/loop_probe_sq.txt
cat gpurun_out/loop_probe_sq.txt
