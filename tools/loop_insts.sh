#!/bin/bash
# GPU box: VALU / SALU instruction counts of zke::k_witness_loop per stub variant (tools/variants.sh) -> gpurun_out/loop_insts.txt
# per-SE means as rocprofv3 reports them; divide by the wavefronts of an SE (SQ_WAVES) for per-wavefront counts
ROOT=$(pwd); mkdir -p gpurun_out; : > gpurun_out/loop_insts.txt
for t in ${VARIANTS:-full ALLV P2 P2LIN FMA FIND M INV}; do
  lib=$ROOT/era-zkevm_circuits_amd/libzkgl_var_$t.so; [ $t = full ] && lib=$ROOT/era-zkevm_circuits_amd/libzkgl.so
  [ -f $lib ] || continue
  export PMC_CMD="env ZKGL_STUB_RUN=1 ZKGL_LIB=$lib python $ROOT/bench.py --batch ${B:-384} --seed-windows 2 --steps 2 --warmup 0 --no-cpu-baseline"
  tools/pmc_pass.sh li_$t SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS > /dev/null
  grep "k_witness_loop " gpurun_out/pmc_li_$t.txt | awk -v t=$t '{printf "%-6s %-20s %s\n", t, $2, $4}' >> gpurun_out/loop_insts.txt
done
cat gpurun_out/loop_insts.txt
