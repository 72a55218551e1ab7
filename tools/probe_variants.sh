#!/bin/bash
# container: the ELIMINATION-PROBE libraries of the witness interpreter (kernels_engine2.hpp `namespace probe`: -DZKGL_EXPERIMENT=<mask> leaves parts of
# the kernel out — WRONG results, timing only) under the names tools/loop_probe.sh / tools/loop_insts.sh expect: era-zkevm_circuits_amd/libzkgl_var_<TAG>.so
#   bits: 1 no operand loads, 2 no stores, 4 no S-box multiplications, 8 no linear layers, 16 no FMA products, 32 no inversions, 64 no table search, 128 no multiplicity atomics
# These are measurement tools of one session; they are not committed and nothing in the product loads them.
cd "$(dirname "$0")/.."
declare -A MASK=( [L]=1 [S]=2 [SL]=3 [M]=128 [SLM]=131 [P2]=4 [P2LIN]=8 [FMA]=16 [INV]=32 [FIND]=64 [ALLV]=252 )
args=()
for t in ${VARIANTS:-M S L SL SLM INV P2 P2LIN FMA FIND ALLV}; do args+=("$t=-DZKGL_EXPERIMENT=${MASK[$t]}"); done
bash tools/variants.sh "${args[@]}"
