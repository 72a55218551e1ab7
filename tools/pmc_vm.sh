#!/bin/bash
# PMC passes over one main_vm step (GPU box, repo root); each pass is its own rocprofv3 run (tools/pmc_pass.sh).
export BATCH=${B:-64}
bash tools/pmc_pass.sh vm_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU | grep -E "k_witness_loop|k_check_gates_c"
bash tools/pmc_pass.sh vm_sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM | grep -E "k_witness_loop|k_check_gates_c"
bash tools/pmc_pass.sh vm_tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum | grep -E "k_witness_loop|k_check_gates_c"
bash tools/pmc_pass.sh vm_tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum | grep -E "k_witness_loop|k_check_gates_c"
bash tools/pmc_pass.sh vm_fetch FETCH_SIZE | grep -E "k_witness|k_check_gates_c|k_seed"
bash tools/pmc_pass.sh vm_write WRITE_SIZE | grep -E "k_witness|k_check_gates_c|k_seed"
bash tools/pmc_pass.sh vm_lat TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum | grep -E "k_witness_loop|k_check_gates_c"
