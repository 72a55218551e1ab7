export PMC_CMD="env CONFIGS=C3k python $PWD/tests/config_timings.py"
tools/pmc_pass.sh st_a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM > /dev/null
tools/pmc_pass.sh st_b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU > /dev/null
tools/pmc_pass.sh st_c GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR > /dev/null
tools/pmc_pass.sh st_d FETCH_SIZE WRITE_SIZE > /dev/null
grep -h "k_witness_strands2" gpurun_out/pmc_st_a.txt gpurun_out/pmc_st_b.txt gpurun_out/pmc_st_c.txt gpurun_out/pmc_st_d.txt | cut -c1-30,60-200
