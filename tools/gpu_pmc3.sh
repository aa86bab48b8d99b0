bash tools/gpu_pmc.sh a "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" --variant 4 2>&1 | grep -v "^$" | tail -80
bash tools/gpu_pmc.sh b "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" --variant 4 2>&1 | grep -v "^$" | tail -70
bash tools/gpu_pmc.sh c "GRBM_GUI_ACTIVE TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" --variant 4 2>&1 | grep -v "^$" | tail -70
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c
