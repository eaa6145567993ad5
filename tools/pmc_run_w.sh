#!/bin/bash
# usage: tools/pmc_run_w.sh <tag> <pmc_wgrad args...>
tag=$1; shift
export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmcw_${tag}_a -- python3 tools/pmc_wgrad.py "$@" > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmcw_${tag}_b -- python3 tools/pmc_wgrad.py "$@" > /dev/null 2>&1
