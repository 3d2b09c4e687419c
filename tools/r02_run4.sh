#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
B="python3 bench.py --bf16 --steps 5 --warmup 2 --frames 0 --no-cpu-baseline --no-small-batch"
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $O/pmc1 -o r -- $B > $O/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc2 -o r -- $B > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $O/pmc3 -o r -- $B > $O/pmc3.log 2>&1; echo "pmc3 rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM -d $O/pmc4 -o r -- $B > $O/pmc4.log 2>&1; echo "pmc4 rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats -o r -- $B > $O/stats.log 2>&1; echo "stats rc=$?"
