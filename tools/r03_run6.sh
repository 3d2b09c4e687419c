#!/bin/bash
# round 3: kernel stats + three PMC passes of the split-precision kernel (mlp_f16s_kernel) on BASELINE config #2's batch
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3s
mkdir -p $O
run() { tag=$1; shift; timeout -k 10 240 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 3 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; rm -f $O/$tag.log; }
B="python3 tools/f16s_run.py 4096 10 10"
run f16s_kernel_stats --kernel-trace --stats -d $O/f16s_kernel_stats -o r -- $B
run f16s_pmc1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $O/f16s_pmc1 -o r -- $B
run f16s_pmc2 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $O/f16s_pmc2 -o r -- $B
run f16s_pmc3 --kernel-trace --pmc TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/f16s_pmc3 -o r -- $B
ls $O
