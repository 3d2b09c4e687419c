#!/bin/bash
# round-2 final evidence for the shipped bf16 kernel (v_mfma_f32_16x16x32_bf16 build): kernel stats + three PMC passes of `bench.py --bf16`
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3i
mkdir -p $O
B="python3 bench.py --bf16 --steps 10 --warmup 2 --no-cpu-baseline --frames 0 --train-steps 0 --no-small-batch"
run() { tag=$1; shift; timeout -k 10 240 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 3 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; rm -f $O/$tag.log; }
run bf16_final_stats --kernel-trace --stats -d $O/bf16_final_stats -o r -- $B
run bf16_final_pmc1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $O/bf16_final_pmc1 -o r -- $B
run bf16_final_pmc2 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $O/bf16_final_pmc2 -o r -- $B
run bf16_final_pmc3 --kernel-trace --pmc TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/bf16_final_pmc3 -o r -- $B
ls $O
