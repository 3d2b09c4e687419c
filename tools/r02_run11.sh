#!/bin/bash
# round-2 final: HBM traffic of the training kernels on the final build (separate FETCH_SIZE / WRITE_SIZE passes) + instruction mix
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3p
mkdir -p $O
run() { tag=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 3 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; rm -f $O/$tag.log; }
T="python3 tools/train_probe.py 4096 4"
run train_final_fetch --kernel-trace --pmc FETCH_SIZE -d $O/train_final_fetch -o r -- $T
run train_final_write --kernel-trace --pmc WRITE_SIZE -d $O/train_final_write -o r -- $T
run train_final_mix --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS -d $O/train_final_mix -o r -- $T
ls $O
