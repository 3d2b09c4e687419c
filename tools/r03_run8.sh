#!/bin/bash
# round 3: PMC passes of the split-precision training step's kernels (tools/train_f16s_probe.py: fp32 and f16s steps interleaved)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3v
mkdir -p $O
run() { tag=$1; shift; timeout -k 10 240 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 2 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; rm -f $O/$tag.log; }
B="python3 tools/train_f16s_probe.py 4096 1"
run train_f16s_pmc1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/train_f16s_pmc1 -o r -- $B
run train_f16s_pmc_fetch --kernel-trace --pmc FETCH_SIZE -d $O/train_f16s_pmc_fetch -o r -- $B
ls $O
