#!/bin/bash
# round-2: instruction mix + cycles of the training kernels and the stand-alone weight-gradient product (after the VALU-free wgrad loop)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2t
mkdir -p $O
run() { tag=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 3 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; tail -c 1500 $O/$tag.log > $O/$tag.log.tail; rm -f $O/$tag.log; }
T="python3 tools/train_probe.py 4096 4"
run train_mix --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS -d $O/train_mix -o r -- $T
W="python3 tools/wgrad_probe.py"
run wgrad_mix --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS -d $O/wgrad_mix -o r -- $W
ls $O
