#!/bin/bash
# round 3: kernel stats of the training step with the fp32 and with the split-precision forward (tools/train_f16s_probe.py)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3u
mkdir -p $O
run() { tag=$1; shift; timeout -k 10 240 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 3 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; rm -f $O/$tag.log; }
run train_f16s_kernel_stats --kernel-trace --stats -d $O/train_f16s_kernel_stats -o r -- python3 tools/train_f16s_probe.py 4096 4
run train_f16s_pmc_write --kernel-trace --pmc WRITE_SIZE -d $O/train_f16s_pmc_write -o r -- python3 tools/train_f16s_probe.py 4096 1
ls $O
