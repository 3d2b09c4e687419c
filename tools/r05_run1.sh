#!/bin/bash
# round 5, VERDICT item 3: the bf16 step at 512 rays (one of eight ranks of config #5) -- timing bound + kernel trace of the same step
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
timeout -k 10 400 python3 tools/bf16_512_bound_probe.py 512 256 1024 2048 > $O/bound.txt 2> $O/bound.err; echo "bound rc=$?"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $O/tl -o r -- python3 tools/step_timeline.py --bf16-only --jitter 512 > $O/tl.log 2>&1; echo "trace rc=$?"
python3 tools/rocpd_summary.py $O/tl/r_results.db --last 3 > $O/timeline_512_bf16.json 2>>$O/tl.log
rm -rf $O/tl
cat $O/bound.txt
