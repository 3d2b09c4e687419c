#!/bin/bash
# round 3: kernel timeline of the split-precision training step (the last dispatches of tools/train_f16s_probe.py are f16s steps)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3w
mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace -d $O/tl -o r -- python3 tools/train_f16s_probe.py 4096 2 > $O/tl.log 2>&1; echo "rc=$?"
python3 tools/rocpd_summary.py $O/tl/r_results.db --timeline 260 > $O/train_f16s_timeline.json 2>>$O/tl.log
rm -rf $O/tl
