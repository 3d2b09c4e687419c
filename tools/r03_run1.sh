#!/bin/bash
# round 3: kernel timeline of a 512-ray bf16 step (jitter drawn inside the step, as bench.py does)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3h
mkdir -p $O
PYTHONPATH=. python3 tools/step_timeline.py --bf16-only --jitter 256 512 1024 4096 > $O/step_wall.txt 2>&1
cat $O/step_wall.txt
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tl -o r -- python3 tools/step_timeline.py --bf16-only --jitter 512 > $O/tl.log 2>&1
echo "rc=$?"
python3 tools/rocpd_summary.py $O/tl/r_results.db --timeline 14 > $O/timeline_512.json 2>>$O/tl.log
rm -rf $O/tl
python3 -c "
import json
d=json.load(open('$O/timeline_512.json'))
for r in d['timeline']: print(r)
"
