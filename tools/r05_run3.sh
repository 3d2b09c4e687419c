#!/bin/bash
# round 5: the stage kernels at the size a FRAME gives them (one slab of 640 000 rays): kernel trace of bench.py --frames 2
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5q
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/fr -o r -- python3 bench.py --steps 2 --warmup 1 --frames 2 --no-cpu-baseline --train-steps 0 --no-small-batch --no-bf16-leg --no-f16s-leg > $O/fr.log 2>&1; echo "rc=$?"
python3 tools/rocpd_summary.py $O/fr/r_results.db --last 3 --clusters > $O/frame_kernel_stats.json 2>>$O/fr.log
rm -rf $O/fr
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5q/frame_kernel_stats.json"))
for k in d["kernels"][:8]:
    print(f"{k['launches']:5d} avg {k['avg_us']:9.1f} med {k['median_us']:9.1f} max {k['max_us']:9.1f}  {k['name'][:90]}")
    for c in k.get("clusters", [])[-2:]:
        print("        ", c)
PY
