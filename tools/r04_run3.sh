#!/bin/bash
# round 4: rocprofv3 --kernel-trace --stats of the DRIVER'S OWN command (python3 bench.py --gpus 1 --steps 20 --warmup 5), final build
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r4r
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/drv -o r -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv.log 2>&1; echo "rc=$?"
python3 tools/rocpd_summary.py $O/drv/r_results.db --last 3 --clusters > $O/driver_command_kernel_stats.json 2>>$O/drv.log
grep "^{" $O/drv.log | tail -1 > $O/driver_command_bench_line.json
rm -rf $O/drv
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4r/driver_command_kernel_stats.json"))
for k in d["kernels"][:3]:
    print(f"{k['launches']:5d} avg {k['avg_us']:9.1f} med {k['median_us']:9.1f} max {k['max_us']:9.1f}  {k['name'][:90]}")
    for c in k.get("clusters", []):
        print("        ", c)
l = json.load(open("gpurun_out/r4r/driver_command_bench_line.json"))
print(l["value"], l["roofline"]["kernel_ms"], l["roofline"]["frac"])
PY
