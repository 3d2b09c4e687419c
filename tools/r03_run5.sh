#!/bin/bash
# round 3: stage kernels at frame scale (one slab of 640 000 rays), fp32 and bf16 frames
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3t
mkdir -p $O
run() { tag=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 2 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; rm -f $O/$tag.log; }
run frame_kernel_stats --kernel-trace --stats -d $O/frame_kernel_stats -o r -- python3 bench.py --steps 2 --warmup 1 --frames 2 --no-cpu-baseline --train-steps 0 --no-small-batch --no-bf16-leg
run frame_bf16_kernel_stats --kernel-trace --stats -d $O/frame_bf16_kernel_stats -o r -- python3 bench.py --bf16 --steps 2 --warmup 1 --frames 2 --no-cpu-baseline --train-steps 0 --no-small-batch
python3 - <<'PY'
import json
for f in ("frame_kernel_stats", "frame_bf16_kernel_stats"):
    d = json.load(open(f"gpurun_out/r3t/{f}.json"))
    print(f)
    for k in d["kernels"][:9]:
        print(f"  {k['launches']:4d} max {k['max_us']:10.1f} us  avg {k['avg_us']:10.1f}  {k['name'][:100]}")
PY
