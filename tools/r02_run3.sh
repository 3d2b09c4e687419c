#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
timeout -k 10 200 python tools/bf16_debug.py > $O/dbg.log 2>&1; echo "debug rc=$?"; grep "D=" $O/dbg.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16" -s > $O/t_bf16.log 2>&1; echo "bf16 tests rc=$?"; grep -E "bf16 vs|passed|failed|Error|error" $O/t_bf16.log | head
timeout -k 10 200 python bench.py --bf16 --frames 1 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?"; python -c "
import json; j=json.loads([l for l in open('$O/bench_bf16.json') if l.startswith('{')][0]); print(j['value'], j['ms_per_step'], j['frame_ms_800x800'], j['roofline']['kernel_ms'], j['roofline']['frac']); print([ (r['rays'], r['fine_kernel_frac']) for r in j['small_batch']])"
