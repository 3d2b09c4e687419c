#!/bin/bash
# round-2 GPU call 2: first run of the rebuilt bf16 kernel
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16" -s > $O/t_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -15 $O/t_bf16.log
timeout -k 10 200 python bench.py --bf16 --frames 1 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?"; tail -c 1500 $O/bench_bf16.json; tail -3 $O/bench_bf16.err
