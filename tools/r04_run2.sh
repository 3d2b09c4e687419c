#!/bin/bash
# round 4 bench lines on the FINAL build: the driver's form, fern (config #4), bf16 as the main line (config #5), a 4-rank gloo rehearsal (the
# `collective` block), rank 3 of 8 alone
set -o pipefail
O=gpurun_out/r4q
mkdir -p $O
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "n1 rc=$?"
timeout -k 10 300 python bench.py --workload fern --no-cpu-baseline --train-steps 0 > $O/bench_n1_fern.json 2> $O/bench_fern.err; echo "fern rc=$?"
timeout -k 10 300 python bench.py --bf16 --no-cpu-baseline --train-steps 0 > $O/bench_n1_bf16.json 2> $O/bench_bf16.err; echo "bf16 rc=$?"
BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 4 --steps 10 --warmup 2 --frames 2 --no-cpu-baseline > $O/bench_n4_gloo_rehearsal.json 2> $O/bench_n4.err; echo "n4 rc=$?"
BENCH_SOLO_RANK=1 RANK=3 LOCAL_RANK=0 WORLD_SIZE=8 timeout -k 10 300 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rank3_of_8_alone.json 2> $O/bench_solo.err; echo "solo rc=$?"
tail -c 600 $O/bench_n1.json
