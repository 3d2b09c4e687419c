#!/bin/bash
# round 6, final build (traffic.json re-tied to it): the driver's own command under --kernel-trace, and the three plain bench lines
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6r
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/drv -o r -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv.log 2>&1; echo "driver rc=$?"
python3 tools/rocpd_summary.py $O/drv/r_results.db --last 3 --clusters > $O/driver_command_kernel_stats.json 2>>$O/drv.log
grep "^{" $O/drv.log | tail -1 > $O/driver_command_bench_line.json
rm -rf $O/drv
timeout -k 10 250 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout -k 10 200 python3 bench.py --bf16 --no-cpu-baseline > $O/bench_n1_bf16.json 2> $O/bf16.err; echo "bf16 rc=$?"
timeout -k 10 200 python3 bench.py --workload fern --no-cpu-baseline > $O/bench_n1_fern.json 2> $O/fern.err; echo "fern rc=$?"
