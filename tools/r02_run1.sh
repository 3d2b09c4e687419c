#!/bin/bash
# round-2 GPU call 1: tests, bench lines (N=1, launcher rehearsal N=2), baseline evidence for the bf16 kernel
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
tail -3 $O/tests.log
timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --bf16 --frames 1 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 rc=$?"
BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --frames 1 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "bench n2 rc=$?"
B="python3 bench.py --bf16 --steps 5 --warmup 2 --frames 0 --no-cpu-baseline --no-small-batch"
cd /tmp 2>/dev/null; cd - >/dev/null
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/bf16_stats -o r -- $B > $O/bf16_stats.log 2>&1; echo "stats rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $O/bf16_pmc1 -o r -- $B > $O/bf16_pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/bf16_pmc2 -o r -- $B > $O/bf16_pmc2.log 2>&1; echo "pmc2 rc=$?"
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $O/bf16_pmc3 -o r -- $B > $O/bf16_pmc3.log 2>&1; echo "pmc3 rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC -d $O/bf16_pmc4 -o r -- $B > $O/bf16_pmc4.log 2>&1; echo "pmc4 rc=$?"
ls -R $O | head -50
