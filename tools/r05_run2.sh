#!/bin/bash
# round 5 evidence on the FINAL build: kernel stats + the four PMC passes of the shipped fp32 inference kernel (profiles/traffic.json), the MFMA-busy pass of the
# 512-wide kernel, kernel stats of the bf16 / f16s legs and of the training step, and the DRIVER'S OWN command under --kernel-trace
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5p
mkdir -p $O
run() { tag=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 3 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; rm -f $O/$tag.log; }
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames 0 --train-steps 0 --no-small-batch --no-bf16-leg --no-f16s-leg"
run fp32_kernel_stats --kernel-trace --stats -d $O/fp32_kernel_stats -o r -- $B
run fp32_pmc1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/fp32_pmc1 -o r -- $B
run fp32_pmc2 --kernel-trace --pmc FETCH_SIZE -d $O/fp32_pmc2 -o r -- $B
run fp32_pmc3 --kernel-trace --pmc WRITE_SIZE -d $O/fp32_pmc3 -o r -- $B
run fp32_pmc4 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA -d $O/fp32_pmc4 -o r -- $B
python3 -c "import bench; print('build', bench.kernel_build_id())" > $O/build_id.txt
run wide_kernel_stats --kernel-trace --stats -d $O/wide_kernel_stats -o r -- python3 tools/wide_probe.py 4096
run wide_pmc1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT -d $O/wide_pmc1 -o r -- python3 tools/wide_probe.py 4096
run wide_pmc2 --kernel-trace --pmc FETCH_SIZE -d $O/wide_pmc2 -o r -- python3 tools/wide_probe.py 4096
run bf16_kernel_stats --kernel-trace --stats -d $O/bf16_kernel_stats -o r -- python3 bench.py --bf16 --steps 10 --warmup 2 --no-cpu-baseline --frames 0 --train-steps 0 --no-small-batch
run f16s_kernel_stats --kernel-trace --stats -d $O/f16s_kernel_stats -o r -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames 0 --train-steps 0 --no-small-batch --no-bf16-leg
run train_kernel_stats --kernel-trace --stats -d $O/train_kernel_stats -o r -- python3 tools/train_probe.py 4096 6
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/drv -o r -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv.log 2>&1; echo "driver rc=$?"
python3 tools/rocpd_summary.py $O/drv/r_results.db --last 3 --clusters > $O/driver_command_kernel_stats.json 2>>$O/drv.log
grep "^{" $O/drv.log | tail -1 > $O/driver_command_bench_line.json
rm -rf $O/drv
ls $O
