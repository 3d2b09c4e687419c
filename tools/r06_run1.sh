#!/bin/bash
# round 6 evidence on one MI355X: (1) `bench.py --gpus 4` with 4 LIVE ranks sharing the card (gloo process group, tests/c_abi/fake_rccl.cpp in librccl's place): ONE run that
# times the frame through both gather routes; (2) the driver's own command under --kernel-trace; (3) the four PMC passes of the shipped fp32 inference kernel;
# (4) FETCH / WRITE passes of the global-batch staging kernels (the materialised shuffle's HBM requests); (5) counters of the 512-ray bf16 step's two network launches
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r6p
mkdir -p $O
hipcc -shared -fPIC -O1 -x hip --offload-arch=gfx950 tests/c_abi/fake_rccl.cpp -o /tmp/libfake_rccl.so -lrt || exit 1
for WL in lego fern; do
  BENCH_BACKEND=gloo MI_NERF_RCCL_LIB=/tmp/libfake_rccl.so HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python3 bench.py --gpus 4 --steps 10 --warmup 3 --no-cpu-baseline --no-f16s-leg --workload $WL \
    > $O/bench_n4_$WL.json 2> $O/bench_n4_$WL.err; echo "n4 $WL rc=$?"
done
BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-small-batch --train-steps 0 --no-f16s-leg > $O/bench_one_rank_rccl.json 2> $O/bench_one_rank_rccl.err; echo "one-rank rccl rc=$?"
run() { tag=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 3 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; rm -f $O/$tag.log; }
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/drv -o r -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv.log 2>&1; echo "driver rc=$?"
python3 tools/rocpd_summary.py $O/drv/r_results.db --last 3 --clusters > $O/driver_command_kernel_stats.json 2>>$O/drv.log
grep "^{" $O/drv.log | tail -1 > $O/driver_command_bench_line.json
rm -rf $O/drv
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames 0 --train-steps 0 --no-small-batch --no-bf16-leg --no-f16s-leg"
run fp32_kernel_stats --kernel-trace --stats -d $O/fp32_kernel_stats -o r -- $B
run fp32_pmc1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/fp32_pmc1 -o r -- $B
run fp32_pmc2 --kernel-trace --pmc FETCH_SIZE -d $O/fp32_pmc2 -o r -- $B
run fp32_pmc3 --kernel-trace --pmc WRITE_SIZE -d $O/fp32_pmc3 -o r -- $B
run fp32_pmc4 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA -d $O/fp32_pmc4 -o r -- $B
python3 -c "import bench; print('build', bench.kernel_build_id())" > $O/build_id.txt
run staging_stats --kernel-trace --stats -d $O/staging_stats -o r -- python3 tools/staging_probe.py
run staging_pmc_fetch --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/staging_pmc_fetch -o r -- python3 tools/staging_probe.py
run staging_pmc_write --kernel-trace --pmc WRITE_SIZE -d $O/staging_pmc_write -o r -- python3 tools/staging_probe.py
S="python3 tools/step_timeline.py --bf16-only --jitter 512"
run bf16_512_stats --kernel-trace --stats -d $O/bf16_512_stats -o r -- $S
run bf16_512_pmc1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA -d $O/bf16_512_pmc1 -o r -- $S
run bf16_512_pmc2 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_LDS_ADDR_CONFLICT -d $O/bf16_512_pmc2 -o r -- $S
ls $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_trained.py -m gpu -q -s -x > $O/trained.log 2>&1; echo "trained rc=$?"
grep "^\[" $O/trained.log > $O/trained_weights.txt
