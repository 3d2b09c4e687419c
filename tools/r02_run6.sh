#!/bin/bash
# round-2: per-dispatch clock / wait counters of tools/mfma_probe3.bin (zero vs random operands, direct vs LDS-staged loads)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2q
mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM -d $O/p3 -o r -- ./tools/mfma_probe3.bin > $O/p3.log 2>&1
echo "rc=$?"
python3 tools/rocpd_summary.py $O/p3/r_results.db --last 40 > $O/probe3_pmc.json 2>>$O/p3.log
rm -rf $O/p3
tail -c 3000 $O/p3.log
