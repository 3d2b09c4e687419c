#!/bin/bash
# round-2: filler-cost probe beside fp32 MFMAs (tools/mfma_probe5.hip), cycle efficiency per dispatch from a PMC pass
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r2r
mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU -d $O/p5 -o r -- ./tools/mfma_probe5.bin > $O/p5.log 2>&1
echo "rc=$?"
python3 tools/rocpd_summary.py $O/p5/r_results.db --last 40 > $O/probe5_pmc.json 2>>$O/p5.log
rm -rf $O/p5
grep variant $O/p5.log
