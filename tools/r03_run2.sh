#!/bin/bash
# round 3: tools/mfma_probe6.bin (operand paths of the bf16 weight stream) -- timings, then two PMC passes (clock; L2 reads)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r3o
mkdir -p $O
timeout -k 10 120 ./tools/mfma_probe6.bin > $O/probe6.txt 2>&1 || exit 1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $O/p1 -o r -- ./tools/mfma_probe6.bin > $O/p1.log 2>&1
echo "pmc1 rc=$?"
python3 tools/rocpd_summary.py $O/p1/r_results.db --last 2 > $O/probe6_pmc1.json 2>>$O/p1.log
rm -rf $O/p1
timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/p2 -o r -- ./tools/mfma_probe6.bin > $O/p2.log 2>&1
echo "pmc2 rc=$?"
python3 tools/rocpd_summary.py $O/p2/r_results.db --last 2 > $O/probe6_pmc2.json 2>>$O/p2.log
rm -rf $O/p2
cat $O/probe6.txt | tail -8
python3 - <<'PY'
import json
for f in ("gpurun_out/r3o/probe6_pmc1.json", "gpurun_out/r3o/probe6_pmc2.json"):
    d = json.load(open(f))
    for c in d["counters"]:
        print({k: (v if not isinstance(v, float) else round(v, 1)) for k, v in c.items()})
PY
