#!/bin/bash
# round 5: what bounds the stage kernels at frame size (640 000 rays)?  instruction mix and wait share of composite_kernel<3> / composite_fine_z_kernel<1> / stratified_kernel
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5s
mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --frames 2 --no-cpu-baseline --train-steps 0 --no-small-batch --no-bf16-leg --no-f16s-leg"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p1 -o r -- $B > $O/p1.log 2>&1; echo "pmc1 rc=$?"
python3 tools/rocpd_summary.py $O/p1/r_results.db --last 2 > $O/frame_stage_pmc1.json 2>>$O/p1.log; rm -rf $O/p1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p2 -o r -- $B > $O/p2.log 2>&1; echo "pmc2 rc=$?"
python3 tools/rocpd_summary.py $O/p2/r_results.db --last 2 > $O/frame_stage_pmc2.json 2>>$O/p2.log; rm -rf $O/p2
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p3 -o r -- $B > $O/p3.log 2>&1; echo "pmc3 rc=$?"
python3 tools/rocpd_summary.py $O/p3/r_results.db --last 2 > $O/frame_stage_pmc3.json 2>>$O/p3.log; rm -rf $O/p3
python3 - <<'PY'
import json
for f in ("frame_stage_pmc1", "frame_stage_pmc2", "frame_stage_pmc3"):
    d = json.load(open(f"gpurun_out/r5s/{f}.json"))
    for c in d["counters"]:
        if c["duration_us"] > 80 and "mlp_" not in c["kernel"]:
            print(f, {k: (v if not isinstance(v, float) else round(v, 1)) for k, v in c.items()})
PY
