#!/bin/bash
# round 5: PMC passes of the bf16 kernel on the final build (round 3's counters; the kernel's object code is unchanged since round 4's wait-state fixes)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r5y
mkdir -p $O
run() { tag=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 3 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; rm -f $O/$tag.log; }
B16="python3 bench.py --bf16 --steps 10 --warmup 2 --no-cpu-baseline --frames 0 --train-steps 0 --no-small-batch"
run bf16_pmc1 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d $O/bf16_pmc1 -o r -- $B16
run bf16_pmc2 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $O/bf16_pmc2 -o r -- $B16
run bf16_pmc3 --kernel-trace --pmc TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/bf16_pmc3 -o r -- $B16
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r5y/bf16_pmc1.json"))
for c in d["counters"]:
    if "mlp_bf16_kernel" in c["kernel"] and c["duration_us"] > 400:
        busy, gui, dur = c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"], c["duration_us"]
        print(f"fine launch {dur:.1f} us: MFMA busy {busy / (gui / 8 * 1024) * 100:.1f} % of SIMD cycles, clock {gui / 8 / dur / 1e3:.3f} GHz, wait {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'] * 100:.1f} %")
d = json.load(open("gpurun_out/r5y/bf16_pmc3.json"))
for c in d["counters"]:
    if "mlp_bf16_kernel" in c["kernel"] and c["duration_us"] > 400:
        print(f"fine launch {c['duration_us']:.1f} us: TCC_READ {c['TCC_READ_sum']:.4g} x 128 B = {c['TCC_READ_sum'] * 128 / 1e9:.2f} GB, hit {c['TCC_HIT_sum']:.4g}, miss {c['TCC_MISS_sum']:.4g}")
PY
