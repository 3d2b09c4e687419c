#!/bin/bash
# round 6: FETCH / WRITE passes of the global-batch staging kernels (rays_rgb_kernel, gather_rows_kernel as a whole-table shuffle): what the materialised shuffle moves through HBM
set -o pipefail
export TMPDIR=/tmp PYTHONPATH=.
O=gpurun_out/r6q
mkdir -p $O
run() { tag=$1; shift; timeout -k 10 300 rocprofv3 "$@" > $O/$tag.log 2>&1; echo "$tag rc=$?"; python3 tools/rocpd_summary.py $O/$tag/r_results.db --last 2 > $O/$tag.json 2>>$O/$tag.log; rm -rf $O/$tag; tail -3 $O/$tag.log; }
run staging_stats --kernel-trace --stats -d $O/staging_stats -o r -- python3 tools/staging_probe.py
run staging_pmc_fetch --kernel-trace --pmc FETCH_SIZE -d $O/staging_pmc_fetch -o r -- python3 tools/staging_probe.py
run staging_pmc_write --kernel-trace --pmc WRITE_SIZE -d $O/staging_pmc_write -o r -- python3 tools/staging_probe.py
run staging_pmc_req --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum -d $O/staging_pmc_req -o r -- python3 tools/staging_probe.py
ls $O
