#!/bin/bash
# A/B of wgrad_big_kernel variants on one box: shipped library vs build_scratch/libmi_nerf_wgsync.so (a --variant build),
# training-step time interleaved x3, then FETCH_SIZE of the weight-gradient launches for both
export TMPDIR=/tmp
mkdir -p gpurun_out/r3r
for i in 1 2 3; do for v in "" _wgsync; do echo "lib$v $(MI_NERF_LIB=$PWD/nerf_pytorch_paeng_amd/libmi_nerf$v.so PYTHONPATH=. python3 tools/train_probe.py 4096 10 2>&1 | tail -1)"; done; done > gpurun_out/r3r/ab_wgsync.txt
cat gpurun_out/r3r/ab_wgsync.txt
for v in "" _wgsync; do
MI_NERF_LIB=$PWD/nerf_pytorch_paeng_amd/libmi_nerf$v.so timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r3r/f -o r -- python3 tools/train_probe.py 4096 4 > gpurun_out/r3r/f.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/r3r/f/r_results.db --last 3 > gpurun_out/r3r/fetch$v.json; rm -rf gpurun_out/r3r/f
python3 -c "
import json
d=json.load(open('gpurun_out/r3r/fetch$v.json'))
for r in d['counters']:
    if 'wgrad_big' in r['kernel']: print('lib$v', r['duration_us'], r['FETCH_SIZE']*2*1024/1e9, 'GB')
" | tee -a gpurun_out/r3r/ab_wgsync.txt
done
