#!/bin/bash
# What the training forward pays for its stash: STASH kernel time (rocprofv3 kernel stats, fine-net launches) of the shipped library and of
# (apply tools/stash_ablation.patch first: it adds the MN_STASH_NOMASK / MN_STASH_NOSTORE switches to mlp_fp32.hip; the shipped source
# does not carry them so that profiles/traffic.json stays tied to it)
# ablation builds without the ReLU' mask packing, without the activation row stores, and without both (timing only: results are garbage)
export TMPDIR=/tmp
O=gpurun_out/r3y
mkdir -p $O
for v in "" _nomask _nostore _neither; do
  MI_NERF_LIB=$PWD/nerf_pytorch_paeng_amd/libmi_nerf$v.so timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/s -o r -- python3 tools/train_probe.py 4096 6 > $O/s.log 2>&1
  python3 tools/rocpd_summary.py $O/s/r_results.db > $O/stats$v.json; rm -rf $O/s
  python3 -c "
import json
d=json.load(open('$O/stats$v.json'))
for k in d['kernels']:
    if 'mlp_fp32_kernel' in k['name'] or 'dgrad' in k['name'] or 'wgrad_big' in k['name']: print('lib$v', k['name'][:60], 'launches', k['launches'], 'avg_us', k['avg_us'], 'max_us', k['max_us'])
" | tee -a $O/stash_ablation.txt
done
