#!/bin/bash
# round 5: what ONE rank of an N-GPU run executes, alone on the one GPU, for N = 1, 2, 4, 8 (BENCH_SOLO_RANK: no process group, this rank's shard of the 4096-ray
# batch and its rows of the frame): the strong-scaling table the driver's sweep would produce if every rank took as long as this one and the gather were free
set -o pipefail
O=gpurun_out/r5sc
mkdir -p $O
for N in 1 2 4 8; do
  R=$((N / 2))
  if [ $N -eq 1 ]; then
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 --no-f16s-leg > $O/n$N.json 2> $O/n$N.err
  else
    BENCH_SOLO_RANK=1 RANK=$R WORLD_SIZE=$N timeout -k 10 300 python3 bench.py --gpus $N --steps 20 --warmup 5 --no-cpu-baseline --no-f16s-leg > $O/n$N.json 2> $O/n$N.err
  fi
  echo "N=$N rc=$?"
done
python3 - <<'PY'
import json
base = None
print("# N   rays/rank  fp32 ms/step  fp32 rays/s (all ranks)  eff.   bf16 ms/step  bf16 rays/s   eff.   frame rows/rank  frame ms")
for N in (1, 2, 4, 8):
    l = json.loads([x for x in open(f"gpurun_out/r5sc/n{N}.json") if x.startswith("{")][-1])
    b = l.get("bf16", {})
    v, vb = l["value"], b.get("rays_per_s")
    if N == 1:
        base = (v, vb)
    print(f"  {N}   {4096 // N:9d}  {l['ms_per_step']:12.4f}  {v:22.1f}  {v / (N * base[0]) * N:5.3f}  {b.get('ms_per_step', float('nan')):12.4f}  {vb or float('nan'):11.1f}  {(vb or 0) / base[1]:5.3f}  {800 // N:15d}  {l.get('frame_ms') or float('nan'):8.2f}")
PY
