#!/bin/bash
# round 6: A/B/C of the bf16 small-shard step: the build before the fused stages (prev), with the coarse launch's fused middle only (mid), and the shipped build
# (both: + the fine launch's owned rays and fused final composite), the three library builds in alternating processes on one box.
# tools/prev/*.so = builds of earlier commits (made by hand for this run, not kept).
export PYTHONPATH=.
for i in 1 2 3; do
  for L in tools/prev/libmi_nerf_prev.so tools/prev/libmi_nerf_mid.so ""; do
    MI_NERF_LIB=$L python3 tools/r06_fused_middle_probe.py 2>/dev/null | grep -v amdgpu
  done
done
