export PYTHONPATH=.
for i in 1 2 3; do
  for L in tools/prev/libmi_nerf_prev.so ""; do
    MI_NERF_LIB=$L python3 tools/r06_fused_middle_probe.py 2>/dev/null | grep -v amdgpu
  done
done
