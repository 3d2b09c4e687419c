"""A/B timing of the fused MLP kernel: the shipped library against a variant built with
`python -m nerf_pytorch_paeng_amd.build --variant TAG -D...`, alternating in ONE process on ONE box (box-to-box
variance is ~0.5 %, more than most single changes):  python tools/ab_probe.py TAG[,TAG2,...] [rounds] [bf16] [points_per_wave] [rays] [S]
A TAG may carry its defines, `diag:-DMN_DIAG` (the timing switches of rounds 2-4 are gone from the sources: tools/ABLATIONS.md): variants live in build_scratch/, which does not travel to the GPU box, so a missing one is
built where the probe runs (~1.5 min of box time each)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_pytorch_paeng_amd import _lib, ops, synthetic, weights

tag = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"
f16s = len(sys.argv) > 3 and sys.argv[3] == "f16s"           # the split-precision variant
ppw = int(sys.argv[4]) if len(sys.argv) > 4 else 0           # bf16 launch shape: 0 = the launcher's choice, 64 / 32 pinned
N = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
S = int(sys.argv[6]) if len(sys.argv) > 6 else 192
use_bf16 = 5 if f16s else ({0: 1, 64: 2, 32: 3}[ppw] if bf16 else 0)
dev = torch.device("cuda:0")
packed = weights.PackedNeRF.from_state_dict(synthetic.make_state_dict(0, 8, 256), dev)
K, H, W = synthetic.lego_camera()
pix = torch.from_numpy(synthetic.pixel_batch(H, W, N, 0)).to(dev)
o, d = ops.make_o_d_pixels(W, H, K, synthetic.pose_spherical(0.0, -30.0, 4.0), pix)
rays = torch.cat([o, d], -1).contiguous()
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0]
raw = torch.empty(N, S, 4, device=dev)

libs = {"shipped": _lib.lib()}
from nerf_pytorch_paeng_amd import build as _build
for spec in tag.split(","):
    tg, _, defs = spec.partition(":")
    path = _build.build_variant(tg, [d for d in defs.split(":") if d]) if defs or not os.path.exists(_build.variant_path(tg)) else _build.variant_path(tg)
    h = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    libs[tg] = h


def time(lib, iters=20):
    ms = C.c_float(0.0)
    blob = packed.f16s()[1] if f16s else (packed.bf16()[1] if bf16 else packed.fine)
    rc = lib.mi_nerf_time_mlp_rays(C.byref(packed.net), blob.data_ptr(), rays.data_ptr(), z.data_ptr(), N, S, raw.data_ptr(), iters, use_bf16,
                                   C.byref(ms), torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0
    return ms.value


for lib in libs.values():
    time(lib, 3)
res = {k: [] for k in libs}
for _ in range(rounds):
    for k, lib in libs.items():
        res[k].append(time(lib))
for k, v in res.items():
    shape = f", {ppw or 'auto'} pts/wave" if bf16 else ""
    print(f"{k:10s} {N} x {S} launch{shape}: min {min(v):.4f} ms  median {sorted(v)[len(v) // 2]:.4f} ms   {[round(x, 4) for x in v]}")
