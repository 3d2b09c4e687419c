"""A/B timing of the batched weight-gradient launch (nine 256 x 256 products over the fine net's 786 432 points, the form the backward pass runs):
the shipped library against variants built with `-D...`, alternating in ONE process on ONE box.
    python tools/wgrad_ab_probe.py TAG:-DFOO[,TAG2:-DBAR] [rounds] [f16s|fp32]
The f16s entry includes one pass over each gradient operand that finds its scale (the same in every variant)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_pytorch_paeng_amd import _lib, build as _build

tags = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f16s = (sys.argv[3] if len(sys.argv) > 3 else "f16s") == "f16s"
dev = torch.device("cuda:0")
P, n = 4096 * 192, 9
g = torch.Generator(device=dev).manual_seed(0)
dlt = [torch.randn(P, 256, device=dev, generator=g) * 1e-3 for _ in range(2)]
xin = [torch.relu(torch.randn(P, 256, device=dev, generator=g)) for _ in range(2)]          # ReLU-sparse like the stash rows
outs = [torch.empty(256, 256, device=dev) for _ in range(n)]
bias = [torch.empty(256, device=dev) for _ in range(n)]

libs = {"shipped": _lib.lib()}
for spec in tags.split(","):
    tg, _, defs = spec.partition(":")
    path = _build.build_variant(tg, [d for d in defs.split(":") if d]) if defs or not os.path.exists(_build.variant_path(tg)) else _build.variant_path(tg)
    h = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    libs[tg] = h
scratch = torch.empty(int(_lib.lib().mi_nerf_wgrad_scratch_bytes()), dtype=torch.uint8, device=dev)
PP, II = C.c_void_p * n, C.c_int * n


def run(lib, iters):
    ms = C.c_float(0.0)
    fn = lib.mi_nerf_wgrad_products_f16s if f16s else lib.mi_nerf_wgrad_products
    rc = fn(n, PP(*[dlt[b % 2].data_ptr() for b in range(n)]), II(*[256] * n), II(*[256] * n), PP(*[xin[b % 2].data_ptr() for b in range(n)]), II(*[256] * n),
            II(*[256] * n), P, PP(*[o.data_ptr() for o in outs]), II(*[256] * n), PP(*[b.data_ptr() for b in bias]), scratch.data_ptr(), scratch.numel(),
            iters, C.byref(ms), torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0, lib.mi_nerf_last_error()
    return ms.value


for lib in libs.values():
    run(lib, 2)
res = {k: [] for k in libs}
for _ in range(rounds):
    for k, lib in libs.items():
        res[k].append(run(lib, 5))
for k, v in res.items():
    print(f"{k:10s} nine 256x256 products, {P} points, {'split precision' if f16s else 'fp32 MFMA'}: min {min(v):.4f} ms  median {sorted(v)[len(v) // 2]:.4f} ms   {[round(x, 4) for x in v]}")
