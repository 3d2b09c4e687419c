"""Time one weight-gradient product (256x256 over P points) at several P: python tools/wgrad_probe.py
Separates the fixed cost (launch, accumulator write-out, slice reduction) from the steady-state MFMA rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nerf_pytorch_paeng_amd import ops

dev = torch.device("cuda:0")
PEAK = 157.3e12
for P in (98304, 196608, 393216, 786432, 1572864):
    d = torch.randn(P + 64, 256, device=dev)
    x = torch.randn(P + 64, 256, device=dev)
    ops.wgrad_product(d, 256, x, 256, P, iters=2)
    _, _, ms = ops.wgrad_product(d, 256, x, 256, P, iters=10, timed=True)
    flop = 2.0 * 256 * 256 * P
    print(f"P={P:8d}  {ms * 1e3:8.1f} us  {flop / (ms * 1e-3) / 1e12:6.1f} TFLOP/s  ({flop / (ms * 1e-3) / PEAK * 100:5.1f} % of fp32 MFMA peak)")
    del d, x
