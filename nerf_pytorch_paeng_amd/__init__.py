"""nerf_pytorch_paeng_amd -- MI355X-native NeRF volume-rendering hot path.

A from-scratch gfx950 implementation of the per-ray forward path of nuggy875/NeRF_pytorch_paeng
(nerf_process.py + rays.py + model/), behind that project's own Python call surface:

    from nerf_pytorch_paeng_amd.nerf_process import batchify_rays_and_render_by_chunk, render_rays
    from nerf_pytorch_paeng_amd.rays import make_o_d
    from nerf_pytorch_paeng_amd.model import NeRF, get_positional_encoder

All compute runs in hand-written HIP kernels (libmi_nerf.so, C ABI in include/mi_nerf.h); there is no
PyTorch-op or CPU fallback -- importing the compute modules without the built library raises.
"""
__version__ = "0.1.0"

__all__ = ["nerf_process", "rays", "model", "ops", "weights", "synthetic", "dist"]
