"""CPU-side checks of the product's host logic: the C-ABI library loads and exports every declared
symbol, the weight packer produces a blob whose lane-level emulation (tests/blob_emulator.py)
reproduces the oracle MLP, argument validation fails loudly.  No GPU compute here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from nerf_pytorch_paeng_amd import _lib, ops, synthetic, weights
from oracle import restate as R
from tests.blob_emulator import Emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mi_nerf.h")).read()
    declared = set(re.findall(r"\b(mi_nerf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mi_nerf_net", "mi_nerf_params"}
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(handle, n)]
    assert not missing, missing
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.lib().mi_nerf_abi_version() == _lib.ABI_VERSION


def test_infer_net_and_blob_sizes():
    for D, W, skip in ((8, 256, 4), (4, 128, -1), (6, 128, 4)):
        sd = synthetic.make_state_dict(1, D, W)
        net = weights.infer_net(sd)
        assert (net.D, net.W, net.skip, net.L_x, net.L_d) == (D, W, skip, 10, 4)
        blob = ops.pack_module(sd, "model_coarse.", net)
        assert blob.numel() == _lib.lib().mi_nerf_packed_bytes(ctypes.byref(net))


def test_pack_rejects_bad_shapes_and_unsupported_nets():
    sd = synthetic.make_state_dict(1, 4, 128)
    net = weights.infer_net(sd)
    bad = dict(sd); bad["model_coarse.linear_feat.weight"] = np.zeros((128, 64), np.float32)
    with pytest.raises(_lib.MiNerfError):
        ops.pack_module(bad, "model_coarse.", net)
    with pytest.raises(_lib.MiNerfError):
        ops.pack_module(synthetic.make_state_dict(1, 4, 64), "model_coarse.", ops.make_net(4, 64, -1))


def test_cpu_tensors_are_refused():
    with pytest.raises(_lib.MiNerfError):
        ops.stratified_z(2.0, 6.0, torch.rand(4, 64))


@pytest.mark.parametrize("D,W", [(8, 256), (4, 128)])
def test_blob_emulation_matches_oracle_embedded(D, W):
    sd = synthetic.make_state_dict(11, D, W)
    net = weights.infer_net(sd)
    blob = ops.pack_module(sd, "model_fine.", net).numpy()
    rs = np.random.RandomState(0)
    x = rs.uniform(-1, 1, size=(32, 90)).astype(np.float32)
    emu = Emu(blob)
    # embedded-mode registers: gather channels exactly as gather_regs<> does
    def gather(row_block, L):
        K = ((3 * L + 2 + 3) // 4) * 4
        reg = np.zeros((K, 64))
        for s in range(3 * L):
            ch = 3 + 6 * (s // 3) + (s % 3)
            reg[s, :32], reg[s, 32:] = row_block[:, ch], row_block[:, ch + 3]
        reg[3 * L, :32], reg[3 * L, 32:] = row_block[:, 0], row_block[:, 1]
        reg[3 * L + 1, :32] = row_block[:, 2]
        return reg
    out = emu.tile(gather(x[:, :63].astype(np.float64), 10), de=gather(x[:, 63:].astype(np.float64), 4))
    ref = R.mlp_forward(sd, "model_fine.", torch.from_numpy(x), D, 63, 27, dtype=torch.float64).numpy()
    np.testing.assert_allclose(out, ref, atol=1e-9, rtol=1e-9)


def test_blob_emulation_matches_oracle_fused():
    D, W = 8, 256
    sd = synthetic.make_state_dict(5, D, W)
    net = weights.infer_net(sd)
    blob = ops.pack_module(sd, "model_coarse.", net).numpy()
    rs = np.random.RandomState(1)
    ray = rs.normal(size=6)
    z = np.sort(rs.uniform(2, 6, 32))
    p = ray[:3, None] + ray[3:, None] * z[None, :]
    v = ray[3:] / np.linalg.norm(ray[3:])
    emu = Emu(blob)
    g = R.posenc(torch.from_numpy(v[None]), 4)[0].numpy()
    out = emu.tile(emu.enc_regs(10, p), dir_gamma=g)
    x = torch.cat([R.posenc(torch.from_numpy(p.T.copy()), 10), torch.from_numpy(g)[None].expand(32, 27)], -1)
    ref = R.mlp_forward(sd, "model_coarse.", x, D, 63, 27, dtype=torch.float64).numpy()
    np.testing.assert_allclose(out, ref, atol=1e-9, rtol=1e-9)


def test_model_mirror_has_reference_checkpoint_keys():
    """The NeRF mirror must accept the reference's checkpoints: same state_dict keys and shapes as model/NeRF.py:24-30,58-59."""
    from nerf_pytorch_paeng_amd.model import NeRF
    for D, W in ((8, 256), (4, 128)):
        m = NeRF(D, W, 63, 27, skips=[4], gt_camera_param=(None, None))
        sd = synthetic.make_state_dict(0, D, W)
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        want = {k: tuple(v.shape) for k, v in sd.items()}
        assert got == want
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        assert m.get_camera_gt() == (None, None)
    with pytest.raises(_lib.MiNerfError):        # parameters on CPU: the MI355X path refuses, it never falls back
        m(torch.rand(2, 90))
